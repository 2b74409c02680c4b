/*
 * ngsq_reference.h -- the reference FASTA of the Edits facet, read and converted by the library.
 *
 * Reference interfaces replaced (paths relative to the reference tree):
 *   - src/utils/formats/fasta.rs:15-41      formats::fasta::open: the extension decides (FASTA; gzipped FASTA is refused
 *                                           with a message of its own), a noodles fasta::Reader over a BufReader
 *   - src/qc/sequence_based/edits.rs:120-133 EditsFacet::try_from: the file is opened once up front "to make sure that all
 *                                           is well" -> ngsq_fasta_open
 *   - src/qc/sequence_based/edits.rs:177-215 EditsFacet::setup, once per @SQ: open the file AGAIN, read records from the
 *                                           top until record.name() == the sequence's name, keep record.sequence();
 *                                           "sequence {} not found in reference FASTA." otherwise -> ngsq_reference_load
 *                                           (one pass over the file for all sequences; the first record of a name wins,
 *                                           as the reference's loop breaks at it)
 *   - src/qc/sequence_based/edits.rs:257-261 current_sequence.get(start..end) (None -> unwrap panic when the read runs
 *                                           past the FASTA's sequence) and Base::try_from per byte
 *
 * What crosses PCIe is the file's TEXT: host threads only copy it into pinned memory (no parsing on the host besides finding
 * the definition lines); newlines are dropped, bytes become 4-bit BAM base codes and the two packed copies the Edits kernels
 * compare with are written by HIP kernels.  A whole-genome FASTA (3.1 GB) costs the host one memcpy per byte.
 *
 * Semantics (shared with the oracle: oracle/oracle.h [N9]):
 *   - a record starts at a line whose first byte is '>'; its name is the text up to the first blank; its sequence is every
 *     byte of the following lines up to the next such line, without the line terminators ("\n" or "\r\n");
 *   - bytes convert as noodles-sam's Base::try_from(u8) does: the sixteen letters "=ACMGRSVTWYHKDBN" in EITHER CASE
 *     (soft-masked references -- about half of the GRCh38 analysis set is lower case -- are the normal input) to their
 *     4-bit BAM codes; any other byte is invalid, and, as in the reference, only a read that COVERS it (alignment start ..
 *     start + reference span, whatever the CIGAR operations) fails -- counted as edits_bad_reference;
 *   - a FASTA sequence shorter than its @SQ LN fails only the reads that run past its end (the reference's unwrap on
 *     `get`); inside a LONGER one a read may end beyond LN and fails only if an `M` base lies there (edits.rs:283-291:
 *     the per-position histograms have LN + 1 bins) or its slice holds an invalid byte; records the BAM has no @SQ
 *     for change nothing.
 */
#ifndef NGSQ_REFERENCE_H
#define NGSQ_REFERENCE_H

#include "ngsq.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ngsq_fasta ngsq_fasta;

/* Open a FASTA file and start indexing its definition lines on `n_threads` background threads (0 = a default).  Touches no
 * HIP: a host calls it first and initialises the device (ngsq_create) meanwhile.  A samtools index next to the file
 * (<path>.fai) is used instead of the scan when it agrees with the file at every record it names.
 * Errors (NGSQ_ERR_INVALID_ARGUMENT, text in ngsq_fasta_last_error): the reference's own, formats/fasta.rs:22-39. */
int ngsq_fasta_open(const char *path, int n_threads, ngsq_fasta **out);
void ngsq_fasta_close(ngsq_fasta *f);
const char *ngsq_fasta_last_error(void);
/* the records of the file, in file order (these wait for the index) */
int64_t ngsq_fasta_n_records(ngsq_fasta *f);
const char *ngsq_fasta_record_name(ngsq_fasta *f, uint32_t i);
/* bytes of the record's sequence lines, line terminators included (>= its length in bases) */
uint64_t ngsq_fasta_record_text_bytes(ngsq_fasta *f, uint32_t i);
/* seconds the index took / 1 when it came from <path>.fai */
double ngsq_fasta_index_seconds(ngsq_fasta *f);
int ngsq_fasta_index_from_fai(ngsq_fasta *f);

/* Base::try_from(u8): the 4-bit BAM code of a FASTA byte, or -1 (pure; the device kernel uses the same table) */
int ngsq_fasta_base_code(uint8_t byte);

/*
 * Load the reference of a context created with ngsq_config.ref_bases_deferred = 1: for every sequence r of the context
 * with wanted[r] != 0 (wanted == NULL: all of them) the FASTA record named ref_names[r].  Returns when the work is
 * QUEUED on a thread of the library; ngsq_process_batch waits for it before the first Edits kernel (the other facets'
 * kernels of the first batches run meanwhile), ngsq_reference_wait waits explicitly and reports.
 *   - a wanted sequence the FASTA does not have: ngsq_reference_wait / ngsq_process_batch fail with
 *     "sequence {name} not found in reference FASTA." (edits.rs:207-209);
 *   - sequences that are NOT wanted (a worker of a sharded scan loads only the sequences its byte range of the file can
 *     reach) hold no bases: an Edits record on one is counted as edits_bad_reference, never compared with zeros.
 * The handle may be closed once ngsq_reference_wait has returned.
 */
int ngsq_reference_load(ngsq_ctx *ctx, ngsq_fasta *f, const char *const *ref_names, const uint8_t *wanted);
int ngsq_reference_wait(ngsq_ctx *ctx);

/* what the load did (valid after ngsq_reference_wait) */
typedef struct ngsq_reference_stats {
    uint64_t text_bytes;      /* bytes of FASTA text that crossed PCIe                         */
    uint64_t bases;           /* bases installed                                               */
    uint64_t invalid_bytes;   /* bytes Base::try_from refuses (kept as positions: see above)   */
    uint32_t sequences;       /* sequences installed                                           */
    uint32_t shorter;         /* of them: shorter in the FASTA than their @SQ LN               */
    uint32_t longer;          /* ... longer (of the bases beyond LN only the count and the invalid positions are kept) */
    uint32_t reserved;
    double index_wait_s;      /* waiting for the definition-line index                         */
    double read_s;            /* first pread to last host-to-device copy queued                */
    double device_s;          /* conversion kernels (HIP events)                               */
    double total_s;           /* ngsq_reference_load to the last kernel's end                  */
} ngsq_reference_stats;
int ngsq_reference_get_stats(ngsq_ctx *ctx, ngsq_reference_stats *out);

#ifdef __cplusplus
}
#endif
#endif /* NGSQ_REFERENCE_H */
