/*
 * ngsq.h -- C ABI of the MI355X-native `ngs qc` record-scanning hot path.
 *
 * This is the drop-in boundary: the entry points a host program (the
 * reference's Rust `ngs qc` driver through `extern "C"` FFI, this repo's C++
 * driver, or a ctypes test) binds in place of the reference's per-record
 * facet loop.  Plain pointers and sizes only; no C++ / torch types.
 *
 * Reference interfaces replaced (paths relative to the reference tree):
 *   - src/qc.rs:151-176   trait RecordBasedQualityControlFacet
 *                         {name, computational_load, process, summarize, aggregate}
 *   - src/qc.rs:184-229   trait SequenceBasedQualityControlFacet
 *                         {supports_sequence_name, setup, process, teardown, aggregate}
 *   - src/qc/command.rs:305-316  pass-1 loop  (record facets .process per record)
 *   - src/qc/command.rs:356-397  pass-2 loop  (sequence facets per reference sequence)
 *   - src/qc/results.rs:23-60    Results (aggregate + JSON)
 *
 * The unit of `process` here is a BATCH of records in structure-of-arrays
 * form instead of one `&Record`; every facet's state after `process` is a sum
 * of per-record integer contributions, so batching (and sharding) commutes.
 *
 * Threading: a context is not thread-safe (the reference's facets are
 * `&mut self`, single thread: src/qc/command.rs:226-421).
 */
#ifndef NGSQ_H
#define NGSQ_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI 6 (round 6): the STATE layout changed under ABI 5's name in round 5 and is now versioned -- the counters block holds
 * n_refs "Edits wrote here" words (one per sequence) in front of the quality table, so n_counters and every later offset
 * moved for ngsq_state_download / _upload and the ngsq_shard_state all-reduce; the edits block holds the difference array
 * of the `M` cover, not refs; device batches must be readable NGSQ_DEVICE_COLUMN_SLACK bytes behind every column.
 * ngsq_exchange compares the ABI version and the block sizes of all ranks before the first sum (a rank built against
 * another layout is refused, not summed).  New in 6: ngsq_config.ref_bases_len, ngsq_reference.h. */
#define NGSQ_ABI_VERSION 6u

/* ---- status codes (reference: anyhow::Result<()> / panic) ---- */
#define NGSQ_OK 0
#define NGSQ_ERR_INVALID_ARGUMENT (-1)
#define NGSQ_ERR_DEVICE (-2)           /* a HIP runtime call failed              */
#define NGSQ_ERR_NO_DEVICE (-3)        /* library loaded but no usable GPU       */
#define NGSQ_ERR_MALFORMED_RECORD (-4) /* the reference would Err or panic here  */
#define NGSQ_ERR_STATE (-5)            /* call out of lifecycle order            */
#define NGSQ_ERR_BUFFER_TOO_SMALL (-6)
#define NGSQ_ERR_UNSUPPORTED (-7)
#define NGSQ_ERR_UNSORTED (-8)         /* sorted_input was promised and a record broke the order */
#define NGSQ_ERR_LIMIT (-9)            /* an implementation limit, not a malformed input: a read longer than the quality
                                          table in a batch that did not say how long its reads are (device memory,
                                          offsets layout, max_l_seq = 0), or longer than NGSQ_QUALITY_ROWS_LIMIT; the
                                          reference has no such limit (quality_scores.rs:18 keeps a map per position) */

/* ---- facets (names: `name()` of each facet under src/qc/record_based, sequence_based) ---- */
#define NGSQ_FACET_GENERAL 0x01u         /* "General"          general.rs:23        */
#define NGSQ_FACET_TEMPLATE_LENGTH 0x02u /* "Template Length"  template_length.rs:71 */
#define NGSQ_FACET_GC_CONTENT 0x04u      /* "GC Content"       gc_content.rs:30     */
#define NGSQ_FACET_QUALITY_SCORE 0x08u   /* "Quality Score"    quality_scores.rs:29 */
#define NGSQ_FACET_COVERAGE 0x10u        /* "Coverage"         coverage.rs:125      */
#define NGSQ_FACET_EDITS 0x20u           /* "Edits"            edits.rs:165         */
#define NGSQ_FACET_FEATURES 0x40u        /* "Genomic Features" features.rs:107 (needs ngsq_set_features) */
#define NGSQ_FACETS_RECORD_BASED 0x4Fu   /* pass 1, command.rs:288-333 */
#define NGSQ_FACETS_SEQUENCE_BASED 0x30u /* pass 2, command.rs:335-400 */
#define NGSQ_FACETS_DEFAULT 0x1Fu        /* qc.rs:60-65,85-89 (Edits only with -r) */

/* ---- fixed shapes of the reference ---- */
#define NGSQ_N_CIGAR_KINDS 9   /* M I D N S H P = X  (BAM op codes 0..8)               */
#define NGSQ_MAX_SCORE 93      /* quality_scores.rs:26                                  */
#define NGSQ_GC_BINS 101       /* gc_content.rs:129-138  Histogram 0..=100              */
#define NGSQ_GC_WINDOW 100     /* gc_content.rs:20  TRUNCATION_LENGTH                   */
#define NGSQ_EDITS_BINS 513    /* histogram.rs:394-398 Histogram::default() 0..=512     */
#define NGSQ_VAF_BINS 101      /* edits.rs:64  zero_based_with_capacity(100)            */
#define NGSQ_N_RECORD_COUNTERS 16
#define NGSQ_MAX_READ_LEN_LIMIT 1024            /* (kept for callers of ABI 3: the table is no longer limited to it) */
#define NGSQ_QUALITY_ROWS_LIMIT (1u << 24)      /* cycles the quality table grows to (16 Mi cycles = 12.6 GB of counters) */

/* where a batch's column pointers live */
#define NGSQ_MEM_HOST 0u
#define NGSQ_MEM_DEVICE 1u
/* NGSQ_MEM_DEVICE batches: the kernels read the seq and qual columns (and cigar) with 16-byte vector loads, so the last
 * row is read up to 15 bytes past its end.  Every column of a device batch must be READABLE for this many bytes behind its
 * last element (the bytes' values do not matter).  ngsq_device_malloc adds the slack to every allocation by itself, the
 * readers of ngsq_bam.h and ngsq_synth.h allocate with it, host batches are staged with it; only a caller that hands
 * over memory from its own hipMalloc has to size it `bytes + NGSQ_DEVICE_COLUMN_SLACK`. */
#define NGSQ_DEVICE_COLUMN_SLACK 64u

typedef struct ngsq_ctx ngsq_ctx;

/*
 * Context configuration.  Mirrors what the reference fixes at facet
 * construction (src/qc.rs:44-126) plus the header facts the sequence-based
 * facets receive per call (`&Map<ReferenceSequence>`: name + length).
 */
typedef struct ngsq_config {
    uint32_t struct_size;   /* = sizeof(ngsq_config)                                       */
    uint32_t facets;        /* NGSQ_FACET_* mask                                           */
    int32_t device;         /* HIP device ordinal                                          */
    uint32_t n_refs;        /* number of @SQ reference sequences (BAM header order)        */
    const uint32_t *ref_len;        /* [n_refs] @SQ LN                                     */
    const uint8_t *ref_is_primary;  /* [n_refs] 1 iff Coverage `supports_sequence_name`
                                       (coverage.rs:133-138, genome.rs:59-83)              */
    uint32_t bin_size;      /* coverage bin, qc.rs:87 (50 000); 0 -> 50 000                */
    uint32_t tlen_cap;      /* template length capacity, qc.rs:62 (1024); 0 -> 1024        */
    uint32_t cov_cap;       /* coverage.rs:76 (2048); 0 -> 2048                            */
    uint32_t max_read_len;  /* rows the per-cycle quality table starts with (0 -> 512); it grows to the longest
                               read of the batches (see ngsq_batch.max_l_seq)                              */
    uint64_t gc_seed;       /* seed of the deterministic GC window offset (ngsq_gc_offset) */
    const uint8_t *const *ref_bases; /* Edits only: [n_refs] pointers to ref_len[r] bytes of
                                        4-bit BAM base codes (one code per byte, every byte <= 15:
                                        anything else is NGSQ_ERR_INVALID_ARGUMENT), host memory;
                                        NULL entries => sequence not in FASTA.  A host that has the
                                        FASTA as a FILE leaves this NULL, sets ref_bases_deferred and
                                        calls ngsq_reference_load (ngsq_reference.h)          */
    void *stream;           /* optional hipStream_t to launch on; NULL -> library creates one */
    uint32_t timing;        /* 1 -> bracket every kernel with HIP events (ngsq_kernel_timing) */
    uint32_t sorted_input;  /* 1 -> the caller promises coordinate-sorted records over all batches of this
                               context (what `ngs qc` requires: a BAI exists only for sorted files,
                               formats/bam.rs:86-96).  Coverage then finishes positions while it streams
                               the records instead of keeping a depth array for the teardown; a record that
                               breaks the promise makes ngsq_finalize fail with NGSQ_ERR_UNSORTED           */
    uint32_t cov_head_guard; /* sorted_input shards other than the first of a file: number of positions
                               after this context's first record on a sequence that records of the shard
                               in front may still cover (kept on the exchanged depth array); 0 = none      */
    uint32_t ref_bases_deferred; /* 1 -> Edits state and room for the bases of EVERY sequence; the bases come from
                               ngsq_reference_load (ngsq_reference.h), which ngsq_process_batch waits for        */
    const uint32_t *ref_bases_len; /* optional, with ref_bases: [n_refs] bases the FASTA holds of each sequence when that
                               differs from ref_len (the reference slices the FASTA's sequence, edits.rs:257-261: a read
                               that runs past ITS end fails, counted as edits_bad_reference; a read that ends beyond
                               ref_len inside a longer sequence fails only if an `M` base lies there, edits.rs:283-291 --
                               the VALUES of bases beyond ref_len are never looked at).  ref_bases[r] then points to at
                               least min(ref_bases_len[r], ref_len[r]) bytes.  NULL = every sequence has ref_len bases */
} ngsq_config;

/*
 * One batch of decoded records, structure of arrays.  Field meanings are those
 * of the BAM record (SAM spec 4.2) that noodles exposes to the facets:
 *   flag        record.flags()                       general.rs:36
 *   mapq        record.mapping_quality() (255 = missing)  general.rs:88-91
 *   ref_id      record.reference_sequence_id()  (-1 = None)
 *   pos         0-based leftmost position (-1 = None); alignment_start = pos+1
 *   mate_ref_id record.mate_reference_sequence_id()  general.rs:82-83
 *   tlen        record.template_length()             template_length.rs:80
 *   l_seq       sequence length in bases
 *   seq         packed 4-bit bases "=ACMGRSVTWYHKDBN", high nibble first, (l_seq+1)/2 bytes
 *   qual        Phred bytes.  With qual_off: a record with missing qualities has ZERO
 *               qual bytes (noodles yields an empty QualityScores for 0xFF-filled BAM
 *               quals).  With a fixed stride: each row holds qual_stride bytes and the
 *               byte 0xFF means "no score at this cycle" (row padding beyond l_seq, or a
 *               whole 0xFF row = missing qualities, exactly BAM's own encoding)
 *   n_cigar     the record's number of CIGAR operations, SATURATED at 65535: the column is 16 bits wide, as the BAM field is.
 *               A record with more operations (SAM specification 4.2.2: its BAM CIGAR is the placeholder <l_seq>S<span>N
 *               and the real one sits in a CG:B,I tag, which noodles resolves while decoding -- the readers of ngsq_bam.h
 *               do the same) says 65535 here and its real count is cigar_off[i+1] - cigar_off[i]: such a batch addresses
 *               its CIGARs through cigar_off (ABI 5; ABI 4 refused these records)
 *   cigar       BAM encoding len<<4|op, op in 0..8 = MIDNSHP=X
 *   record_id   optional: one 64-bit identity per record, the only input of the GC window offset besides gc_seed and
 *               l_seq (ngsq_gc_offset).  NULL -> first_record_index + i, the record's ordinal in the file.  The
 *               readers of ngsq_bam.h fill it with the record's BAM virtual offset (file offset of the BGZF block
 *               its first byte lies in << 16 | offset in that block's data): the same for every way of cutting a
 *               file into shards, known to a shard without counting the records in front of it (a noodles host
 *               has it as reader.virtual_position() before each record)
 * Variable-length columns are addressed either by an offsets array
 * (`*_off[i] .. *_off[i+1]`, n_records+1 entries, units: bytes for seq/qual,
 * ops for cigar) or, when the offsets pointer is NULL, by a fixed stride
 * (`i * *_stride`): fixed-width batches are the fast path.
 */
typedef struct ngsq_batch {
    uint32_t struct_size;       /* = sizeof(ngsq_batch)                               */
    uint32_t location;          /* NGSQ_MEM_HOST or NGSQ_MEM_DEVICE                   */
    uint64_t n_records;
    uint64_t first_record_index; /* index of record 0 in the whole file (GC offset fn when record_id == NULL) */
    const uint16_t *flag;
    const uint8_t *mapq;
    const int32_t *ref_id;
    const int32_t *pos;
    const int32_t *mate_ref_id;
    const int32_t *tlen;
    const uint32_t *l_seq;
    const uint16_t *n_cigar;
    const uint8_t *seq;
    const uint64_t *seq_off;    /* NULL -> fixed stride */
    const uint8_t *qual;
    const uint64_t *qual_off;   /* NULL -> fixed stride rows, 0xFF = absent               */
    const uint32_t *cigar;
    const uint64_t *cigar_off;  /* NULL -> fixed stride (ops per record) */
    uint32_t seq_stride;        /* bytes per record when seq_off == NULL  */
    uint32_t qual_stride;       /* bytes per record when qual_off == NULL */
    uint32_t cigar_stride;      /* ops per record when cigar_off == NULL  */
    uint32_t max_l_seq;         /* the longest read of the batch, or 0 = not known.  The quality table grows to it
                                   before the batch is scanned.  Not needed for fixed-pitch rows (the pitch says it)
                                   nor for host batches (the library looks); a DEVICE batch in the offsets layout
                                   that leaves it 0 must keep to the table's current rows (ngsq_max_read_len) --
                                   longer reads are then counted as read_too_long and ngsq_finalize returns
                                   NGSQ_ERR_LIMIT.  The readers of ngsq_bam.h fill it in. */
    /* totals of the variable-length columns; required for NGSQ_MEM_DEVICE batches
     * that use offsets arrays (the host cannot read them), otherwise 0 = derive */
    uint64_t seq_bytes;
    uint64_t qual_bytes;
    uint64_t cigar_ops;
    const uint64_t *record_id;  /* NULL -> first_record_index + i */
} ngsq_batch;

/* Which pass of the reference driver a batch belongs to (command.rs:288-400).
 * The reference applies `-n` differently to the two passes, so the host may
 * feed different record subsets to each; with no `-n`, pass both. */
#define NGSQ_PASS_RECORD 0x1u
#define NGSQ_PASS_SEQUENCE 0x2u
#define NGSQ_PASS_BOTH 0x3u
/* OR-ed into pass_mask, host batches only: ngsq_process_batch returns when the copies to the device are QUEUED, not when they
 * have landed -- the host columns stay untouched until an event the CALLER records on ngsq_stream(ctx) behind the call has
 * completed (the explicit "batch done" event of SURVEY.md 8b: the next batch is assembled in a second set of columns
 * meanwhile).  include/ngsq_stage.h does exactly that. */
#define NGSQ_PASS_NOWAIT 0x100u

/* ---- plain-integer result blocks (all counters are the reference's usize) ---- */

/* general/metrics.rs:24-92 RecordMetrics + :11-20 ReadDesignationMetrics + :96-104 CigarMetrics */
typedef struct ngsq_general_metrics {
    uint64_t total;
    uint64_t unmapped;
    uint64_t duplicate;
    uint64_t primary;
    uint64_t secondary;
    uint64_t supplementary;
    uint64_t primary_mapped;
    uint64_t primary_duplicate;
    uint64_t paired;
    uint64_t read_1;
    uint64_t read_2;
    uint64_t proper_pair;
    uint64_t singleton;
    uint64_t mate_mapped;
    uint64_t mate_reference_sequence_id_mismatch;
    uint64_t mate_reference_sequence_id_mismatch_hq;
    uint64_t read_one_cigar_ops[NGSQ_N_CIGAR_KINDS]; /* index = BAM op code (MIDNSHP=X) */
    uint64_t read_two_cigar_ops[NGSQ_N_CIGAR_KINDS];
} ngsq_general_metrics;

/* gc_content/metrics.rs:10-33 */
typedef struct ngsq_gc_metrics {
    uint64_t histogram[NGSQ_GC_BINS];
    uint64_t total_gc_count;
    uint64_t total_at_count;
    uint64_t total_other_count;
    uint64_t processed;
    uint64_t ignored_flags;
    uint64_t ignored_too_short;
} ngsq_gc_metrics;

/* data conditions on which the reference aborts the run (SURVEY 8b "Error convention") */
typedef struct ngsq_error_counts {
    uint64_t missing_reference_id;  /* general.rs:81-83 unwrap() on None                     */
    uint64_t bad_quality_score;     /* score > 93: noodles decode error / quality_scores.rs:45 */
    uint64_t read_too_long;         /* a read longer than the quality table's rows in a batch that did not announce it
                                       (ngsq_batch.max_l_seq): implementation limit -> NGSQ_ERR_LIMIT */
    uint64_t edits_bad_reference;   /* edits.rs:242-261 slice out of range / no sequence     */
    uint64_t edits_record_short;    /* alignment.rs:84-87 "consume a record base"            */
    uint64_t edits_not_consumed;    /* alignment.rs:100-104 not fully consumed               */
    uint64_t edits_too_many;        /* edits.rs:297-299 edits > 512 -> unwrap panic          */
    uint64_t bad_cigar_op;          /* op code > 8                                           */
    uint64_t features_missing_reference_id; /* features.rs:132-140 bail!: mapped record, no reference id */
    uint64_t features_missing_position;     /* features.rs:171-174 bail!: no alignment start           */
} ngsq_error_counts;

/* features/metrics.rs:10-80, in declaration order */
typedef struct ngsq_features_metrics {
    uint64_t utr_five_prime_count;
    uint64_t utr_three_prime_count;
    uint64_t coding_sequence_count;
    uint64_t intergenic_count;
    uint64_t exonic_count;
    uint64_t intronic_count;
    uint64_t processed;
    uint64_t ignored_flags;
    uint64_t ignored_nonprimary_chromosome;
} ngsq_features_metrics;

/*
 * The gene model of the Genomic Features facet: what GenomicFeaturesFacet::try_from
 * (features.rs:270-355) keeps of the GFF.  Every GFF record whose sequence is a primary one and whose
 * type is one of the five configured feature names becomes rust_lapper::Interval { start: GFF start,
 * stop: GFF end } (half-open in the lookup, features.rs:314-318 -- a quirk that is reproduced).
 * Names are compared as strings by the reference, so two roles configured with the same name act as
 * one name: role_name[] carries that (distinct names get distinct ids < 5).
 */
#define NGSQ_ROLE_FIVE_PRIME_UTR 0
#define NGSQ_ROLE_THREE_PRIME_UTR 1
#define NGSQ_ROLE_CODING_SEQUENCE 2
#define NGSQ_ROLE_EXON 3
#define NGSQ_ROLE_GENE 4
typedef struct ngsq_features {
    uint32_t struct_size;
    uint32_t role_name[5]; /* name id of each NGSQ_ROLE_*                         */
    uint64_t n;            /* intervals                                             */
    const uint32_t *ref_id; /* [n] index of the interval's sequence in the header   */
    const uint32_t *name;   /* [n] name id (one of role_name[])                     */
    const uint32_t *start;  /* [n] GFF start (1-based)                              */
    const uint32_t *stop;   /* [n] GFF end                                          */
} ngsq_features;

/* one row of ngsq_kernel_timing */
typedef struct ngsq_kernel_time {
    const char *name;    /* static string */
    uint64_t launches;
    double total_ms;     /* sum of HIP-event elapsed times on the context's stream */
    uint64_t algo_bytes; /* algorithmic bytes the launches processed   */
} ngsq_kernel_time;

/* ---- lifecycle ---- */

uint32_t ngsq_abi_version(void);
/* number of HIP devices visible (0 when none); never fails */
int ngsq_device_count(void);
/* PCI address of a HIP device ("0000:c1:00.0", what /sys/bus/pci/devices and the device links under /sys/class/drm name it by), for hosts that
 * want to read the card's clocks or NUMA node; returns the length, 0 when there is no such device */
int ngsq_device_pci_bus_id(int device, char *buf, size_t cap);
/* facet display name for one NGSQ_FACET_* bit ("General", "Template Length", ...) or NULL */
const char *ngsq_facet_name(uint32_t facet_bit);
/* thread-local message of the last failing call that had no context */
const char *ngsq_last_global_error(void);
/* A finished device ingest (ngsq_bam_close) leaves its GiB-sized device and pinned buffers in a process-wide cache for
 * the next file: allocating and freeing them costs about as much as scanning 4 GB of BAM.  This gives them back to the
 * driver (ngsq_destroy does it too) and returns the bytes released.  NGSQ_POOL_MB=0 in the environment disables the cache. */
uint64_t ngsq_release_cached_memory(void);

int ngsq_create(const ngsq_config *cfg, ngsq_ctx **out);
void ngsq_destroy(ngsq_ctx *ctx);
const char *ngsq_last_error(const ngsq_ctx *ctx);

/* Install the gene model of the Genomic Features facet (the arrays are copied).  Replaces GenomicFeaturesFacet::try_from's
 * interval stores (features.rs:300-343).  The model may arrive AFTER the first batches (ABI 6: a host still reading its GFF need
 * not hold the scan back): ngsq_process_batch then keeps what the facet needs of the batch's records on the device (flag,
 * sequence, position, reference span: 16 bytes per record) and this call looks them up; it must have been called before
 * ngsq_exchange / ngsq_finalize when any batch was scanned (NGSQ_ERR_STATE otherwise). */
int ngsq_set_features(ngsq_ctx *ctx, const ngsq_features *features);

/* facet.process for every record of the batch (asynchronous on the context's
 * stream for device batches; host batches are copied to the device first and
 * the host buffers may be reused when the call returns). */
int ngsq_process_batch(ngsq_ctx *ctx, const ngsq_batch *batch, uint32_t pass_mask);

/* Sequence-facet teardown (coverage.rs:182-262, edits.rs:305-344) for every
 * sequence that saw a record, then copy all integer results to the host.
 * Returns NGSQ_ERR_MALFORMED_RECORD when any ngsq_error_counts field is set (NGSQ_ERR_LIMIT when the only one
 * is read_too_long). */
int ngsq_finalize(ngsq_ctx *ctx);

/* zero every accumulator so the context can scan another file */
int ngsq_reset(ngsq_ctx *ctx);
int ngsq_synchronize(ngsq_ctx *ctx);
void *ngsq_stream(ngsq_ctx *ctx); /* the hipStream_t kernels are launched on */

/* ---- results (valid after ngsq_finalize) ---- */

int ngsq_get_error_counts(const ngsq_ctx *ctx, ngsq_error_counts *out);
int ngsq_get_features(const ngsq_ctx *ctx, ngsq_features_metrics *out);
int ngsq_get_general(const ngsq_ctx *ctx, ngsq_general_metrics *out);
/* histogram: tlen_cap+1 bins (template_length.rs:44-53) */
int ngsq_get_template_length(const ngsq_ctx *ctx, uint64_t *histogram, size_t n_bins,
                             uint64_t *processed, uint64_t *ignored);
int ngsq_get_gc_content(const ngsq_ctx *ctx, ngsq_gc_metrics *out);
/* scores: row-major [ngsq_max_read_len(ctx)][94]; row i is 1-based cycle i+1 (quality_scores.rs:39-42) */
int ngsq_get_quality_scores(const ngsq_ctx *ctx, uint64_t *scores, size_t n_rows);
/* number of reference sequences / bins of one sequence's mean_coverage_per_bin */
uint32_t ngsq_n_refs(const ngsq_ctx *ctx);
uint32_t ngsq_max_read_len(const ngsq_ctx *ctx); /* rows of the quality table NOW (it grows) */
uint32_t ngsq_tlen_bins(const ngsq_ctx *ctx);
uint32_t ngsq_cov_bins(const ngsq_ctx *ctx);
/* 1 + floor(L/bin) + (L % bin != 0)   (coverage.rs:206-230) */
uint64_t ngsq_coverage_n_bins(const ngsq_ctx *ctx, uint32_t ref);
/* per sequence (coverage.rs:182-262): `seen` = the sequence has an entry in
 * coverage_per_position; histogram = that sequence's `coverages` (cov_cap+1
 * bins); ignored = pileup_too_large_positions; bin_totals[k] = integer sum of
 * depths of bin k (k = 0 is position 0 alone). */
int ngsq_get_coverage_sequence(const ngsq_ctx *ctx, uint32_t ref, int *seen, uint64_t *histogram,
                               size_t n_hist_bins, uint64_t *ignored, uint64_t *bin_totals,
                               size_t n_bins);
int ngsq_get_coverage_nonsensical(const ngsq_ctx *ctx, uint64_t *nonsensical_records);
/* edits.rs:32-57 */
int ngsq_get_edits(const ngsq_ctx *ctx, uint64_t *read_one_edits, uint64_t *read_two_edits,
                   size_t n_edit_bins, uint64_t *vaf_histogram, size_t n_vaf_bins);
/* refs_per_position / alts_per_position of one sequence (edits.rs:322-324), ref_len+1 entries each:
 * what the reference's --vaf-file lines are computed from (edits.rs:331-340) */
int ngsq_get_edits_positions(ngsq_ctx *ctx, uint32_t ref, uint32_t *refs, uint32_t *alts, size_t n);

/* Full `Results` JSON (results.rs:23-60, serde_json pretty layout, maps in
 * sorted / header order).  `ref_names` = @SQ names in header order.  Returns
 * the number of bytes needed (excluding the NUL); writes at most cap bytes. */
int64_t ngsq_results_json(const ngsq_ctx *ctx, const char *const *ref_names, char *buf, size_t cap);

/* ---- measurement ---- */
int ngsq_kernel_timing_count(const ngsq_ctx *ctx);
int ngsq_kernel_timing(const ngsq_ctx *ctx, int index, ngsq_kernel_time *out);
int ngsq_kernel_timing_reset(ngsq_ctx *ctx);

/* ---- multi-GPU exchange points: device pointers of the state
 * that is summed across shards before teardown.  `counters`: one packed
 * uint64 block (all record-facet tallies and histograms, nonsensical count,
 * per-sequence `seen` flags, error counts).  `depth`: one uint32 block holding,
 * per primary sequence, the coverage difference array (ref_len+2 entries);
 * `edits`: per sequence with reference bases 2 (ref_len+1) uint32 -- until the teardown the difference array of the `M`
 * cover (entry p-1 += 1 / entry q -= 1 for an M over positions p..q) and the mismatches per position (alts); the teardown
 * leaves both as they are (it only takes the VAF histogram); ngsq_get_edits_positions turns the first half into refs = cover - alts.  Any summation (RCCL all-reduce, or a host loop) of
 * these blocks across contexts followed by ngsq_finalize on one of them gives
 * the single-context result -- PROVIDED the contexts' quality tables have the same number of rows: the table is the last
 * part of the counters block and grows with the longest read a context has met (ngsq_max_read_len), so n_u64 and the device
 * pointer change when a batch brings a longer read (fetch them after the last batch, not before), and shards that met
 * different longest reads hold blocks of different sizes.  ngsq_exchange (ngsq_comm.h) equalises the rows first; a host
 * that sums the blocks itself creates its contexts with max_read_len = the longest read of the file. */
int ngsq_state_counters(ngsq_ctx *ctx, void **dev_ptr, uint64_t *n_u64);
int ngsq_state_depth(ngsq_ctx *ctx, void **dev_ptr, uint64_t *n_u32);
int ngsq_state_edits(ngsq_ctx *ctx, void **dev_ptr, uint64_t *n_u32);

/* Owner-computes teardown for sharded scans: instead of summing
 * whole depth blocks, shards exchange only the entries that fall into another
 * shard's part of the reference axis, each shard tears down a disjoint range of
 * 4096-entry chunks, and the (small) teardown results are summed.
 *   ngsq_depth_layout    entries of the difference arrays, number of chunks (the
 *                        per-chunk sums start at entry n_diff of the depth block),
 *                        and the [lo, hi) entry range this context has written
 *   ngsq_set_scan_range  chunks [lo, hi) to tear down and the sum of every entry in front
 *   ngsq_teardown        sequence-facet teardown on the device only (ngsq_finalize calls it
 *                        when it has not run yet)
 *   ngsq_state_teardown  device pointer of the teardown results: uint64 partial sums
 *                        (depth histograms, bin totals, VAF histogram) */
int ngsq_depth_layout(ngsq_ctx *ctx, uint64_t *n_diff, uint64_t *n_chunks, uint64_t *touched_lo,
                      uint64_t *touched_hi);
int ngsq_set_scan_range(ngsq_ctx *ctx, uint64_t chunk_lo, uint64_t chunk_hi, uint32_t carry_in);
int ngsq_teardown(ngsq_ctx *ctx);
int ngsq_state_teardown(ngsq_ctx *ctx, void **dev_ptr, uint64_t *n_u64);
/* sorted_input contexts: one byte per chunk, 1 = Coverage already finished the chunk while streaming
 * (the teardown skips it; a shard must not receive exchanged entries for such a chunk).  n = 0 otherwise. */
int ngsq_state_chunk_flags(ngsq_ctx *ctx, void **dev_ptr, uint64_t *n_u8);
/* host-visible copies for CPU-side reductions and tests */
int ngsq_state_download(ngsq_ctx *ctx, int which /*0 counters,1 depth,2 edits,3 teardown,4 chunk flags*/, void *dst,
                        uint64_t n_bytes);
int ngsq_state_upload(ngsq_ctx *ctx, int which, const void *src, uint64_t n_bytes);

/* ---- device memory helpers so a host without a HIP binding can stage batches ----
 * ngsq_device_malloc allocates n_bytes + NGSQ_DEVICE_COLUMN_SLACK (see there). */
int ngsq_device_malloc(ngsq_ctx *ctx, uint64_t n_bytes, void **dev_ptr);
int ngsq_device_free(ngsq_ctx *ctx, void *dev_ptr);
int ngsq_memcpy_h2d(ngsq_ctx *ctx, void *dev_dst, const void *host_src, uint64_t n_bytes);
int ngsq_memcpy_d2h(ngsq_ctx *ctx, void *host_dst, const void *dev_src, uint64_t n_bytes);
int ngsq_host_malloc_pinned(uint64_t n_bytes, void **host_ptr);
int ngsq_host_free_pinned(void *host_ptr);

/* ---- shared pure functions (same code on host and device) ---- */

/* The reference picks the GC window start with ThreadRng (gc_content.rs:69-74:
 * gen_range(0..l_seq-100), upper bound exclusive; 0 when l_seq == 100), which is
 * not reproducible.  This is the pinned replacement with the same support. */
uint32_t ngsq_gc_offset(uint64_t gc_seed, uint64_t record_index, uint32_t l_seq);

#ifdef __cplusplus
}
#endif
#endif /* NGSQ_H */
