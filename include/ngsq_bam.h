/*
 * ngsq_bam.h -- host ingest for the `ngs qc` hot path: BGZF inflate + BAM record
 * parse -> structure-of-arrays batches (ngsq_batch, NGSQ_MEM_HOST) ready for
 * ngsq_process_batch.  SURVEY.md 8(f) rank 1 "the step before the path".
 *
 * Reference interfaces replaced (all in the un-vendored noodles crates the
 * reference calls; restated from the SAM/BAM specification, sections 4.1-4.2):
 *   - utils/formats/bam.rs:77-123  open_and_parse: open, require + parse <bam>.bai
 *                                  (IndexCheck::Full), read header + reference sequences
 *   - qc/command.rs:305            reader.records(&header): BGZF inflate + record decode
 * No GPU is needed by anything in this header.
 */
#ifndef NGSQ_BAM_H
#define NGSQ_BAM_H

#include "ngsq.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ngsq_bam ngsq_bam;

/* message of the last failing ngsq_bam_* call of this thread */
const char *ngsq_bam_last_error(void);

/* Open a BAM file and read its header and reference sequences.
 * n_threads: BGZF inflate workers (0 = hardware concurrency). */
int ngsq_bam_open(const char *path, int n_threads, ngsq_bam **out);
void ngsq_bam_close(ngsq_bam *bam);

/* utils/formats/bam.rs:86-96: "<path>.bai" must exist and parse as a BAI index
 * (magic, per-reference bins/chunks and linear index, optional n_no_coor). */
int ngsq_bam_check_index(const char *bam_path);

/* Region queries of the sequence-based pass (qc/command.rs:356-397: `reader.query(&header, &index, &region)` with
 * the region = one whole reference sequence).  In a coordinate-sorted file the records of a sequence are contiguous,
 * so the query is: the smallest chunk begin the index holds for the sequence, a seek, and sequential reads until the
 * sequence changes.
 *   ngsq_bam_index_ref_starts  start_voffset[r] = smallest virtual offset (block file offset << 16 | offset in the
 *                              block's data) of any chunk of reference r, 0 when the index holds none;
 *                              *n_bins = bins with records over all references (0: an index without bins, as some
 *                              writers produce for empty files -- scan the file instead)
 *   ngsq_bam_seek              continue ngsq_bam_next_batch (host reader) at a record boundary given as a virtual
 *                              offset; first_record_index of later batches then counts from that point only */
int ngsq_bam_index_ref_starts(const char *bam_path, uint32_t n_refs, uint64_t *start_voffset, uint64_t *n_bins);
int ngsq_bam_seek(ngsq_bam *bam, uint64_t voffset);

uint32_t ngsq_bam_n_refs(const ngsq_bam *bam);
const char *ngsq_bam_ref_name(const ngsq_bam *bam, uint32_t i);
uint32_t ngsq_bam_ref_len(const ngsq_bam *bam, uint32_t i);
const char *ngsq_bam_header_text(const ngsq_bam *bam, uint64_t *len);

/* Decode up to max_records further records into `out` (host SoA columns owned
 * by the reader, valid until the next call / close).  out->n_records == 0 at end
 * of file.  out->first_record_index = index of the first record in the file; out->record_id[i] = the record's BAM
 * virtual offset (include/ngsq.h: what the GC window offset is drawn from).
 * Layout: fixed-pitch rows (pitch = longest read of the batch) when that wastes
 * little, else offsets arrays; cigar pitch 1 when every record has <= 1 op. */
int ngsq_bam_next_batch(ngsq_bam *bam, uint64_t max_records, ngsq_batch *out);

/* records decoded so far */
uint64_t ngsq_bam_records_read(const ngsq_bam *bam);

/* ---- device ingest: the GPU inflates and parses ------------------- */

/* Device ingest of an opened BAM: like ngsq_bam_next_batch, but the compressed BGZF blocks are
 * copied to the context's device, inflated there (csrc/bgzf_inflate.hip) and parsed into
 * NGSQ_MEM_DEVICE columns (csrc/bam_device.hip) owned by the reader and valid until the next
 * call / close; work is enqueued on the context's stream.  Same records, same layout rules and
 * the same errors as the host reader; a call may return fewer than max_records before the end of
 * the file (out->n_records == 0 only at the end).  One handle serves either the host or the device
 * calls, not both (NGSQ_ERR_STATE).  out->record_id (device memory) holds the records' virtual offsets, as the host
 * reader's batches do.  Needs a GPU. */
int ngsq_bam_next_batch_device(ngsq_bam *bam, ngsq_ctx *ctx, uint64_t max_records, ngsq_batch *out);

/* ---- sharded device ingest: one BAM file, several GPUs -----------------
 * Shard s of n scans the BGZF blocks that start in [split(s), split(s+1)), split(k) = the first block start at or
 * after k * file_size / n (found the same way by every shard), through the same pipeline as a whole file: bounded
 * chunks, a reader thread with its pread workers (the node's cores divided by n), two raw buffers, the inflate of
 * chunk k+1 beside the parse of chunk k.  A record belongs to the shard its first byte lies in; the blocks that
 * complete a shard's last record (up to 17 MiB of them) are read too.  Pinned and device memory per shard are bounded
 * by the chunk size, not by the shard.
 *
 * Shards other than 0 do not know where their first record starts.  ngsq_bam_shard_begin(begin_voffset = 0) assumes
 * the first plausible record chain of the shard's first chunk and the scan runs on that assumption; afterwards
 * ngsq_bam_shard_end reports what was assumed (begin_voffset) and where the shard's own chain enters the next shard
 * (end_voffset; at the end of the file: file size << 16).  Shard s+1's begin must equal shard s's end: by induction from the header
 * the boundaries are then exact.  ngsq_bam_shard_verify (ngsq_comm.h) does the comparison with one all-gather and
 * re-arms a shard whose assumption was wrong with the confirmed offset (ngsq_bam_shard_begin again: the caller resets
 * its facets and scans that shard again).
 *
 * Records are numbered from 0 within the shard (first_record_index of the batches); what identifies a record for the
 * GC window is its record_id = BAM virtual offset (block file offset << 16 | offset in the block's data), the same in
 * every sharding.  first_key / last_key: sort keys of the shard's first and last record, for the order check between
 * neighbouring shards of a coordinate-sorted file. */
typedef struct ngsq_bam_shard_info {
    uint64_t n_records;          /* records starting in this shard */
    uint64_t begin_voffset;      /* first record of this shard */
    uint64_t end_voffset;        /* first record after this shard (file size << 16: none) */
    uint64_t first_record_index; /* records of the shards in front (set by ngsq_bam_shard_verify) */
    uint64_t first_key, last_key; /* refID << 32 | pos + 1 of the first / last record, ~0 for an unplaced one (n_records > 0) */
    uint32_t rescan;             /* ngsq_bam_shard_verify: 1 = this shard was re-armed, reset the facets and scan it again */
    uint32_t reserved;
} ngsq_bam_shard_info;
/* begin_voffset = 0: shard 0 starts behind the header, the others assume (see above); else the confirmed first record */
int ngsq_bam_shard_begin(ngsq_bam *bam, ngsq_ctx *ctx, uint32_t shard, uint32_t n_shards, uint64_t begin_voffset);
/* after ngsq_bam_next_batch_device has returned 0 records */
int ngsq_bam_shard_end(ngsq_bam *bam, ngsq_bam_shard_info *out);

/* What the device ingest of this handle has done so far (measurement and tests: how often the record index had to leave its
 * fast path).  segments: 16 KiB pieces of the inflated stream whose record chain was looked for; walk_one: those whose entry
 * was not among the chain starts the wave had kept (crowded out by bytes that look like records -- auxiliary data can -- or a
 * record the strict test rejects) and was walked by one thread instead; one per chunk is the record cut by the chunk's end. */
typedef struct ngsq_bam_ingest_stats {
    uint64_t chunks, segments, walk_one;
    uint64_t batches, batches_fixed_rows, batches_one_op; /* layout of the batches handed out: fixed-pitch SEQ/QUAL rows; <= 1 CIGAR op */
    uint64_t long_cigar_records;                           /* records whose CIGAR came from a CG:B,I tag (SAM specification 4.2.2) */
    uint64_t reserved;
} ngsq_bam_ingest_stats;
int ngsq_bam_device_stats(const ngsq_bam *bam, ngsq_bam_ingest_stats *out);

/* Inflate a buffer of WHOLE BGZF blocks (host memory) on the context's device and copy the
 * decompressed bytes back: one wavefront per block (csrc/bgzf_inflate.hip).  *out_len receives
 * the total ISIZE (also when out_cap is too small).  check_crc != 0 verifies every block's CRC32.
 * Errors (bad framing, invalid DEFLATE data, size or CRC mismatch): NGSQ_ERR_INVALID_ARGUMENT
 * with the block number in ngsq_last_error(ctx). */
int ngsq_bgzf_inflate_device(ngsq_ctx *ctx, const uint8_t *comp, uint64_t comp_len, uint8_t *out, uint64_t out_cap,
                             uint64_t *out_len, int check_crc);

#ifdef __cplusplus
}
#endif
#endif
