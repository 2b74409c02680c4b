/*
 * ngsq_synth.h -- synthetic record batches of SURVEY.md 8(d) for tests and
 * bench.py (the reference's `ngs generate` makes FASTQ from a FASTA with an
 * unseeded RNG, src/generate/command.rs:59-131, and cannot produce these).
 * Records are a pure function of (seed, index): see ngsq_shared.h.
 */
#ifndef NGSQ_SYNTH_H
#define NGSQ_SYNTH_H

#include "ngsq.h"
#include "ngsq_shared.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Byte / op totals of records [first, first+n): sizes of the seq, qual and cigar
 * columns.  FIXED mode uses fixed strides ((l+1)/2, l, 1 op); MIXED uses offsets. */
int ngsq_synth_sizes(const ngsq_synth_config *cfg, uint64_t first, uint64_t n, uint64_t *seq_bytes,
                     uint64_t *qual_bytes, uint64_t *cigar_ops);

/* Fill caller-allocated HOST columns of `batch` (the const is cast away: the
 * pointers must be writable).  For MIXED mode seq_off/qual_off/cigar_off must
 * point to n+1 entries; for FIXED mode they must be NULL and the strides set. */
int ngsq_synth_fill_host(const ngsq_synth_config *cfg, uint64_t first, uint64_t n,
                         const ngsq_batch *batch);

/* Same on the context's device with DEVICE column pointers, generated in place
 * by a HIP kernel (bit-identical to ngsq_synth_fill_host). */
int ngsq_synth_fill_device(ngsq_ctx *ctx, const ngsq_synth_config *cfg, uint64_t first, uint64_t n,
                           const ngsq_batch *batch);

/* The synthetic reference sequence `ref` as ngsq_config.ref_bases wants it: codes[p] = ngsq_synth_ref_code(cfg, ref, p), one
 * 4-bit code per byte, for p in [0, len).  With cfg->seq_model = NGSQ_SYNTH_SEQ_FROM_REFERENCE the reads are sampled from it. */
int ngsq_synth_fill_reference(const ngsq_synth_config *cfg, uint32_t ref, uint8_t *codes, uint64_t len, int n_threads);

/* GENOME mode (ngsq_shared.h): room[r] = the alignment starts of the sequences in front of r; n + 1 entries */
int ngsq_synth_genome_room(const uint32_t *genome_len, uint32_t n, uint64_t *room);

/* ngsq_synth_write_bam for a GENOME-mode configuration: the @SQ lines carry ref_names[r] and cfg->genome_len[r] */
int ngsq_synth_write_bam_named(const ngsq_synth_config *cfg, const char *const *ref_names, const char *path, uint64_t n_records, int level,
                               int n_threads);

/* Write records [0, n_records) as a BGZF BAM (@SQ chr1 [, chr2]) plus a minimal BAI
 * ("<path>.bai"), rendered and deflated by n_threads workers (0 = all cores). */
int ngsq_synth_write_bam(const ngsq_synth_config *cfg, const char *path, uint64_t n_records, int level,
                         int n_threads);

#ifdef __cplusplus
}
#endif
#endif
