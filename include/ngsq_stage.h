/*
 * ngsq_stage.h -- the per-record side of the boundary: a stager that turns the reference's
 * `process(&mut self, &Record)` calls into the batches of ngsq.h.
 *
 * Reference interfaces this adapter sits under (paths relative to the reference tree):
 *   - src/qc.rs:165        RecordBasedQualityControlFacet::process(&mut self, &Record)
 *   - src/qc.rs:203-219    SequenceBasedQualityControlFacet::{setup, process, teardown}
 *   - src/qc/command.rs:305-316   pass 1: every record to every record facet
 *   - src/qc/command.rs:356-397   pass 2: per sequence  setup -> query() -> process per record -> teardown
 *
 * A host that keeps the reference's driver loops (`impl RecordBasedQualityControlFacet for
 * GpuFacet`) calls ngsq_stager_push once per record with what noodles' accessors return, and ngsq_stager_flush when the
 * stager is full, at `summarize` (end of pass 1) and at every `teardown` (end of a sequence of pass 2).  The stager owns
 * pinned host columns in the layout of ngsq_batch (structure of arrays; SEQ packed 4-bit, high nibble first) and hands them
 * to ngsq_process_batch: fixed-pitch rows when every staged read has the same length, qualities and one CIGAR operation
 * (the fast kernels), the offsets layout otherwise -- the choice is made per flush and changes nothing in the results.
 *
 * Threading: as a context -- one thread (the reference's facets are `&mut self`).
 */
#ifndef NGSQ_STAGE_H
#define NGSQ_STAGE_H

#include <stddef.h>
#include <stdint.h>

#include "ngsq.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ngsq_stager ngsq_stager;

/* flags of ngsq_stager_create */
#define NGSQ_STAGE_PINNED 0x0u     /* columns in pinned host memory (hipHostMalloc): needs a HIP device                     */
#define NGSQ_STAGE_PAGEABLE 0x1u   /* ordinary memory: the host-to-device copies are staged by the runtime (slower); for
                                      hosts that fill the stager where no device is visible, and for CPU-side tests        */
#define NGSQ_STAGE_OFFSETS_ONLY 0x2u /* never hand over fixed-pitch rows (measurement aid)                                  */

/* ngsq_stager_push: the record has no identity of its own -- its ordinal in the pass is used (ngsq_batch.record_id == NULL).
 * All records of one flush either carry an id or none does (NGSQ_ERR_INVALID_ARGUMENT otherwise). */
#define NGSQ_STAGE_NO_ID (~0ull)

/* capacity_records: records a flush holds at most (the byte columns grow by themselves: reads of any length).
 * 1 << 21 records of 150 bases are 0.53 GB of pinned memory -- twice that: a pinned stager has two sets of columns (ngsq_stager_flush). */
int ngsq_stager_create(uint64_t capacity_records, uint32_t flags, ngsq_stager **out);
void ngsq_stager_destroy(ngsq_stager *s);
/* message of the stager's last failing call (s == NULL: of the calling thread's last failing create) */
const char *ngsq_stager_last_error(const ngsq_stager *s);

uint64_t ngsq_stager_len(const ngsq_stager *s);      /* records staged and not yet flushed            */
uint64_t ngsq_stager_capacity(const ngsq_stager *s);
uint64_t ngsq_stager_pushed(const ngsq_stager *s);   /* records pushed since create / the last rewind */

/*
 * One decoded record, as noodles-bam hands it to `process` (general.rs:36,81-91,103-105, template_length.rs:80,
 * gc_content.rs:41,50-52, quality_scores.rs:38,44, coverage.rs:159-160, edits.rs:227-265):
 *   flag         u16::from(record.flags())
 *   mapq         record.mapping_quality(), 255 for None
 *   ref_id       record.reference_sequence_id(), -1 for None      mate_ref_id likewise
 *   pos          alignment_start() - 1 (0-based), -1 for None
 *   tlen         record.template_length()
 *   bases        record.sequence(): l_seq entries, ONE 4-bit BAM code per byte ("=ACMGRSVTWYHKDBN": A=1 C=2 G=4 T=8 N=15);
 *                an entry above 15 is NGSQ_ERR_INVALID_ARGUMENT
 *   quals        record.quality_scores(): n_quals entries (Phred, <= 93: a larger score is counted on the device as the
 *                decode error it is in noodles).  n_quals is l_seq, or 0 for a record without qualities; anything else is
 *                NGSQ_ERR_INVALID_ARGUMENT (noodles refuses such a record while decoding).  l_seq scores that are ALL 0xFF
 *                are BAM's own encoding of "no qualities" and are staged as n_quals = 0 (as ngsq_stager_push_packed reads
 *                them), whatever the layout of the flush; l_seq ends at 2^31 - 1 (BAM's field)
 *   cigar        record.cigar(): n_cigar operations, len << 4 | op with op in 0..8 = MIDNSHP=X (any number of operations:
 *                the 16-bit n_cigar column saturates, ngsq.h)
 *   record_id    the record's identity for the GC window offset (ngsq.h): reader.virtual_position() before the record was
 *                read, or NGSQ_STAGE_NO_ID
 * Returns NGSQ_ERR_STATE when the stager is full (flush first: `if len == capacity { flush }` after every push, as
 * INTEGRATION.md section 4 does, never gets there).
 */
int ngsq_stager_push(ngsq_stager *s, uint16_t flag, uint8_t mapq, int32_t ref_id, int32_t pos, int32_t mate_ref_id, int32_t tlen,
                     uint32_t l_seq, const uint8_t *bases, const uint8_t *quals, uint32_t n_quals, const uint32_t *cigar,
                     uint32_t n_cigar, uint64_t record_id);

/* The same record from its own BAM bytes (a host below noodles can memcpy them): seq = (l_seq + 1) / 2 bytes, two bases per
 * byte, high nibble first; quals = l_seq bytes, all 0xFF for "no qualities" (SAM/BAM specification 4.2.3), or NULL for none. */
int ngsq_stager_push_packed(ngsq_stager *s, uint16_t flag, uint8_t mapq, int32_t ref_id, int32_t pos, int32_t mate_ref_id,
                            int32_t tlen, uint32_t l_seq, const uint8_t *seq_packed, const uint8_t *quals, const uint32_t *cigar,
                            uint32_t n_cigar, uint64_t record_id);

/* Records [first, first + count) of a HOST batch (any layout of ngsq.h), one ngsq_stager_push_packed each -- for a host that picks
 * records out of batches it already has (this repo's `ngs qc` applies the two `-n` rules that way), and the per-record path's rate
 * without a foreign-function call per record (bench.py `stager`).  Stops when the stager is full; *pushed = records taken. */
int ngsq_stager_push_records(ngsq_stager *s, const ngsq_batch *host_batch, uint64_t first, uint64_t count, uint64_t *pushed);

/* The staged records as the batch a flush would hand over (host pointers into the stager; valid until the next push, flush
 * or destroy).  For hosts that want to look, and for tests. */
int ngsq_stager_view(ngsq_stager *s, ngsq_batch *out);

/* facet.process for every staged record: ngsq_process_batch(ctx, staged batch, pass_mask), then the stager is empty.  A
 * pinned stager holds its columns TWICE: the flush queues the host-to-device copies of the set it hands over (NGSQ_PASS_NOWAIT),
 * records an event behind them on the context's stream and returns; the next pushes fill the other set, and a set is waited
 * for only when its turn comes again, one flush later -- the copies of a flush (5 ms per million 150-base records) run beside
 * the next flush's pushes (30 ms per million on one core), so a host pays for the pushes only.  (NGSQ_STAGE_PAGEABLE: one set,
 * the copies have landed when the call returns.)  pass_mask as in ngsq.h:
 * NGSQ_PASS_RECORD for the calls of pass 1, NGSQ_PASS_SEQUENCE for those of pass 2, NGSQ_PASS_BOTH for a host that makes one
 * pass.  Nothing staged: NGSQ_OK.  On a failure of ngsq_process_batch the records stay staged and the context has the message. */
int ngsq_stager_flush(ngsq_stager *s, ngsq_ctx *ctx, uint32_t pass_mask);

/* Start counting the records' ordinals from `first_record_index` again (the start of pass 2, command.rs:335: records without
 * an id of their own get their ordinal in the pass as one).  Only on an empty stager (NGSQ_ERR_STATE otherwise). */
int ngsq_stager_rewind(ngsq_stager *s, uint64_t first_record_index);

#ifdef __cplusplus
}
#endif
#endif /* NGSQ_STAGE_H */
