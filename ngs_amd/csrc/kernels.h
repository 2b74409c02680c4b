// kernels.h -- host-callable launchers of the gfx950 kernels (kernels.hip) and the
// layout of the device-resident accumulator blocks they update.
//
// Layout of the packed uint64 "counters" block (one per context).  Every entry is
// a sum of per-record contributions, so blocks of different shards add
// element-wise:
//
//   [C_GENERAL .. +16)      RecordMetrics, order of ngsq_general_metrics
//   [C_CIGAR1  .. +9)       read_one_cigar_ops by BAM op code
//   [C_CIGAR2  .. +9)       read_two_cigar_ops
//   C_TLEN_PROCESSED, C_TLEN_IGNORED
//   C_GC_GC, C_GC_AT, C_GC_OTHER, C_GC_PROCESSED, C_GC_IGN_FLAGS, C_GC_IGN_SHORT
//   C_COV_NONSENSICAL
//   [C_ERR .. +8)           ngsq_error_counts (first eight)
//   [C_FEAT .. +9)          ngsq_features_metrics; C_FEAT_ERR_REF, C_FEAT_ERR_POS (the last two error counts)
//   [OFF_GC_HIST .. +101)   GC histogram
//   [off_tlen .. +tlen_cap+1)
//   [off_qual .. +max_read_len*94)   per-cycle quality table, row = 0-based cycle
//   [off_edits1 .. +513) [off_edits2 .. +513)   per-read edit-count histograms
//   [off_seen .. +n_refs)   records Coverage processed per sequence (entry exists iff > 0)
//   [off_eseen .. +n_refs)  non-zero iff Edits wrote anything for the sequence (its teardown and its reset are skipped otherwise)
#pragma once

#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include "../../include/ngsq.h"

// The kernels of the context's stream (record index, columns, facets) share the CUs with the BGZF decoders of the next chunk
// (another stream, persistent waves that keep the scalar and vector issue ports busy): their waves ask for the highest
// issue priority, or a latency-bound kernel like the record-chain walk runs ten times slower beside the decoders than alone.
#ifndef NGSQ_FOREGROUND_WAVE
#define NGSQ_FOREGROUND_WAVE() __builtin_amdgcn_s_setprio(3)
#endif

namespace ngsq {

enum : uint32_t {
    C_GENERAL = 0,
    C_CIGAR1 = 16,
    C_CIGAR2 = 25,
    C_TLEN_PROCESSED = 34,
    C_TLEN_IGNORED = 35,
    C_GC_GC = 36,
    C_GC_AT = 37,
    C_GC_OTHER = 38,
    C_GC_PROCESSED = 39,
    C_GC_IGN_FLAGS = 40,
    C_GC_IGN_SHORT = 41,
    C_COV_NONSENSICAL = 42,
    C_ERR = 43,
    E_MISSING_REF = 0,
    E_BAD_QUAL = 1,
    E_READ_TOO_LONG = 2,
    E_EDITS_BAD_REF = 3,
    E_EDITS_SHORT = 4,
    E_EDITS_NOT_CONSUMED = 5,
    E_EDITS_TOO_MANY = 6,
    E_BAD_CIGAR = 7,
    C_FEAT = 51,
    F_UTR5 = 0,
    F_UTR3 = 1,
    F_CDS = 2,
    F_INTERGENIC = 3,
    F_EXONIC = 4,
    F_INTRONIC = 5,
    F_PROCESSED = 6,
    F_IGN_FLAGS = 7,
    F_IGN_NONPRIMARY = 8,
    C_FEAT_ERR_REF = 60,
    C_FEAT_ERR_POS = 61,
    C_COV_UNSORTED = 62, // sorted_input contexts: adjacent record pairs out of coordinate order
    OFF_GC_HIST = 64,
    OFF_TLEN_HIST = 168,
};

constexpr uint32_t QUAL_BINS = NGSQ_MAX_SCORE + 1; // 94
constexpr uint32_t QUAL_LDS_MAX_ROWS = 320;        // cycles kept in the LDS table
constexpr uint32_t QUAL_WIN_MAX_R = 20;            // 16-byte windows per row the fast path is built for
constexpr uint64_t NO_DEPTH = ~0ull;

// device view of the context shared by all kernels
struct DeviceState {
    unsigned long long *counters; // packed block above
    uint32_t off_tlen, off_qual, off_edits1, off_edits2, off_seen, off_eseen;
    uint32_t tlen_cap, max_read_len, n_refs, cov_cap;
    uint32_t *depth;               // coverage difference arrays, all primary sequences
    uint32_t *chunk_sums;          // per COV_CHUNK positions: sum of the difference entries
    uint32_t *super_sums;          // per COV_SUPER chunks
    unsigned long long *touched;   // [2] min / max+1 element of `depth` written (shard exchange)
    const uint64_t *ref_depth_off; // [n_refs] element offset into depth, NO_DEPTH if not primary
    const uint32_t *ref_len;       // [n_refs]
    uint32_t *edits;               // per sequence with bases: L+1 entries that hold the difference array of the `M` cover until the
                                   // teardown turns them into refs per position, then L+1 entries of alts (edits_kernel.hip)
    const uint64_t *ref_edits_off; // [n_refs] element offset of that pair; alts follow at +L+1
    const uint8_t *ref_bases;      // the reference as packed 4-bit codes in SEQ's nibble order, starting at base 0 of each sequence ...
    const uint8_t *ref_bases_odd;  // ... and starting at base 1 (a read at an odd 0-based position finds its bytes here)
    const uint64_t *ref_bases_off; // [n_refs] byte offset of the sequence in either copy, NO_DEPTH if absent
    // what the FASTA holds of each sequence (ngsq_config.ref_bases_len, ngsq_reference_load); all null = ref_len bases of each
    const uint32_t *ref_edits_len; // [n_refs] bases of the sequence in the FASTA (more or fewer than ref_len): a read that ends beyond them fails (edits.rs:257-261)
    const uint32_t *ref_fast_len;  // [n_refs] min(that, ref_len): what the window lanes may reach; 0 = every record of the sequence takes the one-record walk (it has bad positions)
    const uint32_t *ref_bad_off;   // [n_refs + 1] the sequence's entries of ref_bad_pos
    const uint32_t *ref_bad_pos;   // sorted 1-based positions whose FASTA byte is no base letter: a read that covers one fails
    uint64_t gc_seed;
    // ---- streaming Coverage (sorted_input contexts, cov_stream.hip); all null otherwise
    uint32_t *cov_end;     // [batch records] scratch column: exclusive alignment end clipped to L+1, 0 = covers nothing
    uint32_t *end_acc;     // [n_refs] largest cov_end of any record so far
    uint32_t *batch_span;  // [1] largest (cov_end - alignment_start) of the current batch (one of two words used in turn)
};

// device view of one batch (all pointers device memory)
struct DeviceBatch {
    uint64_t n;
    uint64_t first_record_index;
    const uint16_t *flag;
    const uint8_t *mapq;
    const int32_t *ref_id;
    const int32_t *pos;
    const int32_t *mate_ref_id;
    const int32_t *tlen;
    const uint32_t *l_seq;
    const uint16_t *n_cigar;
    const uint8_t *seq;
    const uint64_t *seq_off;
    const uint8_t *qual;
    const uint64_t *qual_off;
    const uint32_t *cigar;
    const uint64_t *cigar_off;
    uint32_t seq_stride, qual_stride, cigar_stride;
    const uint64_t *record_id; // null: first_record_index + i
    uint64_t qual_bytes;       // bytes of the qual column as the batch states them (ngsq_batch.qual_bytes; host batches: qual_off[n])
};

// a record's number of CIGAR operations: the 16-bit column says 65535 for "that many or more, look at the offsets" (include/ngsq.h)
#if defined(__HIPCC__)
__device__ __forceinline__ uint32_t batch_n_ops(const DeviceBatch &b, uint64_t i) {
    uint32_t n = b.n_cigar[i];
    if (n == 0xFFFFu && b.cigar_off) n = (uint32_t)(b.cigar_off[i + 1] - b.cigar_off[i]);
    return n;
}
#endif

struct LaunchInfo {
    int n_cu; // compute units of the device
};

// General flag + CIGAR-op tallies (general.rs:31-121), Template Length
// (template_length.rs:79-87) and Coverage range-add (coverage.rs:148-180): fields_kernel.hip
// coverage: 0 = off, 1 = range-add into the difference arrays, 2 = streaming pass 1 (writes st.cov_end)
hipError_t launch_fields(const LaunchInfo &li, const DeviceState &st, const DeviceBatch &b,
                         uint32_t rec_facets, int coverage, hipStream_t s);

// Streaming Coverage for coordinate-sorted batches (cov_stream.hip): pass 2 over (pos, cov_end).
constexpr uint32_t CS_TILE = 256;          // records per wave tile
constexpr uint32_t CS_NONE = 0xFFFFFFFFu;  // "no streamed range on this sequence in this batch"
struct CovStreamArgs {
    uint32_t *plan_a, *plan_z;       // [n_refs] first start of the first / next start after the last streamable tile
    uint32_t *plan_h, *plan_t;       // [n_refs] streamed position range [H, T), chunk-aligned; CS_NONE = none
    uint32_t *prev_end;              // [n_refs] largest exclusive end of the batches before this one
    uint32_t *guard_until;           // [n_refs] head guard: no position below this is streamed (0 = sequence not met yet)
    unsigned long long *last_key;    // [1] sort key of the last record of the previous batch
    uint8_t *chunk_flags;            // [n_chunks] 1 = finished by the streaming pass: the teardown scan skips it
    unsigned long long *hist;        // as CovScanArgs
    unsigned long long *bin_totals;
    const uint64_t *bin_off;
    uint32_t bin_size, cov_cap, head_guard;
    uint32_t *span_next;             // the word the NEXT batch collects its largest span in (zeroed by k_cov_plan_refs)
};
hipError_t launch_cov_stream(const LaunchInfo &li, const DeviceState &st, const DeviceBatch &b, const CovStreamArgs &a,
                             hipStream_t s);
// GC Content (gc_content.rs:38-100)
hipError_t launch_gc(const LaunchInfo &li, const DeviceState &st, const DeviceBatch &b,
                     uint64_t seq_bytes, hipStream_t s);
// Quality Score (quality_scores.rs:37-49)
hipError_t launch_qual(const LaunchInfo &li, const DeviceState &st, const DeviceBatch &b,
                       hipStream_t s);
// Quality Score fast path for fixed-pitch rows (qual_kernel.hip)
bool qual_window_supported(const DeviceState &st, const DeviceBatch &b);
hipError_t launch_qual_window(const LaunchInfo &li, const DeviceState &st, const DeviceBatch &b, uint32_t nrot,
                              hipStream_t s);
// Genomic Features (features.rs:115-242): gene model as sorted coordinate lists, features_kernel.hip
struct FeatureTables {
    // intervals of name id k on sequence r: entries [idx[k * n_refs + r], idx[k * n_refs + r + 1]) of
    // `starts` (sorted) and, independently sorted, `stops`
    const uint32_t *idx;    // [5 * n_refs + 1]
    const uint32_t *starts; // [n]
    const uint32_t *stops;  // [n]
    const uint8_t *primary; // [n_refs]
    uint32_t n_refs;
    uint32_t role_name[5]; // name id of NGSQ_ROLE_*
    unsigned long long *scratch; // [FT_SLOTS * 16 + 1] zero between launches: per-slot partial tallies, then the launch's ticket
};
constexpr uint32_t FT_SLOTS = 32;
// a batch that arrives before the gene model: the four facts the facet needs of each record, as columns of a batch of their own
hipError_t launch_features_defer(const LaunchInfo &li, const DeviceBatch &b, uint16_t *flag, int32_t *ref, int32_t *pos, uint16_t *ncig, uint32_t *cig,
                                 hipStream_t s);
hipError_t launch_features(const LaunchInfo &li, const DeviceState &st, const DeviceBatch &b, const FeatureTables &ft,
                           hipStream_t s);
// Quality Score for the offsets layout with max_read_len <= 320 (qual_kernel.hip)
bool qual_ragged_supported(const DeviceState &st, const DeviceBatch &b);
hipError_t launch_qual_ragged(const LaunchInfo &li, const DeviceState &st, const DeviceBatch &b, hipStream_t s);
// Edits process (edits.rs:217-303), edits_kernel.hip; defer_bits: scratch of at least (b.n + 63) / 64 words (device memory)
// with_gc: the launch also tallies the GC Content facet of the same records (the SEQ column is then read once; only when
// edits_can_take_gc says so: fixed-pitch rows of up to 160 bases) -- launch_gc is then not called for the batch
bool edits_can_take_gc(const DeviceState &st, const DeviceBatch &b);
hipError_t launch_edits(const LaunchInfo &li, const DeviceState &st, const DeviceBatch &b, unsigned long long *defer_bits, bool with_gc,
                        hipStream_t s);

// Coverage teardown for every sequence in ONE launch (coverage.rs:182-246): prefix-sum the
// difference arrays, histogram the depths, integer bin totals; zeroes the arrays behind
// itself when `reset` is set.  cov_scan.hip
//
// Layout of the uint32 "depth" block: per covered sequence a difference array of
// ref_len+2 entries padded to a multiple of COV_CHUNK, then one sum per chunk
// (maintained by the range-add kernel), then one sum per COV_SUPER chunks (scratch of the scan).
constexpr uint32_t COV_CHUNK = 4096; // positions per scan chunk
constexpr uint32_t COV_SUPER = 256;  // chunks per super-chunk
struct CovScanArgs {
    uint32_t *depth;                 // whole block
    uint64_t n_chunks;               // chunks of all sequences
    uint64_t c_begin, c_end;         // chunk range to tear down (shards own disjoint ranges)
    uint32_t carry_in;               // sum of every difference entry in front of c_begin
    const uint32_t *carry_words;     // shard exchange: null, or [world][2] words in device memory; the sum of
    uint64_t carry_mask;             //   carry_words[2r] over the set bits r of carry_mask joins carry_in
    uint32_t *chunk_sums, *super_sums;
    const uint32_t *ref_first_chunk; // [n_refs + 1]
    const uint32_t *ref_len;         // [n_refs]
    const unsigned long long *seen;  // [n_refs] records Coverage processed
    unsigned long long *hist;        // [n_refs][cov_cap + 2]; last entry = positions with depth > cov_cap
    unsigned long long *bin_totals;  // concatenated per sequence
    const uint64_t *bin_off;         // [n_refs + 1] offsets into bin_totals
    uint32_t n_refs, bin_size, cov_cap;
    int reset;
    const uint8_t *chunk_flags;      // null, or per chunk 1 = already finished by the streaming pass (skipped)
};
hipError_t launch_cov_scan(const LaunchInfo &li, const CovScanArgs &a, hipStream_t s);

// Edits, edits_kernel.hip.  The reference bases (one 4-bit code per byte, device memory) -> the two packed copies the
// kernel compares with (n_bytes each, codes behind `len` read as 0); *bad counts bytes that are not 4-bit codes.
hipError_t launch_pack_reference(const LaunchInfo &li, const uint8_t *codes, uint64_t len, uint8_t *even, uint8_t *odd, uint64_t n_bytes,
                                 unsigned long long *bad, hipStream_t s);
// Edits teardown for one sequence (edits.rs:305-344): sums[c] = sum of the difference entries in front of 4096-entry chunk c;
// then, chunk by chunk, refs in place of the difference array and (vaf_hist != null) the VAF histogram
uint64_t edits_teardown_chunks(uint64_t n_entries);
uint64_t edits_teardown_carry_words(uint64_t n_entries); // words of the teardown's scratch for a sequence of n_entries - 1 bases
// touched (optional, device memory): the sequence's word of the counters block at off_eseen -- zero: nothing to do, the kernels return
// write_refs: refs = cover - alts in place (ngsq_get_edits_positions); false: the VAF histogram only, the arrays stay as they are
hipError_t launch_edits_chunk_sums(const uint32_t *diff, uint64_t n_entries, uint32_t *sums, const unsigned long long *touched, hipStream_t s);
// ... and of EVERY sequence in three launches: one entry per sequence with Edits state; a block finds its sequence in the tables of
// first-block numbers (n_seq + 1 entries each: chunk sums, super sums, the blocks that take the sequence's chunks [chunk0, chunk1))
struct EditsSeq {
    uint64_t edits_off;  // elements into st.edits: the difference array (n_entries), alts behind it
    uint64_t n_entries;  // ref_len + 1
    uint64_t carry_off;  // elements into the carry scratch
    uint32_t chunk0, chunk1; // this context's share of the sequence's chunks (a sharded run splits them over the ranks)
    uint32_t ref;        // index of the sequence: its word of the counters block at off_eseen
    uint32_t reserved;
};
hipError_t launch_edits_teardown_all(const EditsSeq *seqs, uint32_t n_seq, const uint32_t *first_sums, uint32_t n_sums, const uint32_t *first_supers,
                                     uint32_t n_supers, const uint32_t *first_refs, uint32_t n_refs_blocks, uint32_t *edits, uint32_t *carry,
                                     unsigned long long *vaf_hist, const unsigned long long *touched, hipStream_t s);
hipError_t launch_edits_refs(uint32_t *refs, const uint32_t *alts, uint64_t n_entries, const uint32_t *carry, uint64_t chunk0, uint64_t chunk1,
                             unsigned long long *vaf_hist, const unsigned long long *touched, bool write_refs, hipStream_t s);

// Up to STATE_SPANS_MAX small blocks of 32-bit words filled with a value (src == null) or copied (src -> dst; dst may be pinned
// host memory the device addresses) by ONE launch: the resets of ngsq_reset, the result download of ngsq_finalize.  kernels.hip
constexpr uint32_t STATE_SPANS_MAX = 16;
struct StateSpans {
    struct Span {
        uint32_t *dst;
        const uint32_t *src;
        uint64_t n_words;
        uint32_t value;
    } span[STATE_SPANS_MAX];
    uint32_t n = 0;
    bool overflow = false; // a span that did not fit: launch_state_spans then fails (hipErrorInvalidValue) instead of leaving a block untouched
    void fill(void *dst, uint64_t n_words, uint32_t value) {
        if (!n_words) return;
        if (n < STATE_SPANS_MAX) span[n++] = Span{static_cast<uint32_t *>(dst), nullptr, n_words, value};
        else overflow = true;
    }
    void copy(void *dst, const void *src, uint64_t n_words) {
        if (!n_words) return;
        if (n < STATE_SPANS_MAX) span[n++] = Span{static_cast<uint32_t *>(dst), static_cast<const uint32_t *>(src), n_words, 0};
        else overflow = true;
    }
};
hipError_t launch_state_spans(const StateSpans &a, hipStream_t s);

// synthetic records generated in place on the device (include/ngsq_shared.h)
struct SynthColumns {
    uint16_t *flag;
    uint8_t *mapq;
    int32_t *ref_id;
    int32_t *pos;
    int32_t *mate_ref_id;
    int32_t *tlen;
    uint32_t *l_seq;
    uint16_t *n_cigar;
    uint8_t *seq;
    uint64_t *seq_off;
    uint8_t *qual;
    uint64_t *qual_off;
    uint32_t *cigar;
    uint64_t *cigar_off;
    uint32_t seq_stride, qual_stride, cigar_stride;
};

} // namespace ngsq
