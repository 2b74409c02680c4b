// context.h -- private definition of ngsq_ctx (shared by context.cpp, results.cpp, synth.hip)
#pragma once

#include <hip/hip_runtime_api.h>

#include <string>
#include <vector>

#include "../../include/ngsq.h"
#include "kernels.h"

struct ngsq_ctx;

namespace ngsq {

enum KernelId { K_FIELDS = 0, K_GC, K_QUAL, K_EDITS, K_COV_SCAN, K_EDITS_VAF, K_H2D, K_FEATURES, K_COV_STREAM, K_INFLATE, K_INFLATE_CRC,
                K_REC_INDEX, K_REC_COLUMNS, K_COUNT };

struct PendingTime {
    int id;
    hipEvent_t a, b;
};

// HIP-event bracket around the launches of one kernel family on the context's stream (cfg.timing);
// always counts launches and algorithmic bytes.  context.cpp
struct KernelTimer {
    ngsq_ctx *c;
    int id;
    hipEvent_t a = nullptr, b = nullptr;
    hipStream_t s; // the stream the bracketed launches go to (default: the context's)
    KernelTimer(ngsq_ctx *c, int id, uint64_t algo_bytes, hipStream_t stream = nullptr);
    ~KernelTimer();
};

struct Staging {
    uint8_t *buf = nullptr;
    uint64_t cap = 0;
    hipEvent_t done = nullptr;
};

int grow_quality_table(ngsq_ctx *c, uint64_t rows); // context.cpp: at least `rows` cycles in the quality table
int reference_join(ngsq_ctx *c);                    // reference.cpp: wait for ngsq_reference_load's thread; its verdict
void reference_abandon(ngsq_ctx *c);                // reference.cpp: ngsq_destroy

} // namespace ngsq

struct ngsq_ctx {
    ngsq_config cfg{};
    std::vector<uint32_t> ref_len;
    std::vector<uint8_t> primary;
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipEvent_t copy_done = nullptr;
    ngsq::LaunchInfo li{};
    ngsq::DeviceState st{};
    uint64_t n_counters = 0, n_depth = 0, n_edits = 0, n_cov_hist = 0;
    std::vector<uint64_t> depth_off, edits_off, bin_off;
    uint32_t *d_ref_len = nullptr;
    uint64_t *d_depth_off = nullptr, *d_edits_off = nullptr, *d_bases_off = nullptr;
    uint8_t *d_ref_bases = nullptr;      // both packed copies of the reference (Edits)
    std::vector<uint64_t> bases_off;     // byte offset of each sequence in either copy (NO_DEPTH: none)
    uint64_t nbases = 0;                 // bytes of one copy
    uint32_t *d_edits_len = nullptr;     // [2 n_refs] st.ref_edits_len | st.ref_fast_len (null: ref_len everywhere)
    uint32_t *d_bad_off = nullptr, *d_bad_pos = nullptr; // st.ref_bad_off / _pos
    void *ref_loader = nullptr;          // reference.cpp: the thread of ngsq_reference_load and what it reports
    bool ref_deferred = false, ref_ready = false;
    // Edits teardown: per sequence the carry of every 4096-entry chunk of its difference array, and which chunks have been
    // turned into refs so far ([lo, hi); a sharded run converts a slice per rank, ngsq_get_edits_positions the rest)
    uint32_t *d_edits_carry = nullptr;
    uint8_t *d_edits_td = nullptr;       // tables of the all-sequences Edits teardown (kernels.h EditsSeq)
    size_t edits_td_cap = 0;
    std::vector<uint8_t> h_edits_td;
    unsigned long long *d_edits_defer = nullptr; // one bit per record of the batch: left to k_edits_walk (edits_kernel.hip)
    uint64_t edits_defer_cap = 0;
    std::vector<uint64_t> edits_carry_off, edits_conv_lo, edits_conv_hi;
    uint32_t *d_first_chunk = nullptr;
    uint64_t *d_bin_off = nullptr;
    uint64_t n_diff = 0, n_chunks = 0; // difference-array entries / scan chunks of the depth block
    // teardown results (per-sequence depth histograms | bin totals | VAF histogram): ONE block of
    // partial sums, so that shards which tear down disjoint chunk ranges can add them up
    unsigned long long *d_td = nullptr;
    uint64_t n_td = 0;
    unsigned long long *d_touched = nullptr; // [2] min / max+1 element of the depth block written
    unsigned long long h_touched[2] = {~0ull, 0};
    uint64_t scan_lo = 0, scan_hi = 0; // chunk range this context tears down (default: all)
    uint32_t scan_carry = 0;
    const uint32_t *scan_words = nullptr; // shard exchange: per-rank words on the device; carry += words[2r] for r in scan_front
    uint64_t scan_front = 0;
    uint32_t vaf_part = 0, vaf_parts = 1; // Edits teardown: this context does slice vaf_part of vaf_parts of every sequence
    void *xchg_scratch = nullptr;         // exchange.cpp
    bool scan_partial = false, torn_down = false;
    unsigned long long *d_cov_hist = nullptr, *d_bin_totals = nullptr, *d_vaf = nullptr;
    std::vector<unsigned long long> h_counters, h_cov_hist, h_bin_totals, h_vaf;
    bool finalized = false;
    bool edits_uploaded = false; // ngsq_state_upload(which = 2) since the last reset: any sequence's slots may hold data
    // ngsq_finalize's results land here (pinned host memory the device addresses, written by one kernel), then in the vectors above
    unsigned long long *pin_results = nullptr, *pin_results_dev = nullptr;
    uint64_t pin_words = 0;
    // streaming Coverage (cfg.sorted_input): per-batch scratch column, per-sequence plan, chunk flags
    bool stream_cov = false;
    uint32_t *d_cov_end = nullptr;
    uint64_t cov_end_cap = 0;
    uint32_t *d_stream_u32 = nullptr; // end_acc | prev_end | plan_a | plan_z | plan_h | plan_t | guard_until (n_refs each) | batch_span
    unsigned long long *d_last_key = nullptr;
    uint32_t span_turn = 0;           // which of the two largest-span words the next batch uses (launch_all)
    uint8_t *d_chunk_flags = nullptr;
    ngsq::CovStreamArgs csa{};
    // Genomic Features gene model (ngsq_set_features)
    ngsq::FeatureTables ft{};
    uint32_t *d_ft_idx = nullptr, *d_ft_starts = nullptr, *d_ft_stops = nullptr;
    uint8_t *d_ft_primary = nullptr;
    bool have_features = false;
    // batches that came before ngsq_set_features: their records' (flag, sequence, position, span) kept on the device
    struct DeferredFeatures {
        uint8_t *buf;
        size_t bytes;
        uint64_t n;
    };
    std::vector<DeferredFeatures> ft_deferred;
    ngsq::Staging stage[2];
    int stage_next = 0;
    ngsq_kernel_time timing[ngsq::K_COUNT]{};
    std::vector<ngsq::PendingTime> pending;
    std::vector<hipEvent_t> event_pool;
    std::string err;
};
