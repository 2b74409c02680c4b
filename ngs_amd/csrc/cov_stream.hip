// cov_stream.hip -- Coverage (coverage.rs:148-246) for coordinate-sorted batches without the
// difference array: the depth of a position is final as soon as every record that can cover it has
// been seen, so a tile of consecutive records finishes the positions between its first start and
// the next tile's first start in LDS (range-add -> prefix sum -> depth histogram + bin totals).
//
// Contract (ngsq_config.sorted_input): the records of a context arrive in coordinate order, as in the
// BAM files `ngs qc` accepts (it requires a BAI, formats/bam.rs:86-96).  k_fields<STREAM> counts every
// adjacent pair that breaks the order and ngsq_finalize then fails with NGSQ_ERR_UNSORTED; the kernels
// here stay memory-safe on such input but their results are discarded.
//
// Per batch and sequence r the tiles (CS_TILE = 256 consecutive records, one wave each) whose first
// record and whose successor's first record lie on r are STREAMABLE; tile t owns the positions
// [start(first record of t), start(first record of t+1)).  The union over the run of streamable
// tiles is [a_r, z_r); of that, whole scan chunks [H_r, T_r) are streamed:
//   H_r = chunk_ceil(max(a_r, 1 + largest end of any earlier batch, a_r + head_guard on first touch))
//   T_r = chunk_floor(min(z_r, L_r + 1))
// Positions outside [H_r, T_r) -- the seams between batches, shards and sequences -- keep the classic
// path: every record adds the parts of its range outside [H_r, T_r) to the difference array (global
// atomics + chunk sums), and the teardown scan (cov_scan.hip) skips the chunks flagged here.  A -1
// that would land exactly on H_r is recorded in the chunk sum only, so no entry is ever written into
// a streamed chunk and the scan's running sum stays exact across the skipped chunks.
//
// A tile must also see the records of EARLIER tiles that reach into its positions.  A wave works through
// consecutive tiles and carries their ends forward in an LDS list (everything in front with end > first
// owned position: about one entry per unit of depth); the list is filtered and extended with the tile's own
// records after each tile.  Without a valid list -- the first tile of a wave, after a sequence boundary, an
// empty tile or a pile deeper than the list -- the tile walks back through the (pos, cov_end) columns, 64
// records at a time, until a group's first record cannot reach any more (start + largest span of the batch
// <= first owned position) or lies on another sequence, and rebuilds the list from what it found.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "kernels.h"

namespace ngsq {

typedef unsigned long long u64;

constexpr uint32_t ST_THREADS = 256;
constexpr uint32_t ST_WAVES = ST_THREADS / 64;
#ifndef ST_W_N
#define ST_W_N 1024
#endif
constexpr uint32_t ST_W = ST_W_N; // positions of one wave's LDS window
#ifndef ST_LIST_N
#define ST_LIST_N 232
#endif
#ifndef ST_HOT_BINS_N
#define ST_HOT_BINS_N 64
#endif
constexpr uint32_t ST_LIST = ST_LIST_N;  // entries of a wave's list of open ends (a deeper pile walks back instead)
// The depths of neighbouring positions are a dozen values around the running depth: 64 lanes adding to the
// histogram word of their depth serialise on those few words (74 % of this kernel's LDS time were same-address
// conflicts).  So the tally goes to a HOT window first: ST_HOT_BINS bins around the running depth, ST_HOT_REP
// copies of each (lane & 3 picks the copy), re-anchored when the depth at the start of a pass drifts out of its
// middle three quarters -- the copies are then summed into the wave's histogram.  Depths outside the window go to the
// histogram directly, in the same instruction (one address select per lane, no branch).
#ifndef ST_HOT_REP_N
#define ST_HOT_REP_N 4
#endif
constexpr uint32_t ST_HOT_BINS = ST_HOT_BINS_N, ST_HOT_REP = ST_HOT_REP_N, ST_HOT = ST_HOT_BINS * ST_HOT_REP;
#ifndef ST_EXP
#define ST_EXP 0 // measurement builds only (tools/exp_stream.sh): 1 no look-back, 2 no prefix passes, 3 no histogram atomics, 4 neither
#endif

__device__ __forceinline__ u64 st_key(int32_t rf, int32_t ps) { // as in k_fields<STREAM>
    return rf < 0 ? 0xFFFFFFFF00000000ull : ((u64)(uint32_t)rf << 32) | (uint32_t)(ps + 1);
}
__device__ __forceinline__ bool st_valid(const DeviceState &st, int32_t rf, int32_t ps) {
    return rf >= 0 && (uint32_t)rf < st.n_refs && ps >= 0 && st.ref_depth_off[rf] != NO_DEPTH;
}

// one thread per tile: first start of each run of streamable tiles (plan_a) and the start that ends it (plan_z)
__global__ __launch_bounds__(256) void k_cov_plan_tiles(DeviceState st, DeviceBatch b, CovStreamArgs a) {
    NGSQ_FOREGROUND_WAVE();
    const uint64_t n_wt = (b.n + CS_TILE - 1) / CS_TILE;
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 0) { // coordinate order across the batch boundary
        const u64 k0 = st_key(b.ref_id[0], b.pos[0]);
        if (k0 < *a.last_key) atomicAdd(&st.counters[C_COV_UNSORTED], 1ull);
        *a.last_key = st_key(b.ref_id[b.n - 1], b.pos[b.n - 1]);
    }
    // first records of tiles t-1 .. t+2 (absent: -1): every thread loads its own tile's, the neighbours' come from
    // the neighbouring lanes (the tile firsts are 1 KiB apart: four loads per thread fetched 0.19 GB per launch)
    const uint32_t lane = threadIdx.x & 63u;
    auto first_of = [&](int64_t u, int32_t &r, int32_t &p) {
        const bool ok = u >= 0 && (uint64_t)u < n_wt;
        r = ok ? b.ref_id[(uint64_t)u * CS_TILE] : -1;
        p = ok ? b.pos[(uint64_t)u * CS_TILE] : -1;
    };
    int32_t rf[4], ps[4];
    first_of((int64_t)t, rf[1], ps[1]);
    rf[0] = __shfl_up(rf[1], 1, 64), ps[0] = __shfl_up(ps[1], 1, 64);
    rf[2] = __shfl_down(rf[1], 1, 64), ps[2] = __shfl_down(ps[1], 1, 64);
    rf[3] = __shfl_down(rf[1], 2, 64), ps[3] = __shfl_down(ps[1], 2, 64);
    if (lane == 0) first_of((int64_t)t - 1, rf[0], ps[0]);
    if (lane == 63) first_of((int64_t)t + 1, rf[2], ps[2]);
    if (lane >= 62) first_of((int64_t)t + 2, rf[3], ps[3]);
    if (t >= n_wt) return;
    auto streamable = [&](int k) { // tile t-1+k and its successor start on the same covered sequence
        return st_valid(st, rf[k], ps[k]) && rf[k + 1] == rf[k] && ps[k + 1] >= 0;
    };
    if (!streamable(1)) return;
    const bool prev_same = streamable(0) && rf[0] == rf[1];
    const bool next_same = streamable(2) && rf[2] == rf[1];
    if (!prev_same) atomicMin(&a.plan_a[rf[1]], (uint32_t)ps[1] + 1);
    if (!next_same) atomicMax(&a.plan_z[rf[1]], (uint32_t)ps[2] + 1);
}

// one thread per sequence: the streamed chunk range of this batch; rolls the per-sequence end forward
__global__ __launch_bounds__(256) void k_cov_plan_refs(DeviceState st, CovStreamArgs a) {
    NGSQ_FOREGROUND_WAVE();
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    // the NEXT batch's largest-span word (the batches use two words in turn): last read by the previous batch's k_cov_stream,
    // next written by the next batch's k_fields -- no memset per batch
    if (r == 0 && a.span_next) *a.span_next = 0;
    if (r >= st.n_refs) return;
    const uint32_t pa = a.plan_a[r], pz = a.plan_z[r], prev = a.prev_end[r];
    const uint32_t acc = st.end_acc[r];
    // Head guard of a shard behind the first: records of the shard in front may cover the first head_guard positions
    // after this context's first record on the sequence, so they stay on the exchanged depth array -- in whichever
    // batch they come.  The bound is fixed when the sequence is first met: from the first streamable start, or (a
    // first batch too small to stream anything) from the largest end seen, which lies beyond every start so far.
    uint32_t guard = a.guard_until[r];
    if (a.head_guard && guard == 0 && (pa != CS_NONE || acc != 0)) {
        const uint64_t g = (uint64_t)(pa != CS_NONE ? pa : acc) + a.head_guard;
        guard = g > 0xFFFFFFFEull ? 0xFFFFFFFEu : (uint32_t)g;
        a.guard_until[r] = guard;
    }
    uint32_t H = CS_NONE, T = CS_NONE;
    if (pa != CS_NONE && pz > pa) {
        uint64_t base = pa > prev + 1 ? pa : (uint64_t)prev + 1; // entry `prev` may hold an earlier batch's -1
        if (guard > base) base = guard;
        const uint64_t h = (base + COV_CHUNK - 1) / COV_CHUNK * COV_CHUNK;
        const uint64_t lim = pz < (uint64_t)st.ref_len[r] + 1 ? pz : (uint64_t)st.ref_len[r] + 1;
        const uint64_t tt = lim / COV_CHUNK * COV_CHUNK;
        if (h < tt) {
            H = (uint32_t)h;
            T = (uint32_t)tt;
        }
    }
    a.plan_h[r] = H;
    a.plan_t[r] = T;
    a.prev_end[r] = acc > prev ? acc : prev;
    a.plan_a[r] = CS_NONE; // ready for the next batch
    a.plan_z[r] = 0;
}

// inclusive prefix sum over the 64 lanes (DPP: four row_shr steps inside the rows of 16, two row broadcasts)
__device__ __forceinline__ uint32_t st_wave_scan(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false); // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false); // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false); // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false); // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false); // row_bcast:15 -> rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false); // row_bcast:31 -> rows 2, 3
    return v;
}
__device__ __forceinline__ uint32_t st_wave_max(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, (uint32_t)__shfl_xor(v, o, 64));
    return v;
}

// 16-bit histogram counters, two per word: bin i < hw is the low half of word i, bin i >= hw the high half of
// word i - hw (neighbouring depths, which are hot together, never share a word)
__host__ __device__ inline uint32_t st_hist_words(uint32_t cov_cap) { return ((cov_cap + 2 + 1) / 2 + 3) & ~3u; }

constexpr uint32_t ST_PER_LANE = 8;              // consecutive positions per lane in one prefix pass
constexpr uint32_t ST_PASS = 64 * ST_PER_LANE;   // positions per pass
static_assert(ST_W % ST_PASS == 0, "whole passes");

// everything a full tile reads, loaded one tile ahead (the addresses depend on the tile number only)
struct StTileIn {
    int4 p;                          // pos of this lane's four records
    uint4 c;                         // their cov_end
    int32_t rf_t, ps_t, rf_n, ps_n;  // first record of the tile and of its successor
};

#ifndef ST_MIN_BLOCKS
#define ST_MIN_BLOCKS 4
#endif
__global__ __launch_bounds__(ST_THREADS, ST_MIN_BLOCKS) void k_cov_stream(DeviceState st, DeviceBatch b, CovStreamArgs a) {
    NGSQ_FOREGROUND_WAVE();
    extern __shared__ __attribute__((aligned(16))) uint32_t s_dyn[];
    const uint32_t nb = a.cov_cap + 2;
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    // wave-private window + depth histogram: no block barrier anywhere.  The histogram holds 16-bit
    // counters, two per word (st_hist_words); it is flushed before any counter can reach 2^16 (`since_flush`).
    const uint32_t hw = st_hist_words(a.cov_cap);
    uint32_t *const win = s_dyn + wave * (ST_W + hw + ST_HOT + ST_LIST);
    uint32_t *const hist = win + ST_W;
    uint32_t *const hot = hist + hw; // [ST_HOT_BINS][ST_HOT_REP]: depth hot_base + o, copy r at hot[o * ST_HOT_REP + r] (the copies of a
                                     // bin lie in neighbouring LDS banks: copy-major they shared one bank and serialised there instead)
    uint32_t *const lst = hot + ST_HOT;
    for (uint32_t i = lane; i < ST_W + hw + ST_HOT + ST_LIST; i += 64) win[i] = 0;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();

    const uint64_t n_wt = (b.n + CS_TILE - 1) / CS_TILE;
    const uint64_t n_full = b.n / CS_TILE; // tiles with all 256 records
    const uint64_t n_units = (uint64_t)gridDim.x * ST_WAVES, unit = (uint64_t)blockIdx.x * ST_WAVES + wave;
    const uint64_t per = (n_wt + n_units - 1) / n_units;
    const uint64_t t_begin = per * unit < n_wt ? per * unit : n_wt;
    const uint64_t t_end = t_begin + per < n_wt ? t_begin + per : n_wt;
    const uint32_t maxspan = *st.batch_span;

    uint32_t hot_base = 0;  // depth of the hot window's first bin (wave-uniform)
    int32_t hist_ref = -1;  // sequence the wave's histogram and bin accumulator belong to
    uint32_t since_flush = 0; // positions tallied since the histogram was last flushed (< 2^16 - ST_PASS)
    u64 zero_run = 0;         // positions of depth 0 skipped in closed form (wave-uniform)
    uint32_t lane_zero = 0;   // positions of depth 0 this lane met inside the windows
    u64 *bins = a.bin_totals;
    u64 lane_bin = 0;                           // this lane's share of the depth sum of bin `bin_q`
    uint32_t bin_q = 0, bin_p0 = 1, bin_p1 = 0; // bin_q holds the positions [bin_p0, bin_p1)
    u64 t_lo = ~0ull, t_hi = 0;                 // entries of the depth block this wave accounts for
    // the list: cov_end of every record in front of tile `lst_tile` on `lst_ref` with cov_end > lst_lo
    bool lst_valid = false;
    uint64_t lst_tile = 0;
    int32_t lst_ref = -1;
    uint32_t lst_lo = 0, lst_n = 0;
    int32_t plan_ref = -1;                      // sequence whose facts are cached in scalar registers
    uint32_t pH = CS_NONE, pT = CS_NONE, pL = 0;
    uint64_t pOff = 0;

    auto flush_bin = [&]() {
        u64 v = lane_bin;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0 && v) atomicAdd(&bins[bin_q], v);
        lane_bin = 0;
    };
    auto flush_hot = [&]() { // the copies of each hot bin -> the wave's histogram (lane = bin of the window)
        static_assert((ST_HOT_REP == 4 || ST_HOT_REP == 8) && ST_HOT_BINS <= 64, "one or two uint4 per lane");
        uint4 *const cell = reinterpret_cast<uint4 *>(hot) + (lane < ST_HOT_BINS ? lane : 0u) * (ST_HOT_REP / 4);
        uint32_t t = 0;
        if (lane < ST_HOT_BINS) {
#pragma unroll
            for (uint32_t q = 0; q < ST_HOT_REP / 4; q++) {
                const uint4 v = cell[q];
                cell[q] = make_uint4(0, 0, 0, 0);
                t += v.x + v.y + v.z + v.w;
            }
        }
        const uint32_t depth = hot_base + lane;
        const uint32_t bin = depth <= a.cov_cap ? depth : a.cov_cap + 1;
        if (t) atomicAdd(&hist[bin < hw ? bin : bin - hw], bin < hw ? t : t << 16);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    auto flush_hist = [&]() {
        if (hist_ref < 0) return;
        flush_hot();
        u64 *dst = a.hist + (u64)hist_ref * nb;
        for (uint32_t i = lane; i < hw; i += 64) {
            const uint32_t v = hist[i];
            if (v) {
                if ((v & 0xFFFFu) && i < nb) atomicAdd(&dst[i], (u64)(v & 0xFFFFu));
                if ((v >> 16) && i + hw < nb) atomicAdd(&dst[i + hw], (u64)(v >> 16));
                hist[i] = 0;
            }
        }
        {
            u64 z = lane_zero;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) z += __shfl_xor(z, o, 64);
            if (lane == 0 && zero_run + z) atomicAdd(&dst[0], zero_run + z);
        }
        zero_run = 0;
        lane_zero = 0;
        since_flush = 0;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    // One chunk-sum update per wave for the lanes that share the leader's chunk (the rule in a sorted file: at
    // 3 reads per position ten thousand records would otherwise queue on the same word); the others add on
    // their own.  Convergent: every lane calls, `active` says whether it takes part.
    auto chunk_add = [&](bool active, uint64_t chunk, uint32_t val) {
        const u64 m = __ballot(active);
        if (!m) return;
        const uint32_t c0 = __shfl((uint32_t)chunk, __ffsll((long long)m) - 1, 64);
        const bool same = active && (uint32_t)chunk == c0;
        uint32_t sum = same ? val : 0u;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
        if (lane == 0 && sum) atomicAdd(&st.chunk_sums[c0], sum);
        if (active && !same) atomicAdd(&st.chunk_sums[chunk], val);
    };
    // +1 at q0, -1 at q1 of the difference array starting at `off` (the classic path, cov_scan.hip); convergent
    auto classic = [&](bool on, uint64_t off, uint32_t q0, uint32_t q1, bool minus_in_sum_only) {
        const uint64_t g0 = off + q0, g1 = off + q1;
        if (on) {
            atomicAdd(&st.depth[g0], 1u);
            if (!minus_in_sum_only) atomicAdd(&st.depth[g1], 0xFFFFFFFFu);
            t_lo = g0 < t_lo ? g0 : t_lo;
            t_hi = g1 + 1 > t_hi ? g1 + 1 : t_hi;
        }
        chunk_add(on, g0 / COV_CHUNK, 1u);
        chunk_add(on, g1 / COV_CHUNK, 0xFFFFFFFFu);
    };
    // the parts of [sj, ej) outside the streamed range [H, T) of its sequence; convergent
    auto outside = [&](bool on, uint64_t off, uint32_t sj, uint32_t ej, uint32_t H, uint32_t T) {
        const bool head = on && sj < H;
        const uint32_t m = ej < H ? ej : H;
        if (__ballot(head)) classic(head, off, sj, m, m == H);
        const bool tl = on && T != CS_NONE && ej > T;
        if (__ballot(tl)) classic(tl, off, sj > T ? sj : T, ej, false);
    };

    auto load_full = [&](uint64_t t) -> StTileIn { // t < n_full; branch-free
        const uint64_t base = t * CS_TILE, r0 = base + (uint64_t)lane * 4;
        StTileIn in;
        in.p = *reinterpret_cast<const int4 *>(b.pos + r0);
        in.c = *reinterpret_cast<const uint4 *>(st.cov_end + r0);
        const uint64_t nx = base + CS_TILE < b.n ? base + CS_TILE : base;
        in.rf_t = b.ref_id[base];
        in.ps_t = b.pos[base];
        in.rf_n = b.ref_id[nx];
        in.ps_n = b.pos[nx];
        return in;
    };

    auto process = [&](const StTileIn &in, uint64_t t) {
        const uint64_t base = t * CS_TILE;
        const uint64_t r0 = base + (uint64_t)lane * 4;
        const uint32_t s[4] = {(uint32_t)in.p.x + 1, (uint32_t)in.p.y + 1, (uint32_t)in.p.z + 1, (uint32_t)in.p.w + 1};
        const uint32_t e[4] = {in.c.x, in.c.y, in.c.z, in.c.w};
        const int32_t rf_t = __builtin_amdgcn_readfirstlane(in.rf_t), ps_t = __builtin_amdgcn_readfirstlane(in.ps_t);
        const int32_t rf_n = __builtin_amdgcn_readfirstlane(in.rf_n), ps_n = __builtin_amdgcn_readfirstlane(in.ps_n);
        // st_valid() without its load: the sequence's facts are cached, and a load per tile would make the
        // compiler wait for everything in flight, i.e. for the prefetch of the next tile
        const bool cand = base + CS_TILE < b.n && rf_t >= 0 && (uint32_t)rf_t < st.n_refs && ps_t >= 0 && rf_n == rf_t && ps_n >= 0;
        if (cand && rf_t != plan_ref) { // the facts leave this branch in scalar registers: no wait at the join
            plan_ref = rf_t;
            pH = (uint32_t)__builtin_amdgcn_readfirstlane((int)a.plan_h[rf_t]);
            pT = (uint32_t)__builtin_amdgcn_readfirstlane((int)a.plan_t[rf_t]);
            pL = (uint32_t)__builtin_amdgcn_readfirstlane((int)st.ref_len[rf_t]);
            const uint64_t o = st.ref_depth_off[rf_t];
            pOff = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(o >> 32)) << 32) |
                   (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)o);
        }
        const bool streamable = cand && pOff != NO_DEPTH; // cand => the cache holds rf_t
        const uint32_t H = streamable ? pH : CS_NONE, T = streamable ? pT : CS_NONE, Lr = pL;
        const uint64_t off = pOff;

        // ---- classic parts of this tile's own records
        if (streamable) { // a sorted tile lies on one sequence; clamping keeps unsorted input inside the array
#pragma unroll
            for (uint32_t j = 0; j < 4; j++) {
                const uint32_t ej = e[j] < Lr + 1 ? e[j] : Lr + 1;
                const bool on = e[j] != 0 && s[j] < ej && (s[j] < H || ej > T);
                if (__ballot(on)) outside(on, off, s[j], ej, H, T); // only tiles at a seam get here
            }
        } else {
            for (uint32_t j = 0; j < 4; j++) {
                const int32_t rj = e[j] != 0 ? b.ref_id[r0 + j] : -1;
                const bool okr = rj >= 0 && (uint32_t)rj < st.n_refs;
                const uint64_t oj = okr ? st.ref_depth_off[rj] : NO_DEPTH;
                const uint32_t Lj = okr ? st.ref_len[rj] : 0u;
                const uint32_t ej = e[j] < Lj + 1 ? e[j] : Lj + 1;
                const uint32_t hj = okr ? a.plan_h[rj] : CS_NONE, tj = okr ? a.plan_t[rj] : CS_NONE;
                const bool on = oj != NO_DEPTH && s[j] < ej;
                if (__ballot(on)) outside(on, oj == NO_DEPTH ? 0 : oj, s[j], ej, hj, tj);
            }
            return;
        }
        if (H == CS_NONE) return;
        const uint32_t lo = (uint32_t)ps_t + 1 > H ? (uint32_t)ps_t + 1 : H;
        const uint32_t hi = (uint32_t)ps_n + 1 < T ? (uint32_t)ps_n + 1 : T;
        if (lo >= hi) return;

        // ---- the positions [lo, hi) are finished here
        if (hist_ref != rf_t) {
            flush_hist();
            flush_bin();
            hist_ref = rf_t;
            bins = a.bin_totals + a.bin_off[rf_t];
            bin_p0 = 1;
            bin_p1 = 0;
        }
        for (uint64_t p = ((uint64_t)lo + COV_CHUNK - 1) / COV_CHUNK * COV_CHUNK + (uint64_t)lane * COV_CHUNK; p < hi;
             p += 64ull * COV_CHUNK)
            a.chunk_flags[(off + p) / COV_CHUNK] = 1;
        t_lo = off + lo < t_lo ? off + lo : t_lo;
        t_hi = off + hi > t_hi ? off + hi : t_hi;

        bool rec[4];
#pragma unroll
        for (uint32_t j = 0; j < 4; j++) rec[j] = e[j] > lo && s[j] < hi;
        const bool from_list = lst_valid && lst_tile == t && lst_ref == rf_t && lst_lo == lo; // wave-uniform
        uint32_t new_n = 0; // the list for tile t + 1 (built in place: it never outruns the read position)
        bool new_ok = true;
        uint32_t w = lo, carry = 0;
        uint32_t lb_reach = CS_NONE; // wave-uniform: largest end among the earlier tiles' records (CS_NONE: not walked yet)
        uint32_t lb_lane = 0;        // per lane, reduced only if the owned range needs another window
        bool first = true;
        while (w < hi) {
            const uint32_t wend = hi - w > ST_W ? w + ST_W : hi;
#pragma unroll
            for (uint32_t j = 0; j < 4; j++) {
                const uint32_t sp = s[j] > lo ? s[j] : lo;
                if (rec[j] && sp >= w && sp < wend) atomicAdd(&win[sp - w], 1u);
                if (rec[j] && e[j] >= w && e[j] < wend) atomicAdd(&win[e[j] - w], 0xFFFFFFFFu);
            }
            if (lb_reach >= w && ST_EXP != 1 && ST_EXP != 4) { // earlier tiles' records that reach into [lo, hi); >=: an end exactly on w still owes its -1
                uint32_t reach = 0;
                if (from_list) {
                    if (first && lane == 0 && lst_n) atomicAdd(&win[0], lst_n); // they all start before lo
                    for (uint32_t c = 0; c < lst_n; c += 64) {
                        const uint32_t ee = c + lane < lst_n ? lst[c + lane] : 0u;
                        if (ee >= w && ee < wend && ee > lo) atomicAdd(&win[ee - w], 0xFFFFFFFFu);
                        reach = ee > reach ? ee : reach;
                    }
                } else {
                    uint32_t cnt0 = 0;
                    bool more = true; // wave-uniform
                    // one group of 64 records in front of the tile; in a sorted batch everything between the group's
                    // first record and the tile lies on one sequence, so the per-record column is read only across a
                    // sequence boundary.  The first walk also collects the ends beyond hi: the next tile's list.
                    for (uint64_t back = 64; more; back += 64) {
                        const bool okl = base + lane >= back;
                        const uint64_t idx = okl ? base + lane - back : 0;
                        const int32_t rf0 = b.ref_id[base >= back ? base - back : 0];
                        const uint32_t ss = (uint32_t)b.pos[idx] + 1, ee = st.cov_end[idx];
                        bool same = okl;
                        if (base < back || rf0 != rf_t) same = same && b.ref_id[idx] == rf_t;
                        const bool hit = same && ee > lo;
                        if (first) {
                            cnt0 += (uint32_t)__popcll(__ballot(hit));
                            const u64 km = __ballot(hit && ee > hi);
                            const uint32_t kn = (uint32_t)__popcll(km);
                            if (new_n + kn <= ST_LIST) {
                                if (hit && ee > hi) lst[new_n + __builtin_amdgcn_mbcnt_hi((uint32_t)(km >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)km, 0))] = ee;
                                new_n += kn;
                            } else {
                                new_ok = false;
                            }
                        }
                        if (hit && ee >= w && ee < wend) atomicAdd(&win[ee - w], 0xFFFFFFFFu);
                        reach = hit && ee > reach ? ee : reach;
                        const uint32_t ss0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)ss); // the group's first record
                        more = base > back && rf0 == rf_t && (uint64_t)ss0 + maxspan > lo;      // can anything in front reach?
                    }
                    if (first && lane == 0 && cnt0) atomicAdd(&win[0], cnt0); // they all start before lo
                }
                lb_lane = reach;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();

            // ---- prefix sum of the window: passes of 8, 4 or 2 consecutive positions per lane -- the widest that
            // the rest of the window fills to more than half, so that few lanes of the last pass are dead (a tile
            // of 256 reads owns ~640 positions at 60x: one pass of 512 and one of 128 instead of two of 512)
            const uint32_t len = wend - w;
            auto pass = [&](auto pl_tag, uint32_t pb) {
                constexpr uint32_t PL = decltype(pl_tag)::value;
                uint32_t x[PL];
                if constexpr (PL == 8) {
                    uint4 *cell = reinterpret_cast<uint4 *>(win + pb + lane * PL);
                    const uint4 d0 = cell[0], d1 = cell[1];
                    cell[0] = make_uint4(0, 0, 0, 0);
                    cell[1] = make_uint4(0, 0, 0, 0);
                    x[0] = d0.x, x[1] = d0.y, x[2] = d0.z, x[3] = d0.w, x[4] = d1.x, x[5] = d1.y, x[6] = d1.z, x[7] = d1.w;
                } else if constexpr (PL == 4) {
                    uint4 *cell = reinterpret_cast<uint4 *>(win + pb + lane * PL);
                    const uint4 d0 = cell[0];
                    cell[0] = make_uint4(0, 0, 0, 0);
                    x[0] = d0.x, x[1] = d0.y, x[2] = d0.z, x[3] = d0.w;
                } else {
                    uint2 *cell = reinterpret_cast<uint2 *>(win + pb + lane * PL);
                    const uint2 d0 = cell[0];
                    cell[0] = make_uint2(0, 0);
                    x[0] = d0.x, x[1] = d0.y;
                }
#pragma unroll
                for (uint32_t k = 1; k < PL; k++) x[k] += x[k - 1];
                // the hot window follows the running depth: keep the depth at the start of the pass off its outer eighths
                if (carry - hot_base - ST_HOT_BINS / 8 >= 3 * ST_HOT_BINS / 4) { // wave-uniform
                    const uint32_t nb0 = carry > ST_HOT_BINS / 2 ? carry - ST_HOT_BINS / 2 : 0u;
                    if (nb0 != hot_base) { // (shallow stretches keep the window at depth 0)
                        flush_hot();
                        hot_base = nb0;
                    }
                }
                const uint32_t hot_lane = (uint32_t)(hot - hist) + (lane & (ST_HOT_REP - 1)) - hot_base * ST_HOT_REP;
                const uint32_t inc = st_wave_scan(x[PL - 1]);
                const uint32_t before = carry + inc - x[PL - 1];
                carry += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
                const uint32_t p_first = w + pb, p_last = (len - pb > 64 * PL ? p_first + 64 * PL - 1 : wend - 1);
                const bool one_bin = p_first >= bin_p0 && p_last < bin_p1; // wave-uniform
                if (!one_bin) { // coverage.rs:206-230: position i >= 1 is in bin 1 + (i-1)/bin_size
                    const uint32_t q0 = (p_first - 1) / a.bin_size, q1 = (p_last - 1) / a.bin_size;
                    if (q0 == q1) {
                        flush_bin();
                        bin_q = q0 + 1;
                        bin_p0 = q0 * a.bin_size + 1;
                        const uint64_t top = (uint64_t)(q0 + 1) * a.bin_size + 1;
                        bin_p1 = top > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)top;
                    }
                }
                const bool in_acc = p_first >= bin_p0 && p_last < bin_p1;
                uint32_t lsum = 0;
#pragma unroll
                for (uint32_t k = 0; k < PL; k++) { // branch-free: positions past the end add 0
                    const uint32_t i = pb + lane * PL + k;
                    const uint32_t depth = before + x[k];
                    const uint32_t bin = depth <= a.cov_cap ? depth : a.cov_cap + 1;
                    // Depth 0 is counted in a register (targeted sequencing leaves most positions uncovered: every
                    // lane would queue on bin 0).  A dead or zero lane adds 0 to a word of its own: the same word
                    // for all of them would serialise just the same.
                    lane_zero += i < len && depth == 0;
                    const uint32_t one = i < len && depth != 0 ? 1u : 0u;
                    const bool in_hot = depth - hot_base < ST_HOT_BINS;
                    const uint32_t word = !one ? lane : in_hot ? hot_lane + depth * ST_HOT_REP : bin < hw ? bin : bin - hw;
                    if (ST_EXP != 3) atomicAdd(&hist[word], in_hot || bin < hw ? one : one << 16);
                    lsum += i < len ? depth : 0u;
                }
                if (in_acc) {
                    lane_bin += lsum;
                } else { // the pass straddles a bin boundary (once per bin_size positions): position by position
                    for (uint32_t k = 0; k < PL; k++) {
                        const uint32_t i = pb + lane * PL + k;
                        const uint32_t depth = before + x[k];
                        if (i < len && depth) atomicAdd(&bins[(u64)(w + i - 1) / a.bin_size + 1], (u64)depth);
                    }
                }
                since_flush += 64 * PL;
            };
            for (uint32_t pb = 0; pb < len && ST_EXP != 2 && ST_EXP != 4;) {
                const uint32_t rem = len - pb;
                if (rem > 256) {
                    pass(std::integral_constant<uint32_t, 8>{}, pb);
                    pb += 512;
                } else if (rem > 128) {
                    pass(std::integral_constant<uint32_t, 4>{}, pb);
                    pb += 256;
                } else {
                    pass(std::integral_constant<uint32_t, 2>{}, pb);
                    pb += 128;
                }
            }
            if (since_flush > 0xFFFFu - 2 * ST_PASS) flush_hist(); // wave-uniform
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            first = false;
            w = wend;
            if (w < hi) { // jump over positions nothing covers
                lb_reach = st_wave_max(lb_lane);
                uint32_t reach = lb_reach, nxt = 0xFFFFFFFFu - hi;
#pragma unroll
                for (uint32_t j = 0; j < 4; j++) {
                    if (rec[j] && s[j] < w) reach = max(reach, e[j]);
                    if (rec[j] && s[j] >= w) nxt = max(nxt, 0xFFFFFFFFu - s[j]);
                }
                reach = st_wave_max(reach);
                nxt = 0xFFFFFFFFu - st_wave_max(nxt); // the smallest start >= w (hi if none)
                if (reach <= w && nxt > w) {
                    zero_run += nxt - w;
                    carry = 0;
                    w = nxt;
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }

        // ---- the list for the next tile: what was open in front and still is beyond hi, plus this tile's records
        if (ST_EXP == 1 || ST_EXP == 4) return;
        if (from_list) {
            for (uint32_t c = 0; c < lst_n; c += 64) {
                const uint32_t ee = c + lane < lst_n ? lst[c + lane] : 0u;
                const u64 km = __ballot(ee > hi);
                if (ee > hi) lst[new_n + __builtin_amdgcn_mbcnt_hi((uint32_t)(km >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)km, 0))] = ee;
                new_n += (uint32_t)__popcll(km);
            }
        }
#pragma unroll
        for (uint32_t j = 0; j < 4; j++) {
            const u64 km = __ballot(e[j] > hi);
            const uint32_t kn = (uint32_t)__popcll(km);
            if (new_ok && new_n + kn <= ST_LIST) {
                if (e[j] > hi) lst[new_n + __builtin_amdgcn_mbcnt_hi((uint32_t)(km >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)km, 0))] = e[j];
                new_n += kn;
            } else {
                new_ok = false;
            }
        }
        lst_valid = new_ok;
        lst_n = new_n;
        lst_tile = t + 1;
        lst_ref = rf_t;
        lst_lo = hi;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };

    // ---- full tiles: the next tile's inputs are in flight while this one is processed (two tiles ahead was
    // measured: no faster, and the extra registers spill)
    uint64_t t = t_begin;
    const uint64_t t_full_end = t_end < n_full ? t_end : n_full;
    if (t < t_full_end) {
        StTileIn cur = load_full(t);
        while (t < t_full_end) {
            const uint64_t tn = t + 1;
            const StTileIn nxt = load_full(tn < n_full ? tn : t); // past the end: a re-read
            process(cur, t);
            cur = nxt;
            t = tn;
        }
    }
    if (n_full < n_wt && t_begin <= n_full && n_full < t_end) { // the partial last tile: element by element, never streamable
        const uint64_t base = n_full * CS_TILE, r0 = base + (uint64_t)lane * 4;
        int32_t p[4];
        uint32_t c[4];
#pragma unroll
        for (uint32_t j = 0; j < 4; j++) {
            const bool ok = r0 + j < b.n;
            p[j] = ok ? b.pos[r0 + j] : -1;
            c[j] = ok ? st.cov_end[r0 + j] : 0u;
        }
        StTileIn in;
        in.p = make_int4(p[0], p[1], p[2], p[3]);
        in.c = make_uint4(c[0], c[1], c[2], c[3]);
        in.rf_t = in.rf_n = b.ref_id[base];
        in.ps_t = in.ps_n = b.pos[base];
        process(in, n_full);
    }
    flush_hist();
    flush_bin();
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const u64 l2 = __shfl_xor(t_lo, o, 64), h2 = __shfl_xor(t_hi, o, 64);
        t_lo = l2 < t_lo ? l2 : t_lo;
        t_hi = h2 > t_hi ? h2 : t_hi;
    }
    if (lane == 0 && t_hi > 0) {
        atomicMin(&st.touched[0], t_lo);
        atomicMax(&st.touched[1], t_hi);
    }
}

hipError_t launch_cov_stream(const LaunchInfo &li, const DeviceState &st, const DeviceBatch &b, const CovStreamArgs &a,
                             hipStream_t s) {
    if (!b.n) return hipSuccess;
    const uint64_t n_wt = (b.n + CS_TILE - 1) / CS_TILE;
    hipLaunchKernelGGL(k_cov_plan_tiles, dim3((uint32_t)((n_wt + 255) / 256)), dim3(256), 0, s, st, b, a);
    hipLaunchKernelGGL(k_cov_plan_refs, dim3((st.n_refs + 255) / 256), dim3(256), 0, s, st, a);
    const size_t lds = (size_t)ST_WAVES * (ST_W + st_hist_words(a.cov_cap) + ST_HOT + ST_LIST) * sizeof(uint32_t);
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_cov_stream),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
        if (e != hipSuccess) return e;
        attr = true;
    }
    uint64_t g = (n_wt + ST_WAVES - 1) / ST_WAVES;
    const uint64_t per_cu = (160 * 1024 - 1024) / lds; // blocks that fit the 160 KB of LDS of one CU
    const uint64_t cap = (uint64_t)li.n_cu * (per_cu < 1 ? 1 : per_cu > 6 ? 6 : per_cu);
    if (g > cap) g = cap;
    hipLaunchKernelGGL(k_cov_stream, dim3((uint32_t)g), dim3(ST_THREADS), lds, s, st, b, a);
    return hipGetLastError();
}

} // namespace ngsq
