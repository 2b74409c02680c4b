// features_kernel.hip -- Genomic Features facet (gfx950).
// reference: src/qc/record_based/features.rs:115-242 (process), :270-355 (interval stores)
//
// The reference keeps two rust_lapper stores per sequence (UTR/CDS features; gene/exon features)
// and, per record, walks `find(start, end + 1)`.  All it takes from the walk is, per feature NAME,
// how many overlapping intervals there are (capped by the few flags it sets).  For half-open
// intervals [s, e) with s <= e and a query [qs, qe) with qs < qe
//     #overlapping = #{ s < qe } - #{ e <= qs }
// (an interval fails `s < qe && e > qs` either by s >= qe or by e <= qs, never both), so two binary
// searches per name on the separately sorted starts and stops give the EXACT count -- no interval
// tree, no max-length scan: 10 searches per record over L2-resident lists.
//
// The 256 records a block takes at a time are neighbours in a coordinate-sorted file: their query bounds span a
// few hundred positions, between which hardly any interval starts or ends.  Twenty threads first search the lists
// for the tile's smallest and largest bounds (per name: starts at the smallest / largest query end, stops at the
// smallest / largest query start); every record's own searches then run inside those brackets -- one or two steps
// instead of eighteen.  Records on another sequence than the tile's first (and any order of records) stay
// correct: their brackets are the whole lists.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "kernels.h"

namespace ngsq {

namespace {

__device__ __forceinline__ uint32_t lower_bound(const uint32_t *a, uint32_t lo, uint32_t hi, uint32_t v) {
    // first index in [lo, hi) with a[i] >= v
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (a[mid] < v) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}
__device__ __forceinline__ uint32_t upper_bound(const uint32_t *a, uint32_t lo, uint32_t hi, uint32_t v) {
    // first index in [lo, hi) with a[i] > v
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (a[mid] <= v) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

// The same two searches started from where the previous tile's search ended: in a coordinate-sorted file the tile's bounds
// move on by a few hundred positions, i.e. by zero to two list entries, and a search of eighteen dependent L2 loads
// (a list holds ~10^5 entries) -- made by twenty threads while the rest of the block waits at a barrier -- was what a
// tile's time consisted of.  `from` is only a hint: bounds that moved backwards (unsorted input, another shape of read)
// are found by a search of the part in front of it.
template <bool UPPER>
__device__ __forceinline__ uint32_t bound_from(const uint32_t *a, uint32_t lo, uint32_t hi, uint32_t from, uint32_t v) {
    auto below = [&](uint32_t x) { return UPPER ? x <= v : x < v; }; // the entries in front of the answer
    if (from > hi) from = hi;
    if (from < lo) from = lo;
    if (from > lo && !below(a[from - 1])) return UPPER ? upper_bound(a, lo, from, v) : lower_bound(a, lo, from, v);
    uint32_t at = from, step = 1;
    while (at < hi && below(a[at])) { // gallop: from, from + 1, from + 3, from + 7, ...
        at += step;
        step <<= 1;
    }
    const uint32_t h = at < hi ? at : hi;
    const uint32_t l = step > 1 ? at - (step >> 1) + 1 : from; // behind the last entry seen below the bound
    return UPPER ? upper_bound(a, l < h ? l : h, h, v) : lower_bound(a, l < h ? l : h, h, v);
}

// intervals of name id k on sequence r overlapping [qs, qe); br = brackets of the tile for this name, or null
__device__ __forceinline__ uint32_t count_overlaps(const FeatureTables &ft, uint32_t k, uint32_t r, uint32_t qs, uint32_t qe,
                                                   const uint32_t *br) {
    if (br) {
        const uint32_t started = lower_bound(ft.starts, br[0], br[1], qe); // s < qe
        const uint32_t ended = upper_bound(ft.stops, br[2], br[3], qs);    // e <= qs
        return started - ended;
    }
    const uint32_t b = ft.idx[k * ft.n_refs + r], e = ft.idx[k * ft.n_refs + r + 1];
    if (b == e) return 0;
    const uint32_t started = lower_bound(ft.starts, b, e, qe) - b; // s < qe
    const uint32_t ended = upper_bound(ft.stops, b, e, qs) - b;    // e <= qs
    return started - ended;
}

// per-wave count kept in a (uniform) register over the block's whole loop: one global atomic per counter and
// BLOCK at the end -- an atomic per wave and tile on the same eleven addresses serialised at the L2 (~9 ns each)
// and was what this kernel's time consisted of
__device__ __forceinline__ void tally(uint32_t &counter, bool pred) { counter += (uint32_t)__popcll(__ballot(pred)); }

} // namespace

// A TILE is F_RPT x 64 consecutive records of ONE WAVE, F_RPT per lane (record r * 64 + lane of the tile: coalesced columns).
// Round 4: until then a tile was a block's (F_RPT x 256 records, three barriers, the tile's bounds through LDS atomics, the
// twenty bracket searches by twenty threads while 236 waited at a barrier).  A wave does all of it by itself now -- the bounds
// by cross-lane reductions, the searches by its lanes 0..19, which keep their own previous result in a register as the next
// tile's hint, the brackets handed to the records' searches as SCALARS (v_readlane: the names are uniform) -- so no wave ever
// waits for another one's L2 probes, and the kernel has no LDS traffic and no barrier inside its loop.
#ifndef NGSQ_FEATURES_RPT
#define NGSQ_FEATURES_RPT 4
#endif
constexpr uint32_t F_RPT = NGSQ_FEATURES_RPT;
__global__ __launch_bounds__(256) void k_features(DeviceState st, DeviceBatch b, FeatureTables ft) {
    NGSQ_FOREGROUND_WAVE();
    __shared__ uint32_t s_cnt[11];
    if (threadIdx.x < 11) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t cnt[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    // a wave takes CONSECUTIVE tiles (of a sorted file: consecutive positions), so that a tile's searches can start from
    // the previous tile's
    constexpr uint64_t TILE = 64ull * F_RPT;
    const uint64_t n_waves = (uint64_t)gridDim.x * (blockDim.x >> 6), wave_id = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint64_t per = ((b.n + n_waves - 1) / n_waves + TILE - 1) / TILE * TILE;
    const uint64_t lo_i = min(wave_id * per, b.n), hi_i = lo_i + per < b.n ? lo_i + per : b.n;
    int32_t prev_ref = -1; // the sequence `my_br` belongs to (-1: none)
    uint32_t my_br = 0;    // lanes 0..19: bracket `lane & 3` of name id `lane >> 2` of the previous tile, relative to the list's begin
    uint32_t my_lo = 0;    // ... and the begin of that list
    for (uint64_t t0 = lo_i; t0 < hi_i; t0 += TILE) {
        uint32_t qs[F_RPT], qe[F_RPT], what[F_RPT]; // what: 0 nothing (past the end), 1 ignored flags, 2 error: reference, 3 ignored: not primary, 4 error: position, 5 looked up
        int32_t ref[F_RPT];
        {
            // every column of the tile's records first, unconditionally (a record past the end reads the wave's first), then
            // what depends on them: the loads of a stage are in flight together -- one record after the other, each behind its
            // own branches, the tile waited out a dozen memory latencies here
            uint32_t flag[F_RPT], n_ops[F_RPT], op0[F_RPT];
            int32_t pos[F_RPT];
            uint64_t c0[F_RPT], idx[F_RPT];
            uint8_t prim[F_RPT];
#pragma unroll
            for (uint32_t r = 0; r < F_RPT; r++) {
                const uint64_t i = t0 + r * 64 + lane;
                idx[r] = i < hi_i ? i : lo_i;
                flag[r] = b.flag[idx[r]];
                ref[r] = b.ref_id[idx[r]];
                pos[r] = b.pos[idx[r]];
                n_ops[r] = batch_n_ops(b, idx[r]);
                c0[r] = b.cigar_off ? b.cigar_off[idx[r]] : idx[r] * b.cigar_stride;
            }
#pragma unroll
            for (uint32_t r = 0; r < F_RPT; r++) {
                op0[r] = b.cigar[c0[r]]; // (the column has slack behind its last operation)
                prim[r] = ref[r] >= 0 && (uint32_t)ref[r] < ft.n_refs ? ft.primary[ref[r]] : (uint8_t)0;
            }
#pragma unroll
            for (uint32_t r = 0; r < F_RPT; r++) {
                const uint64_t i = t0 + r * 64 + lane;
                qs[r] = qe[r] = what[r] = 0;
                if (i >= hi_i) {
                    ref[r] = -1;
                } else if (flag[r] & 0x4u) { // features.rs:127-130
                    what[r] = 1;
                } else if (ref[r] < 0 || (uint32_t)ref[r] >= ft.n_refs) { // :132-155
                    what[r] = 2;
                } else if (!prim[r]) { // :157-165
                    what[r] = 3;
                } else if (pos[r] < 0) { // :171-174
                    what[r] = 4;
                } else {
                    // :176-178  start = alignment_start (1-based), end = start + cigar.alignment_span()
                    uint32_t span = 0;
                    for (uint32_t k = 0; k < n_ops[r]; k++) {
                        const uint32_t op = k ? b.cigar[c0[r] + k] : op0[r], code = op & 15u;
                        // M, D, N, =, X consume the reference
                        if (code == 0 || code == 2 || code == 3 || code == 7 || code == 8) span += op >> 4;
                    }
                    qs[r] = (uint32_t)pos[r] + 1u;
                    qe[r] = qs[r] + span + 1u; // find(start, end + 1)
                    what[r] = 5;
                }
            }
        }
        // the tile's first (sequence, start) among the records that are looked up
        unsigned long long key = ~0ull;
#pragma unroll
        for (uint32_t r = 0; r < F_RPT; r++) {
            const unsigned long long kr = what[r] == 5 ? (unsigned long long)(uint32_t)ref[r] << 32 | qs[r] : ~0ull;
            key = kr < key ? kr : key;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long other = __shfl_xor(key, o, 64);
            key = other < key ? other : key;
        }
        const int32_t r0 = key == ~0ull ? -1 : (int32_t)(key >> 32);
        // on that sequence: largest query start, smallest / largest query end
        uint32_t m0 = 0u, m1 = 0xFFFFFFFFu, m2 = 0u;
#pragma unroll
        for (uint32_t r = 0; r < F_RPT; r++)
            if (what[r] == 5 && ref[r] == r0) {
                m0 = max(m0, qs[r]);
                m1 = min(m1, qe[r]);
                m2 = max(m2, qe[r]);
            }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            m0 = max(m0, (uint32_t)__shfl_xor((int)m0, o, 64));
            m1 = min(m1, (uint32_t)__shfl_xor((int)m1, o, 64));
            m2 = max(m2, (uint32_t)__shfl_xor((int)m2, o, 64));
        }
        if (r0 >= 0 && lane < 20) {
            const uint32_t k = lane >> 2, which = lane & 3;
            const uint32_t lo = ft.idx[k * ft.n_refs + r0], hi = ft.idx[k * ft.n_refs + r0 + 1];
            const uint32_t from = prev_ref == r0 ? lo + my_br : lo; // (this lane's own result for the previous tile)
            uint32_t v;
            if (which == 0) v = bound_from<false>(ft.starts, lo, hi, from, m1) - lo;                         // fewest starts below a query end
            else if (which == 1) v = bound_from<false>(ft.starts, lo, hi, from, m2) - lo;                    // most
            else if (which == 2) v = bound_from<true>(ft.stops, lo, hi, from, (uint32_t)(key & 0xFFFFFFFFu)) - lo; // fewest stops at or below a query start
            else v = bound_from<true>(ft.stops, lo, hi, from, m0) - lo;                                      // most
            // kept relative to the list's begin, so that counts come out directly; the searches add the begin back
            my_br = v;
            my_lo = lo;
        }
        prev_ref = r0;
#pragma unroll
        for (uint32_t r = 0; r < F_RPT; r++) {
            bool utr5 = false, utr3 = false, cds = false, intergenic = false, exonic = false, intronic = false;
            const bool look = what[r] == 5;
            if (look) {
                const bool narrow = ref[r] == r0;
                auto count = [&](uint32_t name) -> uint32_t {
                    if (!narrow) return count_overlaps(ft, name, (uint32_t)ref[r], qs[r], qe[r], nullptr);
                    // (brackets that have closed -- no interval of this name begins or ends inside the tile's span, the usual
                    // case -- give the count without touching the lists: lower_bound / upper_bound of an empty range)
                    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)my_lo, (int)(4u * name));
                    const uint32_t br[4] = {lo + (uint32_t)__builtin_amdgcn_readlane((int)my_br, (int)(4u * name)),
                                            lo + (uint32_t)__builtin_amdgcn_readlane((int)my_br, (int)(4u * name + 1u)),
                                            lo + (uint32_t)__builtin_amdgcn_readlane((int)my_br, (int)(4u * name + 2u)),
                                            lo + (uint32_t)__builtin_amdgcn_readlane((int)my_br, (int)(4u * name + 3u))};
                    return count_overlaps(ft, name, (uint32_t)ref[r], qs[r], qe[r], br);
                };
                // :186-214  UTR / CDS store: the if / else-if chain over the overlapping intervals only
                // depends on how many there are of each name (roles may share a name)
                bool c5 = false, c3 = false, cc = false;
                const uint32_t n5 = ft.role_name[NGSQ_ROLE_FIVE_PRIME_UTR], n3 = ft.role_name[NGSQ_ROLE_THREE_PRIME_UTR],
                               nc = ft.role_name[NGSQ_ROLE_CODING_SEQUENCE];
#pragma unroll
                for (uint32_t role = 0; role < 3; role++) {
                    const uint32_t name = ft.role_name[role];
                    bool seen = false; // count each distinct name once
                    for (uint32_t q = 0; q < role; q++) seen |= ft.role_name[q] == name;
                    if (seen) continue;
                    uint32_t c = min(count(name), 3u);
                    for (; c; c--) {
                        if (!c5 && name == n5) c5 = true;
                        else if (!c3 && name == n3) c3 = true;
                        else if (!cc && name == nc) cc = true;
                    }
                }
                utr5 = c5, utr3 = c3, cds = cc;
                // :216-238  gene / exon store.  A name that is also a UTR/CDS name never reaches this
                // store (:322-333), and `name == gene` is tested before `name == exon`.
                const uint32_t ne = ft.role_name[NGSQ_ROLE_EXON], ng = ft.role_name[NGSQ_ROLE_GENE];
                const bool gene_in_store = ng != n5 && ng != n3 && ng != nc;
                const bool exon_in_store = ne != n5 && ne != n3 && ne != nc && ne != ng;
                const bool has_gene = gene_in_store && count(ng) > 0;
                const bool has_exon = exon_in_store && count(ne) > 0;
                if (has_gene) {
                    if (has_exon) exonic = true;
                    else intronic = true;
                } else {
                    intergenic = true;
                }
            }
            tally(cnt[F_UTR5], utr5);
            tally(cnt[F_UTR3], utr3);
            tally(cnt[F_CDS], cds);
            tally(cnt[F_INTERGENIC], intergenic);
            tally(cnt[F_EXONIC], exonic);
            tally(cnt[F_INTRONIC], intronic);
            tally(cnt[F_PROCESSED], look); // :240
            tally(cnt[F_IGN_FLAGS], what[r] == 1);
            tally(cnt[F_IGN_NONPRIMARY], what[r] == 3);
            tally(cnt[9], what[r] == 2);
            tally(cnt[10], what[r] == 4);
        }
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0)
        for (int k = 0; k < 11; k++)
            if (cnt[k]) atomicAdd(&s_cnt[k], cnt[k]);
    __syncthreads();
    // Two thousand blocks adding to the same eleven words at their end serialise at the L2 (~9 ns each: 0.1 of this kernel's
    // 0.33 ms).  They add to one of FT_SLOTS copies instead, and the block that finishes last -- a ticket tells it -- sums the
    // copies into the counters and leaves the scratch zero for the next launch.
    __shared__ bool s_last;
    // (no __threadfence: on gfx950 an agent-scope release is a write-back of the XCD's L2, and eight thousand waves doing one
    // cost more than the serialised atomics did.  The tallies are device-scope atomics whose RESULT is waited for, so they have
    // been performed when the barrier lets the ticket go; the last block reads them with atomics too.)
    if (threadIdx.x < 11 && s_cnt[threadIdx.x]) {
        const unsigned long long before = atomicAdd(&ft.scratch[(blockIdx.x % FT_SLOTS) * 16 + threadIdx.x], (unsigned long long)s_cnt[threadIdx.x]);
        asm volatile("" ::"v"((uint32_t)before));
    }
    __syncthreads();
    if (threadIdx.x == 0) s_last = atomicAdd(&ft.scratch[FT_SLOTS * 16], 1ull) == gridDim.x - 1;
    __syncthreads();
    if (!s_last) return;
    if (threadIdx.x < 11) {
        unsigned long long sum = 0;
        for (uint32_t k = 0; k < FT_SLOTS; k++) sum += atomicExch(&ft.scratch[k * 16 + threadIdx.x], 0ull);
        unsigned long long *dst = threadIdx.x < 9 ? st.counters + C_FEAT + threadIdx.x
                                                  : st.counters + (threadIdx.x == 9 ? C_FEAT_ERR_REF : C_FEAT_ERR_POS);
        if (sum) atomicAdd(dst, sum);
    }
    if (threadIdx.x == 0) ft.scratch[FT_SLOTS * 16] = 0;
}

hipError_t launch_features(const LaunchInfo &li, const DeviceState &st, const DeviceBatch &b, const FeatureTables &ft,
                           hipStream_t s) {
    if (!b.n) return hipSuccess;
    // Blocks per CU (each takes a contiguous slice of the batch), measured on 100 M records (same box, ms): 5 -> 1.99, 6 -> 1.77,
    // 7 (what is resident at once) -> 2.33, 8 -> 2.13 (a second round with one block per CU), 9 -> 2.02, 10 -> 1.91,
    // 11 -> 1.78, 12 -> 1.73, 14 -> 1.85, 20 -> 1.78, 24 -> 1.76; the same order on 10 M.  Seven resident blocks are slower
    // than six (28 waves per CU wait on each other's L2 probes); a grid of nearly two rounds evens out the slices that meet
    // long gene lists.  NGSQ_FEATURES_BLOCKS_PER_CU: measurement aid.
    static int per_cu = 0;
    if (!per_cu) {
        const char *e = getenv("NGSQ_FEATURES_BLOCKS_PER_CU");
        per_cu = e && atoi(e) > 0 ? atoi(e) : 12;
    }
    uint64_t g = (b.n + 256 * F_RPT - 1) / (256 * F_RPT); // four waves per block, a tile of 64 x F_RPT records each
    if (g > (uint64_t)li.n_cu * (uint64_t)per_cu) g = (uint64_t)li.n_cu * (uint64_t)per_cu;
    hipLaunchKernelGGL(k_features, dim3((uint32_t)g), dim3(256), 0, s, st, b, ft);
    return hipGetLastError();
}

} // namespace ngsq
