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

#include <algorithm>

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
    // the usual answer is `from` itself (the entry in front of it below the bound, the one at it not): both are requested at once
    // (a value that decides nothing where there is no such entry)
    const uint32_t before = from > lo ? a[from - 1] : 0u, here = from < hi ? a[from] : 0u;
    if (from > lo && !below(before)) return UPPER ? upper_bound(a, lo, from, v) : lower_bound(a, lo, from, v);
    if (from >= hi || !below(here)) return from;
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

// wave-wide minimum / maximum without the LDS pipe: prefix steps inside the rows of sixteen lanes (row_shr), then across them
// (row_bcast); lane 63 holds the result.  (Five of them per tile: as __shfl_xor butterflies they were thirty ds_bpermute.)
template <bool MAX>
__device__ __forceinline__ uint32_t wave_extreme(uint32_t v) {
    constexpr int ID = MAX ? 0 : -1; // what a lane without a source contributes
    auto pick = [](uint32_t a, uint32_t b) { return MAX ? (a > b ? a : b) : (a < b ? a : b); };
    v = pick(v, (uint32_t)__builtin_amdgcn_update_dpp(ID, (int)v, 0x111, 0xF, 0xF, false)); // row_shr:1
    v = pick(v, (uint32_t)__builtin_amdgcn_update_dpp(ID, (int)v, 0x112, 0xF, 0xF, false)); // row_shr:2
    v = pick(v, (uint32_t)__builtin_amdgcn_update_dpp(ID, (int)v, 0x114, 0xF, 0xF, false)); // row_shr:4
    v = pick(v, (uint32_t)__builtin_amdgcn_update_dpp(ID, (int)v, 0x118, 0xF, 0xF, false)); // row_shr:8
    v = pick(v, (uint32_t)__builtin_amdgcn_update_dpp(ID, (int)v, 0x142, 0xA, 0xF, false)); // row_bcast:15
    v = pick(v, (uint32_t)__builtin_amdgcn_update_dpp(ID, (int)v, 0x143, 0xC, 0xF, false)); // row_bcast:31
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

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
    uint32_t my_lo = 0, my_hi = 0; // ... and the begin and the end of that list
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
            }
            // `primary` of the records' sequences: one (uniform) load for the sequence of the wave's first record, a load of its own
            // only for a record on another one
            const int32_t g = __builtin_amdgcn_readfirstlane(ref[0]);
            const uint8_t prim_g = g >= 0 && (uint32_t)g < ft.n_refs ? ft.primary[g] : (uint8_t)0;
#pragma unroll
            for (uint32_t r = 0; r < F_RPT; r++)
                prim[r] = ref[r] == g ? prim_g : ref[r] >= 0 && (uint32_t)ref[r] < ft.n_refs ? ft.primary[ref[r]] : (uint8_t)0;
            // (selects, not branches: every divergent `if` costs the scalar unit -- the busier one here -- half a dozen instructions)
#pragma unroll
            for (uint32_t r = 0; r < F_RPT; r++) {
                const uint64_t i = t0 + r * 64 + lane;
                const bool live = i < hi_i, valid_ref = ref[r] >= 0 && (uint32_t)ref[r] < ft.n_refs;
                // features.rs:127-130 unmapped, :132-155 no such sequence, :157-165 not a primary one, :171-174 no position
                what[r] = !live ? 0u : (flag[r] & 0x4u) ? 1u : !valid_ref ? 2u : !prim[r] ? 3u : pos[r] < 0 ? 4u : 5u;
                if (!live) ref[r] = -1;
                // :176-178  start = alignment_start (1-based), end = start + cigar.alignment_span(): M, D, N, =, X consume the reference
                uint32_t span = n_ops[r] && ((0x18Du >> (op0[r] & 15u)) & 1u) ? op0[r] >> 4 : 0u;
                if (__ballot(what[r] == 5 && n_ops[r] > 1)) { // (uniform: a batch of one-operation CIGARs never gets here)
                    if (what[r] == 5)
                        for (uint32_t k = 1; k < n_ops[r]; k++) {
                            const uint32_t op = b.cigar[c0[r] + k];
                            if ((0x18Du >> (op & 15u)) & 1u) span += op >> 4;
                        }
                }
                qs[r] = (uint32_t)pos[r] + 1u;
                qe[r] = qs[r] + span + 1u; // find(start, end + 1)  (both only looked at where what == 5)
            }
        }
        // the tile's first (sequence, start) among the records that are looked up: the smallest sequence, then the smallest start on it
        uint32_t fr = 0xFFFFFFFFu;
#pragma unroll
        for (uint32_t r = 0; r < F_RPT; r++)
            if (what[r] == 5) fr = min(fr, (uint32_t)ref[r]);
        fr = wave_extreme<false>(fr);
        const int32_t r0 = fr == 0xFFFFFFFFu ? -1 : (int32_t)fr;
        // on that sequence: smallest / largest query start, smallest / largest query end
        uint32_t q0 = 0xFFFFFFFFu, m0 = 0u, m1 = 0xFFFFFFFFu, m2 = 0u;
#pragma unroll
        for (uint32_t r = 0; r < F_RPT; r++)
            if (what[r] == 5 && ref[r] == r0) {
                q0 = min(q0, qs[r]);
                m0 = max(m0, qs[r]);
                m1 = min(m1, qe[r]);
                m2 = max(m2, qe[r]);
            }
        if (r0 >= 0) { // (uniform)
            q0 = wave_extreme<false>(q0);
            m0 = wave_extreme<true>(m0);
            m1 = wave_extreme<false>(m1);
            m2 = wave_extreme<true>(m2);
        }
        if (r0 >= 0 && lane < 20) {
            const uint32_t k = lane >> 2, which = lane & 3;
            // (the list's ends: kept from the previous tile while the sequence stays the same -- a dependent load less per tile)
            if (prev_ref != r0) {
                my_lo = ft.idx[k * ft.n_refs + r0];
                my_hi = ft.idx[k * ft.n_refs + r0 + 1];
            }
            const uint32_t lo = my_lo, hi = my_hi;
            const uint32_t from = prev_ref == r0 ? lo + my_br : lo; // (this lane's own result for the previous tile)
            uint32_t v;
            if (which == 0) v = bound_from<false>(ft.starts, lo, hi, from, m1) - lo;                         // fewest starts below a query end
            else if (which == 1) v = bound_from<false>(ft.starts, lo, hi, from, m2) - lo;                    // most
            else if (which == 2) v = bound_from<true>(ft.stops, lo, hi, from, q0) - lo;                      // fewest stops at or below a query start
            else v = bound_from<true>(ft.stops, lo, hi, from, m0) - lo;                                      // most
            // kept relative to the list's begin, so that counts come out directly; the searches add the begin back
            my_br = v;
        }
        prev_ref = r0;
        // ---- the five role names' overlap counts of every record, two bits each (0..3: the chains below never ask for more).
        // Round 5: a role at a time for all F_RPT records of the lane, its brackets taken out of the bracket lanes ONCE (they are
        // uniform: scalar registers), and where a bracket holds at most two list entries -- nearly always: a tile spans a few hundred
        // positions, a list has an entry every few thousand -- those entries are loaded once per tile (uniform addresses) and a
        // record's count is two compares against each: no search, no loop, no load per record.  (Until then every record ran its ten
        // searches inside the brackets, each with its own loop set-up: 300 vector instructions per record and lane.)
        uint32_t acc[F_RPT];
#pragma unroll
        for (uint32_t r = 0; r < F_RPT; r++) acc[r] = 0;
#pragma unroll 1
        for (uint32_t role = 0; role < 5; role++) {
            const uint32_t name = ft.role_name[role];
            uint32_t lo = 0, b0 = 0, b1 = 0, b2 = 0, b3 = 0;
            if (r0 >= 0) {
                lo = (uint32_t)__builtin_amdgcn_readlane((int)my_lo, (int)(4u * name));
                b0 = (uint32_t)__builtin_amdgcn_readlane((int)my_br, (int)(4u * name));
                b1 = (uint32_t)__builtin_amdgcn_readlane((int)my_br, (int)(4u * name + 1u));
                b2 = (uint32_t)__builtin_amdgcn_readlane((int)my_br, (int)(4u * name + 2u));
                b3 = (uint32_t)__builtin_amdgcn_readlane((int)my_br, (int)(4u * name + 3u));
            }
            const bool few = b1 - b0 <= 2u && b3 - b2 <= 2u; // (uniform)
            // the bracket's entries, or a value no query bound is above
            uint32_t s0 = 0xFFFFFFFFu, s1 = 0xFFFFFFFFu, e0 = 0xFFFFFFFFu, e1 = 0xFFFFFFFFu;
            if (few) {
                if (b1 - b0 >= 1u) s0 = ft.starts[lo + b0];
                if (b1 - b0 >= 2u) s1 = ft.starts[lo + b0 + 1u];
                if (b3 - b2 >= 1u) e0 = ft.stops[lo + b2];
                if (b3 - b2 >= 2u) e1 = ft.stops[lo + b2 + 1u];
                s0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)s0), s1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)s1);
                e0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)e0), e1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)e1);
            }
#pragma unroll
            for (uint32_t r = 0; r < F_RPT; r++) {
                // #{s < qe} - #{e <= qs}, the entries in front of the brackets counted by the brackets themselves -- computed by every
                // lane, looked up or not (no branch); the two rare cases behind UNIFORM tests
                uint32_t c = (b0 + (s0 < qe[r] ? 1u : 0u) + (s1 < qe[r] ? 1u : 0u)) - (b2 + (e0 <= qs[r] ? 1u : 0u) + (e1 <= qs[r] ? 1u : 0u));
                const bool look = what[r] == 5;
                if (!few) { // a bracket with more than two entries: the searches inside it
                    if (look && ref[r] == r0) {
                        const uint32_t br[4] = {lo + b0, lo + b1, lo + b2, lo + b3};
                        c = count_overlaps(ft, name, (uint32_t)ref[r], qs[r], qe[r], br);
                    }
                }
                if (__ballot(look && ref[r] != r0)) { // a record on another sequence than the tile's first: the whole lists
                    if (look && ref[r] != r0) c = count_overlaps(ft, name, (uint32_t)ref[r], qs[r], qe[r], nullptr);
                }
                acc[r] |= min(c, 3u) << (2u * role); // (of a record that is not looked up: never read)
            }
        }
        // features.rs:186-214  UTR / CDS store: the if / else-if chain over the overlapping intervals sets, per distinct NAME, one flag
        // per overlapping interval in the order 5' UTR, 3' UTR, CDS among the roles that carry the name -- role X is set iff its name's
        // count exceeds the number of roles in front of X with the same name (roles may share a name).
        // :216-238  gene / exon store: a name that is also a UTR/CDS name never reaches this store (:322-333), and `name == gene` is
        // tested before `name == exon`.
        const uint32_t n5 = ft.role_name[NGSQ_ROLE_FIVE_PRIME_UTR], n3 = ft.role_name[NGSQ_ROLE_THREE_PRIME_UTR],
                       nc = ft.role_name[NGSQ_ROLE_CODING_SEQUENCE], ne = ft.role_name[NGSQ_ROLE_EXON], ng = ft.role_name[NGSQ_ROLE_GENE];
        const uint32_t rank3 = n3 == n5 ? 1u : 0u, rankc = (nc == n5 ? 1u : 0u) + (nc == n3 ? 1u : 0u);
        const bool gene_in_store = ng != n5 && ng != n3 && ng != nc;
        const bool exon_in_store = ne != n5 && ne != n3 && ne != nc && ne != ng;
        static_assert(NGSQ_ROLE_FIVE_PRIME_UTR == 0 && NGSQ_ROLE_THREE_PRIME_UTR == 1 && NGSQ_ROLE_CODING_SEQUENCE == 2, "the UTR/CDS chain's order");
#pragma unroll
        for (uint32_t r = 0; r < F_RPT; r++) {
            const bool look = what[r] == 5;
            auto cnt_of = [&](uint32_t role) { return (acc[r] >> (2u * role)) & 3u; };
            const bool utr5 = look && cnt_of(NGSQ_ROLE_FIVE_PRIME_UTR) > 0u;
            const bool utr3 = look && cnt_of(NGSQ_ROLE_THREE_PRIME_UTR) > rank3;
            const bool cds = look && cnt_of(NGSQ_ROLE_CODING_SEQUENCE) > rankc;
            const bool has_gene = gene_in_store && cnt_of(NGSQ_ROLE_GENE) > 0u;
            const bool has_exon = exon_in_store && cnt_of(NGSQ_ROLE_EXON) > 0u;
            const bool exonic = look && has_gene && has_exon, intronic = look && has_gene && !has_exon, intergenic = look && !has_gene;
            tally(cnt[F_UTR5], utr5);
            tally(cnt[F_UTR3], utr3);
            tally(cnt[F_CDS], cds);
            tally(cnt[F_INTERGENIC], intergenic);
            tally(cnt[F_EXONIC], exonic);
            tally(cnt[F_INTRONIC], intronic);
            tally(cnt[F_PROCESSED], look); // :240
            if (__ballot(what[r] != 5 && what[r] != 0)) { // (a record that is not looked up: one in a hundred)
                tally(cnt[F_IGN_FLAGS], what[r] == 1);
                tally(cnt[F_IGN_NONPRIMARY], what[r] == 3);
                tally(cnt[9], what[r] == 2);
                tally(cnt[10], what[r] == 4);
            }
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

// What Genomic Features needs of a record -- flag, sequence, position, reference span (features.rs:127-178) -- written out as a
// batch of its own (16 bytes per record: the span as ONE `M` operation, clamped to 2^28 - 1, beyond every sequence) for batches that
// arrive BEFORE the gene model does: ngsq_set_features runs k_features over them when the model is there (round 6: a gzipped GFF
// inflates on one host thread for seconds; the scan no longer waits for it).
__global__ __launch_bounds__(256) void k_features_defer(DeviceBatch b, uint16_t *__restrict__ flag, int32_t *__restrict__ ref, int32_t *__restrict__ pos,
                                                        uint16_t *__restrict__ ncig, uint32_t *__restrict__ cig) {
    NGSQ_FOREGROUND_WAVE();
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < b.n; i += (uint64_t)gridDim.x * 256) {
        const uint32_t n_ops = batch_n_ops(b, i);
        const uint64_t c0 = b.cigar_off ? b.cigar_off[i] : i * b.cigar_stride;
        uint64_t span = 0;
        for (uint32_t k = 0; k < n_ops; k++) {
            const uint32_t op = b.cigar[c0 + k];
            if ((0x18Du >> (op & 15u)) & 1u) span += op >> 4;
        }
        flag[i] = b.flag[i];
        ref[i] = b.ref_id[i];
        pos[i] = b.pos[i];
        ncig[i] = 1;
        cig[i] = (uint32_t)(span < 0x0FFFFFFFull ? span : 0x0FFFFFFFull) << 4; // <span>M
    }
}

hipError_t launch_features_defer(const LaunchInfo &li, const DeviceBatch &b, uint16_t *flag, int32_t *ref, int32_t *pos, uint16_t *ncig, uint32_t *cig,
                                 hipStream_t s) {
    if (!b.n) return hipSuccess;
    const uint32_t grid = (uint32_t)std::min<uint64_t>((b.n + 255) / 256, (uint64_t)li.n_cu * 16);
    hipLaunchKernelGGL(k_features_defer, dim3(grid), dim3(256), 0, s, b, flag, ref, pos, ncig, cig);
    return hipGetLastError();
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
