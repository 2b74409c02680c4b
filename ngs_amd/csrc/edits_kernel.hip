// edits_kernel.hip -- the Edits facet (edits.rs:177-353, utils/alignment.rs:29-126, utils/cigar.rs:6-23), round 4.
//
// What the reference keeps per position of a sequence is refs[p] (read bases under an `M` that equal the reference base) and
// alts[p] (those that differ), and per read the number of differing bases.  Rounds 1-3 tallied both arrays base by base: one
// atomic per compared base whether or not it matched (150 per read), ~6 instructions per base, 0.15 of the HBM roofline.
// On real data fewer than 1 % of the compared bases differ, so the arrays are kept in another form until the teardown:
//
//   alts[p]     as before -- one global atomic per MISMATCHING base only;
//   cover[p]    the number of reads whose `M` operations cover p, as a DIFFERENCE array in the slot that held refs
//               (entry p-1 += 1 at the first position of an `M`, entry q -= 1 at its last position q): two adds per `M`
//               operation, through the wave's LDS window and coalesced flushes (the scheme of fields_kernel.hip);
//   refs[p]     = cover[p] - alts[p], computed in place by the teardown (k_edits_refs) before the VAF histogram is taken.
//
// Both forms are sums of per-record contributions, so shards add their blocks as before (ngsq_exchange all-reduces them).
// The comparison itself is nibble-parallel: the reference is resident as PACKED 4-bit codes in the code space and nibble
// order of BAM's SEQ -- twice: once starting at base 0 and once starting at base 1, so that a read at an even or an odd
// position finds its reference bytes at a byte address -- and a read under one `M` is a run of 16-byte XORs whose non-zero
// nibbles are the mismatches: ~7 vector instructions per eight bases, a population count for the read's edit count, and the
// (rare) dwords that hold a mismatch are revisited for their positions afterwards.  Reads with other CIGARs walk their
// operations and compare each `M` eight bases at a time from any nibble offset (big-endian 64-bit windows).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "../../include/ngsq_shared.h"
#include "kernels.h"

namespace ngsq {

typedef unsigned long long u64;

namespace {

#ifndef EDITS_EXP
#define EDITS_EXP 0 // measurement builds only: 1 = no atomics for the mismatches, 2 = no cover (window, flush), 3 = no comparison, 4 = neither 1 nor 2;
                    // k_edits_rows: 5 = the cover window is not flushed to global memory, 6 = the alts window is not, 7 = neither
#endif
constexpr uint32_t ED_THREADS = 256;
constexpr uint32_t ED_PASSES = 4;                      // records per thread and tile: lane + 64 * pass of the wave's 256 consecutive records
constexpr uint32_t ED_WAVE_TILE = 64 * ED_PASSES;
constexpr uint32_t ED_TILE = ED_THREADS * ED_PASSES;   // records per block and tile
constexpr uint32_t ED_WINDOW = 1536;                   // entries of the difference array in one wave's LDS window
constexpr uint32_t ED_CHUNKS = 5;                      // 16-byte pieces of packed sequence compared per round (160 bases)
constexpr uint32_t ED_LIST = 4;                        // mismatching dwords a read may have on the fast path
constexpr uint32_t ED_HIST = 64;                       // k_edits: per-read edit counts tallied in LDS (the rest straight to the counters)
// k_edits_rows: ALL 513 bins in LDS, two 16-bit counters per word, flushed before one can wrap (a block tallies at most 1024
// records per tile).  Until round 5 only the counts below 64 were kept in LDS and a read with more went to the global counters
// with an atomic of its own: reads that differ from the reference in more than 63 bases (independent random bases: 112 of 150)
// then queued on a few dozen addresses of one L2 channel, ~9 ns each -- 127 ms per 100 M reads, the "16 x cliff" of VERDICT r4,
// none of it the comparison's.
constexpr uint32_t ED_HW = (NGSQ_EDITS_BINS + 1) / 2;  // words per histogram
constexpr uint32_t ED_HW_FLUSH_TILES = 63;             // 63 x 1024 < 65536

__device__ __forceinline__ uint32_t ed_wave_sum(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
// bit 4k set iff nibble k of x is not zero
__device__ __forceinline__ uint32_t nz_nibbles(uint32_t x) {
    uint32_t t = x | (x >> 2);
    t |= t >> 1;
    return t & 0x11111111u;
}
// the nibbles of a packed dword that hold its first nb bases (nb in 1..7; base 2b is the HIGH nibble of byte b)
__device__ __forceinline__ uint32_t lead_bases_mask(uint32_t nb) {
    const uint32_t full = nb & ~1u; // bases in whole bytes
    uint32_t m = full ? (0xFFFFFFFFu >> (32u - 4u * full)) : 0u;
    if (nb & 1u) m |= 0xF0u << (4u * full);
    return m;
}
__device__ __forceinline__ u64 ld64(const uint8_t *p) {
    u64 v;
    __builtin_memcpy(&v, p, 8);
    return v;
}
// the eight nibbles that start at nibble offset `q` of the packed run at `base`, first nibble in bits 31..28
__device__ __forceinline__ uint32_t nibbles8(const uint8_t *base, uint64_t q) {
    const u64 v = ld64(base + (q >> 1));
    const uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    const u64 be = (u64)__builtin_bswap32(lo) << 32 | __builtin_bswap32(hi); // bytes in stream order from the top
    return (uint32_t)((be << ((q & 1u) * 4u)) >> 32);
}

// one record, any shape, straight to the global arrays (utils/alignment.rs:48-107 operation by operation; every `M` is compared
// eight bases at a time from any nibble offset).  The main loop sends here what its fast path does not take: CIGARs other than
// [clip] M [clip], reads outside the wave's window or on another sequence, reads with more mismatching dwords than its
// list holds.  Returns 0, or 1 + the error counter the record belongs to; *out_edits = the read's edit count when 0.
__device__ __forceinline__ uint32_t ed_walk_record(const DeviceState &st, const DeviceBatch &b, uint64_t i, uint32_t *out_edits) {
    const uint32_t f = b.flag[i];
    const int32_t ref = b.ref_id[i], pos = b.pos[i];
    *out_edits = 0xFFFFFFFFu; // "not an Edits record"
    if (!(ref >= 0 && (uint32_t)ref < st.n_refs && pos >= 0)) return 0;
    const uint32_t n_ops = batch_n_ops(b, i);
    const uint64_t cbase = b.cigar_off ? b.cigar_off[i] : i * (uint64_t)b.cigar_stride;
    uint64_t span = 0;
    for (uint32_t k = 0; k < n_ops; k++) {
        const uint32_t cg = b.cigar[cbase + k], op = cg & 0xFu;
        if (op <= 8u && ((0x18Du >> op) & 1u)) span += cg >> 4;
    }
    const uint64_t L = st.ref_len[ref];
    const uint64_t s = (uint64_t)pos + 1, e = s + span - 1;
    if (e == 0 || s > L || (f & 0x404u)) return 0; // not yielded by query(); unmapped | duplicate (edits.rs:227-229)
    const uint64_t boff = st.ref_bases_off[ref];
    // edits.rs:245-261: no such sequence in the FASTA / the slice start..start+span runs past the FASTA's sequence (its own
    // length, which need not be @SQ LN: ref_edits_len -- a read may end beyond LN inside a longer FASTA sequence as long as
    // no `M` base lies there: see the operation loop) ...
    if (boff == NO_DEPTH || e > (st.ref_edits_len ? (uint64_t)st.ref_edits_len[ref] : L)) return 1;
    if (st.ref_bad_off) { // ... or holds a byte Base::try_from refuses (edits.rs:257-261 `?` on the collected Result)
        uint32_t lo = st.ref_bad_off[ref];
        const uint32_t hi0 = st.ref_bad_off[ref + 1];
        uint32_t hi = hi0;
        while (lo < hi) { // first listed position >= s
            const uint32_t mid = (lo + hi) >> 1;
            if ((uint64_t)st.ref_bad_pos[mid] < s) lo = mid + 1;
            else hi = mid;
        }
        if (lo < hi0 && (uint64_t)st.ref_bad_pos[lo] <= e) return 1;
    }
    uint32_t *const diff = st.edits + st.ref_edits_off[ref]; // entry p - 1 <-> position p (the slot that holds refs after the teardown)
    uint32_t *const alts = diff + (L + 1);
    const uint8_t *const sq = b.seq + (b.seq_off ? b.seq_off[i] : i * (uint64_t)b.seq_stride);
    const uint8_t *const rb = st.ref_bases + boff;
    const uint32_t l = b.l_seq[i];
    uint32_t edits = 0, qp = 0; // record_ptr
    uint64_t rp = 0;            // reference_ptr
    for (uint32_t k = 0; k < n_ops; k++) {
        const uint32_t cg = b.cigar[cbase + k], op = cg & 0xFu, len = cg >> 4;
        if (op > 8u) continue;
        const bool c_ref = (0x18Du >> op) & 1u; // M D N = X
        const bool c_seq = (0x193u >> op) & 1u; // M I S = X
        if (op == 0u) { // only Kind::Match compares (edits.rs:277)
            // (a record that runs out of bases inside an M: alignment.rs:84-87; like the reference, the positions visited
            // before the error stay counted -- the error aborts the run anyway)
            const uint32_t avail = (uint64_t)qp + len > l ? l - qp : len;
            const uint64_t p0 = (uint64_t)pos + rp;
            // edits.rs:283-291: refs/alts_per_position have LN + 1 bins and increment(..).unwrap() panics beyond them -- the
            // first `M` base at a position above LN stops the run (bases under D, N, = and X there do not: nothing is
            // incremented for them).  It comes before the read runs out iff fewer bases fit than the read still has; the
            // bases in front of it are counted like those in front of any other stop.
            const uint64_t fit = p0 < L ? L - p0 : 0ull;
            const bool beyond = fit < (uint64_t)avail;
            const uint32_t m = beyond ? (uint32_t)fit : avail;
            // 32 bases per round: the eight loads of four 8-base steps first, then their mismatches (an atomic between two
            // steps' loads kept the compiler from having more than one step's loads in flight: the walk of an aligner's 6 % of
            // reads with an insertion or a deletion took as long as the fast path took for the other 94 %)
            for (uint32_t j = 0; j < m; j += 32) {
                uint32_t xx[4];
#pragma unroll
                for (uint32_t u = 0; u < 4; u++) {
                    const uint32_t ju = j + 8u * u;
                    xx[u] = ju < m ? nibbles8(sq, qp + ju) ^ nibbles8(rb, p0 + ju) : 0u;
                }
#pragma unroll
                for (uint32_t u = 0; u < 4; u++) {
                    const uint32_t ju = j + 8u * u;
                    if (ju >= m) break;
                    if (m - ju < 8u) xx[u] &= 0xFFFFFFFFu << (4u * (8u - (m - ju)));
                    uint32_t t = nz_nibbles(xx[u]);
                    edits += (uint32_t)__popc(t);
                    while (t && EDITS_EXP != 1 && EDITS_EXP != 4) {
                        const uint32_t q = 7u - ((uint32_t)__builtin_ctz(t) >> 2); // the first base of the window is the top nibble
                        t &= t - 1;
                        atomicAdd(&alts[p0 + 1 + ju + q], 1u);
                    }
                }
            }
            if (m && EDITS_EXP != 2 && EDITS_EXP != 4) {
                atomicAdd(&diff[p0], 1u);
                atomicAdd(&diff[p0 + m], 0xFFFFFFFFu);
                st.counters[st.off_eseen + ref] = 1ull; // (a plain store: the sequence has Edits state)
            }
            if (beyond) return 1;
            qp += m;
            rp += m;
            if (m < len) return 2;
        } else {
            if (c_seq) {
                if ((uint64_t)qp + len > l) return 2;
                qp += len;
            }
            if (c_ref) rp += len;
        }
    }
    if (qp != l) return 3;      // alignment.rs:102-103 (the reference side is consumed by construction)
    if (edits > 512u) return 4; // edits.rs:296-300 unwrap()
    *out_edits = edits | (f & 0x40u ? 0x80000000u : 0u);
    return 0;
}

// what the main loop knows of a record before it touches its bases
struct EdCols {
    uint32_t flag, l, n_ops;
    int32_t ref, pos;
    uint64_t cbase;
};
struct EdCigar {
    uint32_t g0, g1, g2;
};

} // namespace

// (four waves per SIMD, what the block's 33 KB of LDS allow: left alone the two-segment comparison takes 136 registers)
__global__ __launch_bounds__(ED_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_edits(DeviceState st, DeviceBatch b, u64 *__restrict__ defer_bits) {
    NGSQ_FOREGROUND_WAVE();
    __shared__ uint32_t s_h1[ED_HIST], s_h2[ED_HIST];           // edit counts below ED_HIST; the rest goes straight to the counters
    __shared__ uint32_t s_win[(ED_THREADS / 64) * ED_WINDOW];
    __shared__ uint2 s_list[ED_LIST * ED_THREADS];              // per thread: (masked XOR dword, its index) of the dwords that hold a mismatch
    __shared__ u64 s_acc[4];
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    for (uint32_t i = tid; i < ED_HIST; i += ED_THREADS) s_h1[i] = s_h2[i] = 0;
    for (uint32_t i = tid; i < (ED_THREADS / 64) * ED_WINDOW; i += ED_THREADS) s_win[i] = 0;
    if (tid < 4) s_acc[tid] = 0;
    __syncthreads();
    uint32_t c[4] = {0, 0, 0, 0}; // bad_ref, record_short, not_consumed, too_many
    uint32_t *const win = s_win + (tid >> 6) * ED_WINDOW;
    uint2 *const list = s_list + (tid >> 6) * (ED_LIST * 64) + lane; // entry k at list[64 * k]

    // facts of the sequence the wave's window is anchored on (reloaded only when it changes: scalar registers)
    int32_t meta_ref = -1;
    uint64_t meta_eoff = NO_DEPTH, meta_boff = NO_DEPTH;
    uint32_t meta_L = 0, meta_E = 0; // @SQ LN; what the fast path may reach (the FASTA's bases of the sequence: st.ref_fast_len)

    auto tally = [&](uint32_t edits, bool first) { // edits.rs:296-300
        if (edits < ED_HIST) atomicAdd(first ? &s_h1[edits] : &s_h2[edits], 1u);
        else atomicAdd(&st.counters[(first ? st.off_edits1 : st.off_edits2) + edits], 1ull);
    };

    const uint64_t n_tiles = (b.n + ED_TILE - 1) / ED_TILE;
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint64_t w0 = tile * ED_TILE + (uint64_t)(tid >> 6) * ED_WAVE_TILE; // the wave's first record
        if (w0 >= b.n) continue;
        // ---- stage 1 of a pass: the fixed-width columns of its 64 records (records behind the end read as unmapped)
        auto load_cols = [&](uint32_t pass) -> EdCols {
            const uint64_t i = w0 + (uint64_t)pass * 64 + lane;
            const bool live = i < b.n;
            const uint64_t ii = live ? i : b.n - 1;
            EdCols r;
            r.flag = live ? (uint32_t)b.flag[ii] : 0x4u;
            r.ref = b.ref_id[ii];
            r.pos = b.pos[ii];
            r.l = b.l_seq[ii];
            r.n_ops = b.n_cigar[ii];
            r.cbase = b.cigar_off ? b.cigar_off[ii] : ii * (uint64_t)b.cigar_stride;
            return r;
        };
        // ---- stage 2: the first three CIGAR operations
        auto load_cigar = [&](const EdCols &r) -> EdCigar {
            EdCigar g{0, 0, 0};
            if (r.n_ops > 0) g.g0 = b.cigar[r.cbase];
            if (r.n_ops > 1) g.g1 = b.cigar[r.cbase + 1];
            if (r.n_ops > 2) g.g2 = b.cigar[r.cbase + 2];
            return g;
        };
        int32_t win_ref = -1;
        uint32_t win_base = 0, top = 0;
        // ---- stage 3 + the comparison: returns true when the record must take ed_walk_record instead
        auto compare = [&](const EdCols &r, const EdCigar &g, uint32_t pass) -> bool {
            if (!(r.ref >= 0 && (uint32_t)r.ref < st.n_refs && r.pos >= 0) || (r.flag & 0x404u)) return false; // never an Edits record
            // [clip] M [clip]: read base q lies on 0-based reference position P + q, P = pos - (leading clip).
            // M (I|D|N) M (round 5): two such segments -- the first as if the rest were clipped, the second with the reference
            // gap - ins bases further on; an aligner gives one read in sixteen an insertion or a deletion, the 50-300 bp workload one
            // in seven an insertion, a deletion or a skip, and until round 5 all of those went to the walk kernel (5.2 of this shape's 11.3 ms)
            uint32_t a = 0, m = 0, z = 0, m2 = 0, ins = 0, gap = 0;
            const uint32_t o0 = g.g0 & 15u, o1 = g.g1 & 15u, o2 = g.g2 & 15u;
            bool shape = false;
            if (r.n_ops == 1) shape = o0 == 0u, m = g.g0 >> 4;
            else if (r.n_ops == 2 && o0 == 4u && o1 == 0u) shape = true, a = g.g0 >> 4, m = g.g1 >> 4;
            else if (r.n_ops == 2 && o0 == 0u && o1 == 4u) shape = true, m = g.g0 >> 4, z = g.g1 >> 4;
            else if (r.n_ops == 3 && o0 == 4u && o1 == 0u && o2 == 4u) shape = true, a = g.g0 >> 4, m = g.g1 >> 4, z = g.g2 >> 4;
            else if (r.n_ops == 3 && o0 == 0u && o1 >= 1u && o1 <= 3u && o2 == 0u && (g.g2 >> 4)) {
                shape = true, m = g.g0 >> 4, m2 = g.g2 >> 4;
                if (o1 == 1u) ins = g.g1 >> 4;
                else gap = g.g1 >> 4;
                z = ins + m2; // (read bases behind the first M)
            }
            const uint64_t s = (uint64_t)r.pos + 1, e = s + m + gap + m2 - 1; // 1-based first and last position
            const uint64_t i0 = (uint64_t)(uint32_t)r.pos - win_base, i1 = i0 + m; // the first M's entries of the difference array in the window
            const bool fast = shape && m && (uint64_t)a + m + z == r.l && r.ref == win_ref && (uint32_t)r.pos >= a + ins && e <= meta_E &&
                              (uint32_t)r.pos >= win_base && i1 < ED_WINDOW && r.l < (1u << 16);
            if (!fast) return true;
            const uint64_t i = w0 + (uint64_t)pass * 64 + lane;
            const uint8_t *const sq = b.seq + (b.seq_off ? b.seq_off[i] : i * (uint64_t)b.seq_stride);
            uint32_t edits = 0, cnt = 0;
            // one M at a time: read bases [v0, v1) against the reference, base q on 0-based position P + q (a loop, not two copies of
            // the body: inlined twice it took 138 registers -- three waves per SIMD instead of four)
            const uint32_t P1 = (uint32_t)r.pos - a, P2 = (uint32_t)r.pos + gap - ins;
            // the listed dwords' mismatches -> alts (fewer than one per read on real data: after the comparison).  A read with more
            // mismatching dwords than the list holds is left to the walk kernel: this layout has no LDS window for alts, so reads
            // that differ much from the reference cost a global atomic per mismatch either way (50-300 bp reads with 5 % of their
            // bases substituted: 38.8 ms per 100 M; emptying the list as it fills, inside the comparison: 32 ms, and 4 % slower on
            // the reads the kernel is built for -- measured in round 5, not kept)
            auto drain = [&]() {
                for (uint32_t k = 0; k < cnt && EDITS_EXP != 1 && EDITS_EXP != 4; k++) {
                    const uint2 en = list[64 * k];
                    uint32_t t = nz_nibbles(en.x);
                    uint32_t *const alts = st.edits + meta_eoff + ((uint64_t)meta_L + 1) + (uint64_t)(en.y >> 16 ? P2 : P1) + 1 + 8 * (en.y & 0xFFFFu);
                    while (t) {
                        const uint32_t q = (uint32_t)__builtin_ctz(t) >> 2; // nibble q: byte q >> 1, its high nibble (q odd) is the earlier base
                        t &= t - 1;
                        atomicAdd(&alts[(q & ~1u) + ((q & 1u) ^ 1u)], 1u);
                    }
                }
                cnt = 0;
            };
#pragma unroll 1
            for (uint32_t seg = 0; seg < (m2 ? 2u : 1u); seg++) {
                const uint32_t v0 = seg ? m + ins : a, v1 = seg ? r.l : a + m, P = seg ? P2 : P1;
                const uint8_t *const rb = (P & 1u ? st.ref_bases_odd : st.ref_bases) + meta_boff + (P >> 1);
                for (uint32_t c0 = v0 >> 5; c0 * 32 < v1 && EDITS_EXP != 3; c0 += ED_CHUNKS) {
                    uint4 sv[ED_CHUNKS], rv[ED_CHUNKS];
#pragma unroll
                    for (uint32_t k = 0; k < ED_CHUNKS; k++) {
                        sv[k] = rv[k] = make_uint4(0, 0, 0, 0);
                        if ((c0 + k) * 32 < v1) {
                            __builtin_memcpy(&sv[k], sq + 16 * (c0 + k), 16);
                            __builtin_memcpy(&rv[k], rb + 16 * (c0 + k), 16);
                        }
                    }
#pragma unroll
                    for (uint32_t k = 0; k < ED_CHUNKS; k++) {
                        const uint32_t x[4] = {sv[k].x ^ rv[k].x, sv[k].y ^ rv[k].y, sv[k].z ^ rv[k].z, sv[k].w ^ rv[k].w};
#pragma unroll
                        for (uint32_t d = 0; d < 4; d++) {
                            const uint32_t j = (c0 + k) * 4 + d, b0 = 8 * j; // dword j holds read bases [b0, b0 + 8)
                            uint32_t xx = x[d];
                            if (b0 < v0 || b0 + 8 > v1) { // a dword at an end of the M (or outside it)
                                const uint32_t u = b0 < v0 ? min(v0 - b0, 8u) : 0u, v = b0 + 8 > v1 ? (v1 > b0 ? v1 - b0 : 0u) : 8u;
                                const uint32_t keep = (v >= 8u ? 0xFFFFFFFFu : v ? lead_bases_mask(v) : 0u) & ~(u >= 8u ? 0xFFFFFFFFu : u ? lead_bases_mask(u) : 0u);
                                xx &= keep;
                            }
                            const uint32_t t = nz_nibbles(xx);
                            edits += (uint32_t)__popc(t);
                            if (t) {
                                if (cnt < ED_LIST) list[64 * cnt] = make_uint2(xx, j | seg << 16);
                                cnt += 1;
                            }
                        }
                    }
                }
            }
            if (cnt > ED_LIST) return true; // more mismatching dwords than the list holds: the walk does this record
            drain();
            if (EDITS_EXP != 2 && EDITS_EXP != 4) {
                atomicAdd(&win[i0], 1u);
                atomicAdd(&win[i1], 0xFFFFFFFFu);
                top = max(top, (uint32_t)i1);
                if (m2) { // the second M's cover: in the window when it reaches that far, else straight to the array (a skip of kilobases)
                    const uint64_t q0 = i1 + gap, q1 = q0 + m2;
                    if (q1 < ED_WINDOW) {
                        atomicAdd(&win[q0], 1u);
                        atomicAdd(&win[q1], 0xFFFFFFFFu);
                        top = max(top, (uint32_t)q1);
                    } else {
                        uint32_t *const diff = st.edits + meta_eoff + win_base;
                        atomicAdd(&diff[q0], 1u);
                        atomicAdd(&diff[q1], 0xFFFFFFFFu);
                        st.counters[st.off_eseen + win_ref] = 1ull;
                    }
                }
            }
            if (edits > 512u) c[3] += 1;
            else tally(edits, r.flag & 0x40u);
            return false;
        };
        // ---- the four passes, software-pipelined: the columns of pass p + 2 and the CIGARs of pass p + 1 are in flight while
        // pass p is compared (nothing in compare() but its own loads touches memory on the usual path)
        EdCols c0 = load_cols(0), c1 = load_cols(1);
        {   // the wave's window: entries [win_base, win_base + ED_WINDOW) of the difference array of the sequence of its first record
            const int32_t fr = __builtin_amdgcn_readfirstlane(c0.ref), fp = __builtin_amdgcn_readfirstlane(c0.pos);
            if (fr >= 0 && (uint32_t)fr < st.n_refs && fp >= 0) {
                if (fr != meta_ref) {
                    const uint64_t eo = st.ref_edits_off[fr], bo = st.ref_bases_off[fr];
                    const uint32_t l = st.ref_len[fr], fl = st.ref_fast_len ? st.ref_fast_len[fr] : l;
                    meta_ref = fr;
                    meta_eoff = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(eo >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)eo);
                    meta_boff = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(bo >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)bo);
                    meta_L = (uint32_t)__builtin_amdgcn_readfirstlane((int)l);
                    meta_E = (uint32_t)__builtin_amdgcn_readfirstlane((int)fl);
                }
                if (meta_boff != NO_DEPTH) {
                    win_ref = fr;
                    win_base = (uint32_t)fp & ~3u; // entry index = 0-based position: <= the first record's
                }
            }
        }
        const EdCigar g0 = load_cigar(c0);
        const EdCols c2 = load_cols(2);
        const EdCigar g1 = load_cigar(c1);
        // (a record the fast path does not take is marked in the launch's bitmap: k_edits_walk does it afterwards)
        auto mark = [&](uint32_t pass, bool d) {
            const u64 dm = __ballot(d);
            if (lane == 0 && w0 + (uint64_t)pass * 64 < b.n) defer_bits[(w0 >> 6) + pass] = dm;
        };
        mark(0, compare(c0, g0, 0));
        const EdCols c3 = load_cols(3);
        const EdCigar g2 = load_cigar(c2);
        mark(1, compare(c1, g1, 1));
        const EdCigar g3 = load_cigar(c3);
        mark(2, compare(c2, g2, 2));
        mark(3, compare(c3, g3, 3));
        // ---- the wave flushes the touched part of its window with coalesced global atomics and leaves it zeroed.
        // No barrier: LDS operations of one wave execute in order.
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) top = max(top, (uint32_t)__shfl_xor(top, o, 64));
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (win_ref >= 0 && top) {
            uint32_t *const dst = st.edits + meta_eoff + win_base;
            if (lane == 0) st.counters[st.off_eseen + win_ref] = 1ull; // the sequence has Edits state (plain store)
            for (uint32_t ib = 0; ib <= top; ib += 256) {
                uint32_t v[4];
#pragma unroll
                for (uint32_t k = 0; k < 4; k++) {
                    const uint32_t i = ib + 64 * k + lane;
                    v[k] = i < ED_WINDOW ? win[i] : 0u;
                }
#pragma unroll
                for (uint32_t k = 0; k < 4; k++) {
                    const uint32_t i = ib + 64 * k + lane;
                    if (v[k]) {
                        atomicAdd(&dst[i], v[k]);
                        win[i] = 0;
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    __syncthreads();
    for (uint32_t i = tid; i < ED_HIST; i += ED_THREADS) {
        uint32_t v = s_h1[i];
        if (v) atomicAdd(&st.counters[st.off_edits1 + i], (u64)v);
        v = s_h2[i];
        if (v) atomicAdd(&st.counters[st.off_edits2 + i], (u64)v);
    }
    const uint32_t idx[4] = {C_ERR + E_EDITS_BAD_REF, C_ERR + E_EDITS_SHORT, C_ERR + E_EDITS_NOT_CONSUMED, C_ERR + E_EDITS_TOO_MANY};
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t r = ed_wave_sum(c[k]);
        if (lane == 0 && r) atomicAdd(&s_acc[k], (u64)r);
    }
    __syncthreads();
    if (tid < 4 && s_acc[tid]) atomicAdd(&st.counters[idx[tid]], s_acc[tid]);
}

// ---------------------------------------------------------------------------
// Fixed-pitch sequence rows (reads of one length, the layout the readers choose for them): the comparison with a LANE PER
// 16-BYTE WINDOW of the packed rows instead of a lane per record.  With a lane per record every one of a read's five 16-byte
// loads sweeps the wave's whole 4.8 KB of rows (64 lanes 75 bytes apart: ~38 cache lines per instruction, each line fetched
// by five instructions); consecutive lanes on consecutive windows read 1 KiB contiguous per instruction -- every line once
// (measured with the bytes deliberately wrong: 3.65 -> 2.70 ms per 100 M reads).  Per 64 records a wave runs
//   1. lane = record: columns, CIGAR shape, the record's descriptor (reference position of its base 0, compared bases
//      [v0, v1)) into LDS, its `M` into the cover window;
//   2. lane = window g of the 64 rows (record g / R, window g % R; R windows per row): 16 bytes of sequence XOR 16 bytes of
//      the packed reference -- consecutive lanes of a record read consecutive reference bytes, neighbouring records nearly the
//      same ones (L1) --, masked to [v0, v1), mismatches counted into the record's LDS slot and added to alts;
//   3. lane = record: the per-read edit count into the histogram.
// Records the fast path does not take (other CIGARs, outside the window, another sequence) go through ed_walk_record.
// ---------------------------------------------------------------------------
constexpr uint32_t ED_NW = 5;                    // windows per row the kernel is built for (reads of up to 160 bases)
constexpr uint32_t EDR_TILE = ED_THREADS * 4;    // records a block is launched for (the grid; the waves cut their runs of passes themselves)
#ifndef NGSQ_EDR_WINDOW
#define NGSQ_EDR_WINDOW 1408 // (1536 until the histograms and the GC tally took 2.8 KB more: the block's LDS stays at 32 granules of 1280 bytes, four blocks per CU)
#endif
// waves per SIMD the register allocation is made for: four, what the block's LDS allows (the variant that also tallies GC Content
// takes 143-148 registers when left alone -- three waves per SIMD: 3.9 ms per 100 M reads against 3.56 at four; the plain variant
// fits by itself).  NGSQ_EDR_WAVES: measurement builds.
#ifndef NGSQ_EDR_FAR
#define NGSQ_EDR_FAR 1 // measurement builds: 0 = a read whose first M reaches beyond the wave's window is the walk kernel's
#endif
#ifndef NGSQ_EDR_WAVES
#define NGSQ_EDR_WAVES 4
#endif
#define EDR_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(NGSQ_EDR_WAVES, NGSQ_EDR_WAVES)))
constexpr uint32_t EDR_WINDOW = NGSQ_EDR_WINDOW;            // entries of the wave's cover window (256 sorted reads: 790 positions at 60x, 1430 at 30x: there the last few reads of a tile go to the walk)
constexpr uint32_t GC_NONE = 0x3FFu;             // "no GC window": an offset whose window lies behind every 16-byte window of a row
constexpr uint32_t GC_HW = (NGSQ_GC_BINS + 1) / 2; // words of a 101-bin histogram of 16-bit pairs

struct EdRowCols {
    uint32_t flag, l, n_ops, g0, g1, g2;
    int32_t ref, pos;
    uint64_t so;  // RAGGED: where the record's packed bases begin in the column,
    uint32_t sb;  //         and how many bytes it has there (saturated)
};
constexpr uint32_t EDG_MAXW = 16;     // RAGGED: 16-byte windows a record may have on the fast path: 255 bytes, reads of up to 510 bases
constexpr uint32_t EDG_WINDOW = 1152; // RAGGED: entries of the wave's cover window (256 fewer than EDR_WINDOW pay for the window -> record map)

// CIG_OFF: the CIGARs are addressed through cigar_off (their loads then wait for the offsets; with a fixed pitch they are
// prefetched with the columns of the pass)
// GC (round 5): the GC Content facet (gc_content.rs:38-100) of the same records in the same pass -- when both facets are enabled
// the packed SEQ column (a third of a record's bytes) is read ONCE: the window lanes hold every 16 bytes of every row anyway, the
// GC window is bases [off, off + 100) of the row, and its tally is the same nibble-parallel classification k_gc does
// ((C ^ G) & ~(A | T) on the bit planes) under the mask table the comparison uses.  Per pass: lane = record works out the
// window's offset (flag filter, length filter, ngsq_gc_offset_fn) into LDS; lane = window adds its (gc, at) counts to the
// record's 16-bit pair; lane = record tallies the two 101-bin histograms (gc and at per read, 16-bit pairs) from which the
// flush derives every counter of the facet (total_gc = sum g * hist[g], ...).  k_gc is then not launched at all.
// RAGGED (round 5): the offsets layout (reads of different lengths: SEQ at seq_off[i]) through the same window lanes.  A record has
// ceil(bytes / 16) windows that start at ITS first byte (the last one reads on into the next record's bytes and is masked, as a
// row's last window is); the records of a pass lie back to back in the column, so consecutive lanes still read consecutive
// 16-byte pieces but for one overlap per record.  Which record a window belongs to comes from a byte map in LDS that the record
// lanes fill behind a prefix sum of their window counts (k_qual_ragged's scheme): one byte store per window, and per window one
// byte read + one 8-byte read (the record's first window and byte offset).  Records that are not on the fast path have no windows.
// Until then this layout went through k_edits, a lane per RECORD: 64 lanes 25..150 bytes apart per load instruction, 11.1 ms per
// 100 M reads of 50-300 bases, and 32 ms with 5 % of their bases substituted (no alts window in LDS there).
template <bool CIG_OFF, bool GC, bool RAGGED>
__global__ __launch_bounds__(ED_THREADS) EDR_WAVES_ATTR void k_edits_rows(DeviceState st, DeviceBatch b, uint32_t R, uint32_t recip, u64 *__restrict__ defer_bits) {
    NGSQ_FOREGROUND_WAVE();
    __shared__ uint32_t s_h1[ED_HW], s_h2[ED_HW];               // per-read edit counts, 16-bit pairs
    static_assert(!(GC && RAGGED), "the GC tally is fused into the fixed-pitch variants only");
    constexpr uint32_t WIN = RAGGED ? EDG_WINDOW : GC ? EDR_WINDOW - 64 : EDR_WINDOW, ALTW = WIN / 2; // (GC: its 1.4 KB come out of the window -- the block stays at 32 granules)
    __shared__ uint32_t s_win[(ED_THREADS / 64) * WIN];  // cover: difference entries
    __shared__ uint32_t s_alt[(ED_THREADS / 64) * ALTW]; // mismatches per position of the same window
    __shared__ uint8_t s_wmap[RAGGED ? ED_THREADS * EDG_MAXW : 4]; // RAGGED: per wave, the record slot of each of the pass's windows
    __shared__ uint2 s_rg[RAGGED ? ED_THREADS : 1];                // RAGGED: per record slot (byte offset of its bases from the pass's first byte, first window | windows << 16)
    __shared__ uint2 s_desc[ED_THREADS];   // per record of the wave's current 64: (byte offset of its base 0 in the packed reference, v0 | v1 << 9 | window entry of base 0 << 18); v1 = 0: not on the fast path
    __shared__ uint32_t s_edits[ED_THREADS];
    __shared__ uint32_t s_tmask[33];       // [n]: the bits 4 q + d of the first n bases of a window (base 8 d + (q ^ 1))
    __shared__ uint8_t s_ilist[ED_THREADS]; // per wave: the lanes of the pass's records of the shape M (I|D|N) M, in order
    __shared__ uint32_t s_gap[ED_THREADS];  // per record slot of that shape: the reference bases between its two M (a deletion's, a skip's)
    __shared__ u64 s_acc[4];
    // GC: per record of the wave's current 64 its window offset (GC_NONE: not processed) and its (gc | at << 7) sum, two records per
    // word; the block's two histograms; records ignored for their flags / their length
    __shared__ uint16_t s_goff[GC ? ED_THREADS : 2];
    __shared__ uint32_t s_gacc[GC ? ED_THREADS / 2 : 1];
    __shared__ uint32_t s_ghist[GC ? GC_HW : 1], s_ahist[GC ? GC_HW : 1], s_gign[2];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (uint32_t i = tid; i < ED_HW; i += ED_THREADS) s_h1[i] = s_h2[i] = 0;
    if (GC) {
        if (tid < GC_HW) s_ghist[tid] = s_ahist[tid] = 0;
        if (tid < 2) s_gign[tid] = 0;
        if (tid < ED_THREADS / 2) s_gacc[tid] = 0;
    }
    if (tid < 33) {
        uint32_t m = 0;
        for (uint32_t i = 0; i < tid; i++) m |= 1u << (4u * ((i & 7u) ^ 1u) + (i >> 3));
        s_tmask[tid] = m;
    }
    for (uint32_t i = tid; i < (ED_THREADS / 64) * WIN; i += ED_THREADS) s_win[i] = 0;
    for (uint32_t i = tid; i < (ED_THREADS / 64) * ALTW; i += ED_THREADS) s_alt[i] = 0;
    if (tid < 4) s_acc[tid] = 0;
    __syncthreads();
    uint32_t c_too_many = 0;
    uint32_t *const win = s_win + wv * WIN;
    uint32_t *const altw = s_alt + wv * ALTW;
    uint8_t *const wmap = s_wmap + (RAGGED ? wv * (64 * EDG_MAXW) : 0);
    uint2 *const rg = s_rg + (RAGGED ? wv * 64 : 0);
    uint2 *const desc = s_desc + wv * 64;
    uint32_t *const red = s_edits + wv * 64; // bits 0..9: the record's edit count; bits 10..: what its second M needs (below)
    uint8_t *const ilist = s_ilist + wv * 64;
    uint32_t *const gapl = s_gap + wv * 64;
    uint16_t *const goff = s_goff + (GC ? wv * 64 : 0);
    uint32_t *const gacc = s_gacc + (GC ? wv * 32 : 0);

    int32_t meta_ref = -1;
    uint64_t meta_eoff = NO_DEPTH, meta_boff = NO_DEPTH;
    uint32_t meta_L = 0, meta_E = 0; // @SQ LN; what the fast path may reach (the FASTA's bases of the sequence: st.ref_fast_len)

    const uint32_t stride = b.seq_stride;
    const uint32_t odd_delta = (uint32_t)(st.ref_bases_odd - st.ref_bases); // (both copies lie inside 4 GiB: launch_edits)
    const uint64_t n = b.n;
    // the columns of a pass (records behind the end read as unmapped) and, with a fixed CIGAR pitch, the first three operations
    auto load_cols = [&](uint64_t r0) -> EdRowCols {
        const uint64_t i = r0 + lane;
        const bool live = i < n;
        const uint64_t ii = live ? i : n - 1;
        EdRowCols r;
        r.flag = live ? (uint32_t)b.flag[ii] : 0x4u;
        r.ref = b.ref_id[ii];
        r.pos = b.pos[ii];
        r.l = b.l_seq[ii];
        r.n_ops = b.n_cigar[ii];
        r.g0 = r.g1 = r.g2 = 0;
        r.so = 0, r.sb = 0;
        if (RAGGED) {
            r.so = b.seq_off[ii];
            const uint64_t nb = b.seq_off[ii + 1] - r.so;
            r.sb = nb < 0xFFFFFFFFull ? (uint32_t)nb : 0xFFFFFFFFu;
        }
        if (!CIG_OFF) {
            const uint64_t cb = ii * (uint64_t)b.cigar_stride;
            r.g0 = b.cigar[cb];
            if (b.cigar_stride > 1) r.g1 = b.cigar[cb + 1];
            if (b.cigar_stride > 2) r.g2 = b.cigar[cb + 2];
        }
        return r;
    };
    int32_t win_ref = -1;
    uint32_t win_base = 0, top = 0;
    // the block's edit-count histograms -> the counters (every ED_HW_FLUSH_TILES tiles, and at the end)
    auto flush_hist = [&]() {
        __syncthreads();
        for (uint32_t i = tid; i < ED_HW; i += ED_THREADS) {
            const uint32_t v1 = s_h1[i], v2 = s_h2[i];
            if (v1 & 0xFFFFu) atomicAdd(&st.counters[st.off_edits1 + 2 * i], (u64)(v1 & 0xFFFFu));
            if (v1 >> 16) atomicAdd(&st.counters[st.off_edits1 + 2 * i + 1], (u64)(v1 >> 16));
            if (v2 & 0xFFFFu) atomicAdd(&st.counters[st.off_edits2 + 2 * i], (u64)(v2 & 0xFFFFu));
            if (v2 >> 16) atomicAdd(&st.counters[st.off_edits2 + 2 * i + 1], (u64)(v2 >> 16));
            s_h1[i] = s_h2[i] = 0;
        }
        if (GC && tid < 64) { // the GC facet's counters from its two histograms (the first wave holds all 51 words)
            u64 pg = 0, pa = 0, pn = 0;
            if (tid < GC_HW) {
                const uint32_t g = s_ghist[tid], a = s_ahist[tid];
                const uint32_t g0 = g & 0xFFFFu, g1 = g >> 16;
                if (g0) atomicAdd(&st.counters[OFF_GC_HIST + 2 * tid], (u64)g0);
                if (g1) atomicAdd(&st.counters[OFF_GC_HIST + 2 * tid + 1], (u64)g1); // (bin 101 does not exist: its half stays 0)
                pg = (u64)(2 * tid) * g0 + (u64)(2 * tid + 1) * g1;
                pa = (u64)(2 * tid) * (a & 0xFFFFu) + (u64)(2 * tid + 1) * (a >> 16);
                pn = g0 + g1;
                s_ghist[tid] = s_ahist[tid] = 0;
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                pg += __shfl_down(pg, o, 64);
                pa += __shfl_down(pa, o, 64);
                pn += __shfl_down(pn, o, 64);
            }
            if (tid == 0) {
                if (pn) {
                    atomicAdd(&st.counters[C_GC_GC], pg);
                    atomicAdd(&st.counters[C_GC_AT], pa);
                    atomicAdd(&st.counters[C_GC_OTHER], (u64)NGSQ_GC_WINDOW * pn - pg - pa);
                    atomicAdd(&st.counters[C_GC_PROCESSED], pn);
                }
                if (s_gign[0]) atomicAdd(&st.counters[C_GC_IGN_FLAGS], (u64)s_gign[0]);
                if (s_gign[1]) atomicAdd(&st.counters[C_GC_IGN_SHORT], (u64)s_gign[1]);
                s_gign[0] = s_gign[1] = 0;
            }
        }
        __syncthreads();
    };
    // ---- the wave adds the touched part of its two windows to the global arrays (coalesced atomics) and leaves them zeroed
    auto flush_windows = [&]() {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) top = max(top, (uint32_t)__shfl_xor(top, o, 64));
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (win_ref >= 0 && top) {
            uint32_t *const dst = st.edits + meta_eoff + win_base;
            uint32_t *const adst = dst + ((uint64_t)meta_L + 1) + 1; // alts[1 + 0-based position]
            if (lane == 0) st.counters[st.off_eseen + win_ref] = 1ull; // the sequence has Edits state (plain store)
            for (uint32_t ib = 0; ib <= top; ib += 256) {
                uint32_t v[4];
#pragma unroll
                for (uint32_t k = 0; k < 4; k++) {
                    const uint32_t i = ib + 64 * k + lane;
                    v[k] = i < WIN ? win[i] : 0u;
                }
#pragma unroll
                for (uint32_t k = 0; k < 4; k++) {
                    const uint32_t i = ib + 64 * k + lane;
                    if (v[k]) {
                        if (EDITS_EXP != 5 && EDITS_EXP != 7) atomicAdd(&dst[i], v[k]);
                        win[i] = 0;
                    }
                }
            }
            // The alts window: two 16-bit counters per LDS word = two neighbouring 32-bit entries of the global array, added with ONE
            // 64-bit atomic (counts stay far below 2^32, so nothing carries from the low entry into the high one).  The flush is paid
            // per atomic instruction and line, not per lane (EDITS_EXP 6: the alts' flush cost as much as the cover's although only one
            // position in four holds a mismatch), so this halves it.  Whether a window word's two entries share an aligned 8 bytes
            // depends on the sequence's place in the block: if not, a lane adds the high half of the word before its own and the
            // low half of its own.  (A mismatch lies below the end of its read's M: below `top`.)
            const uint32_t n_dw = (top >> 1) + 1u;
            const bool straddle = ((meta_eoff + (uint64_t)meta_L) & 1ull) != 0; // &adst[0] = edits + meta_eoff + win_base + L + 2 entries, win_base % 4 == 0
            for (uint32_t ib = 0; ib < n_dw + (straddle ? 1u : 0u); ib += 64) {
                const uint32_t i = ib + lane;
                const uint32_t own = i < n_dw && i < ALTW ? altw[i] : 0u;
                u64 v;
                uint32_t *at;
                if (!straddle) {
                    v = (u64)(own & 0xFFFFu) | (u64)(own >> 16) << 32;
                    at = adst + 2 * i;
                } else {
                    const uint32_t before = i >= 1 && i - 1 < n_dw && i - 1 < ALTW ? altw[i - 1] : 0u;
                    v = (u64)(before >> 16) | (u64)(own & 0xFFFFu) << 32;
                    at = adst + 2 * i - 1;
                }
                if (v && EDITS_EXP != 6 && EDITS_EXP != 7) atomicAdd(reinterpret_cast<u64 *>(at), v);
            }
            for (uint32_t ib = 0; ib < n_dw; ib += 64) // (behind the adds: a lane reads its neighbour's word above)
                if (ib + lane < n_dw && ib + lane < ALTW) altw[ib + lane] = 0;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        top = 0;
    };
    // Round 5: a wave walks a CONTIGUOUS run of passes (64 records each) and flushes its windows when the next pass would not fit them
    // -- not every four passes: at 60x a window of 1408 entries holds seven passes of sorted 150-base reads, at 120x fourteen -- and
    // re-anchors at that pass's first record.  Fewer flushes: fewer entries flushed twice where neighbouring tiles overlap, and fewer
    // rounds of global atomics for the pass behind them to wait out (vmcnt is in order on gfx9: a wave that wants a load's result waits
    // for every atomic it issued before the load as well).
    const uint64_t n_pass = (n + 63) / 64, waves_total = (uint64_t)gridDim.x * (ED_THREADS / 64);
    const uint64_t ppw = (n_pass + waves_total - 1) / waves_total;              // passes per wave: the same for all (flush_hist is a block's)
    const uint64_t p0 = ((uint64_t)blockIdx.x * (ED_THREADS / 64) + wv) * ppw;  // this wave's first pass
    EdRowCols cur = load_cols(p0 * 64 < n ? p0 * 64 : 0);
    uint32_t rounds = 0;
#pragma unroll 1
    for (uint64_t it = 0; it < ppw; it++) {
        if (rounds == 4 * ED_HW_FLUSH_TILES) { // (uniform over the block; a block tallies at most 256 records per round)
            flush_hist();
            rounds = 0;
        }
        rounds += 1;
        {
            const uint64_t r0 = (p0 + it) * 64;
            if (r0 >= n) continue;
            {   // does the pass fit the window?  (its records reach at most l + 16 entries behind their position)
                const int32_t fr = __builtin_amdgcn_readfirstlane(cur.ref), fp = __builtin_amdgcn_readfirstlane(cur.pos);
                // (judged by the pass's LAST record: in a sorted file the one that reaches furthest -- three scalar instructions; a wave-wide
                // maximum cost the kernel 3 %.  A record that reaches beyond the window all the same is left to the walk kernel, as ever)
                const uint32_t lastl = (uint32_t)min(n - 1 - r0, (uint64_t)63);
                const int32_t lr = __builtin_amdgcn_readlane(cur.ref, lastl), lp = __builtin_amdgcn_readlane(cur.pos, lastl);
                const uint32_t ll = (uint32_t)__builtin_amdgcn_readlane((int)cur.l, lastl);
                uint32_t need = (lr == win_ref && lp >= 0 && (uint32_t)lp >= win_base) ? (uint32_t)lp - win_base + ll + 16u : 0u;
                if (RAGGED) { // reads of different lengths: the last record is not the one that reaches furthest
                    uint32_t reach = (cur.ref == win_ref && cur.pos >= 0 && (uint32_t)cur.pos >= win_base) ? (uint32_t)cur.pos - win_base + min(cur.l, 511u) + 16u : 0u;
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) reach = max(reach, (uint32_t)__shfl_xor(reach, o, 64));
                    need = reach;
                }
                if (win_ref < 0 || fr != win_ref || need >= WIN || fp < 0 || (uint32_t)fp < win_base) {
                    flush_windows();
                    // the wave's window: entries [win_base, win_base + EDR_WINDOW) of the difference array of the sequence of this pass's first record
                    win_ref = -1;
                    if (fr >= 0 && (uint32_t)fr < st.n_refs && fp >= 0) {
                        if (fr != meta_ref) {
                            const uint64_t eo = st.ref_edits_off[fr], bo = st.ref_bases_off[fr];
                            const uint32_t l = st.ref_len[fr], fl = st.ref_fast_len ? st.ref_fast_len[fr] : l;
                            meta_ref = fr;
                            meta_eoff = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(eo >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)eo);
                            meta_boff = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(bo >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)bo);
                            meta_L = (uint32_t)__builtin_amdgcn_readfirstlane((int)l);
                            meta_E = (uint32_t)__builtin_amdgcn_readfirstlane((int)fl);
                        }
                        if (meta_boff != NO_DEPTH) {
                            win_ref = fr;
                            win_base = (uint32_t)fp & ~3u;
                        }
                    }
                }
            }
            // ---- 2a. (ahead of 1: it does not depend on it) lane = window g = 64 k + lane of the pass's rows (record g / R, window
            // g % R): the 16 bytes of sequence of the first two windows are requested before the records' descriptors are worked out
            // (RAGGED: the pass's first byte -- lane 0's record exists)
            const uint64_t so0 = RAGGED ? ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(cur.so >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)cur.so) : 0ull;
            const uint8_t *const rows = RAGGED ? b.seq + so0 : b.seq + r0 * stride;
            const uint32_t last = (uint32_t)min(n - 1 - r0, (uint64_t)63); // rows of the pass that exist
            uint32_t n_win = 0, w_g = lane; // RAGGED: windows of the pass (known behind step 1); the next window of this lane
            struct Win {
                uint4 sv, rv;
                uint32_t slot, ww, x0, lohi; // record slot, window of the row, window entry of base 0 of the window, compared bases lo | hi << 8 (none: lo >= hi)
            };
            // window g = 64 k + lane is window ww of row rr at byte `off` of the pass's rows: stepped from k to k + 1 without a division
            uint32_t w_rr = (lane * recip) >> 16, w_ww = lane - w_rr * R, w_off = w_rr * stride + 16 * w_ww;
            const uint32_t step_rr = 64 / R, step_ww = 64 - step_rr * R, step_off = step_rr * stride + 16 * step_ww;
            auto begin_win = [&]() -> Win { // the sequence bytes of window (w_rr, w_ww); then on to the next one
                Win w;
                if (RAGGED) { // window w_g of the pass: its record from the map; a lane behind the last window compares nothing (ww = 255)
                    const bool live = w_g < n_win;
                    const uint32_t rr = live ? wmap[w_g] : 0u;
                    const uint2 q = rg[rr];
                    w.slot = rr, w.ww = live ? w_g - (q.y & 0xFFFFu) : 255u;
                    __builtin_memcpy(&w.sv, rows + (live ? q.x + 16u * w.ww : 0u), 16);
                    w_g += 64;
                    return w;
                }
                __builtin_memcpy(&w.sv, rows + (w_rr <= last ? w_off : 0u), 16);
                w.slot = w_rr, w.ww = w_ww;
                w_rr += step_rr, w_ww += step_ww, w_off += step_off;
                if (w_ww >= R) w_ww -= R, w_rr += 1, w_off += stride - 16 * R;
                return w;
            };
            // the reference bytes under it, once its record's descriptor is in LDS
            // (a window without compared bases -- a record that is not on the fast path: descriptor 0; a window behind the read's
            // end -- reads what lies at its place all the same: inside both buffers, and all of it masked away)
            auto finish_win = [&](Win &w) {
                uint2 d = desc[w.slot];
                if (RAGGED && w.ww == 255u) d = make_uint2(0u, 0u), w.ww = 0u; // (nothing compared: v0 = v1 = 0)
                const uint32_t v0 = d.y & 0x1FFu, v1 = (d.y >> 9) & 0x1FFu, b0 = 32 * w.ww;
                __builtin_memcpy(&w.rv, st.ref_bases + (d.x + 16 * w.ww), 16);
                w.x0 = (d.y >> 18) + b0;
                const uint32_t lo = min(v0 > b0 ? v0 - b0 : 0u, 32u), hi = min(v1 > b0 ? v1 - b0 : 0u, 32u);
                w.lohi = lo | hi << 8;
                if (GC) { // the bases of this window inside the record's GC window [off, off + 100): bits 16.. of lohi
                    const uint32_t go = goff[w.slot], ge = go + NGSQ_GC_WINDOW;
                    const uint32_t glo = min(go > b0 ? go - b0 : 0u, 32u), ghi = min(ge > b0 ? ge - b0 : 0u, 32u);
                    w.lohi |= glo << 16 | ghi << 24;
                }
            };
            auto load_win = [&]() -> Win {
                Win w = begin_win();
                finish_win(w);
                return w;
            };
            Win wa, wb, wc;
            if (!RAGGED) {
                wa = begin_win(), wb = wa, wc = wa;
                if (R > 1) wb = begin_win();
            }
            // ---- 1. lane = record
            EdRowCols r = cur;
            uint64_t rid = 0; // GC: the record's identity, what its window's offset is drawn from (requested here, used at the end of the step)
            if (GC) rid = b.record_id ? b.record_id[r0 + lane < n ? r0 + lane : n - 1] : b.first_record_index + r0 + lane;
            if (CIG_OFF) {
                // (the offset loaded a pass ahead with the columns: measured in round 5, the same time)
                const uint64_t cb = b.cigar_off[r0 + lane < n ? r0 + lane : n - 1];
                if (r.n_ops > 0) r.g0 = b.cigar[cb];
                if (r.n_ops > 1) r.g1 = b.cigar[cb + 1];
                if (r.n_ops > 2) r.g2 = b.cigar[cb + 2];
            }
            bool own = false, deferred = false;
            uint32_t P = 0, vv = 0, info = 0;
            if (r.ref >= 0 && (uint32_t)r.ref < st.n_refs && r.pos >= 0 && !(r.flag & 0x404u)) {
                // [clip] M [clip]: read base q lies on 0-based reference position P + q, P = pos - (leading clip)
                uint32_t a = 0, m = 0, z = 0, m2 = 0, ins = 0, del = 0;
                const uint32_t o0 = r.g0 & 15u, o1 = r.g1 & 15u, o2 = r.g2 & 15u;
                bool shape = false;
                if (r.n_ops == 1) shape = o0 == 0u, m = r.g0 >> 4;
                else if (r.n_ops == 2 && o0 == 4u && o1 == 0u) shape = true, a = r.g0 >> 4, m = r.g1 >> 4;
                else if (r.n_ops == 2 && o0 == 0u && o1 == 4u) shape = true, m = r.g0 >> 4, z = r.g1 >> 4;
                else if (r.n_ops == 3 && o0 == 4u && o1 == 0u && o2 == 4u) shape = true, a = r.g0 >> 4, m = r.g1 >> 4, z = r.g2 >> 4;
                // M (I|D|N) M -- an insertion of up to 15 bases, a deletion or a skip of any length (an aligner gives one read in sixteen an
                // insertion or a deletion; a spliced aligner many reads a skip of kilobases): the first M is this record's entry of the
                // window loop, as if the rest were clipped; the second M is compared in a step of its own behind that loop (2c), with the
                // reference shifted by gap - ins.  Where the second M lies beyond the wave's window (a skip), its cover and its mismatches
                // go straight to the arrays.  (Until late in round 5 a deletion of more than 15 bases and every skip were the walk kernel's.)
                else if (r.n_ops == 3 && o0 == 0u && o1 >= 1u && o1 <= 3u && o2 == 0u && (o1 != 1u || (r.g1 >> 4) <= 15u) && (r.g0 >> 4) < 512u && (r.g2 >> 4) &&
                         r.pos >= 16) {
                    shape = true, m = r.g0 >> 4, m2 = r.g2 >> 4;
                    ins = o1 == 1u ? r.g1 >> 4 : 0u, del = o1 != 1u ? r.g1 >> 4 : 0u;
                    z = ins + m2; // (read bases behind the first M)
                }
                const uint64_t e = (uint64_t)r.pos + m + del + m2; // 1-based last position
                const uint64_t i0 = (uint64_t)(uint32_t)r.pos - win_base, i1 = i0 + m + del + m2;
                const bool far2 = m2 && i1 >= WIN; // the second M reaches beyond the window
                // (P >= win_base: a mismatch's window entry is counted from P's)
                // (RAGGED: the record's bytes hold its bases, at most EDG_MAXW windows of them, within 4 GiB of the pass's first byte)
                const bool holds = RAGGED ? ((r.l + 1u) >> 1) <= r.sb && r.sb <= 16u * EDG_MAXW - 1u && r.so - so0 < 0xFFFF0000ull : r.l <= 2 * stride;
                // Round 6: a read whose first M reaches beyond the wave's window (sparse data: at 5x the 64 sorted reads of a pass span
                // 2 k positions, the window holds 1.4 k) stays on the fast path -- its cover goes straight to the difference array and
                // its mismatches beyond the window's end to alts, as a second M beyond the window always did -- as long as its base 0
                // lies within the 14 bits the descriptor has for it.  Until then such reads were the one-record walk's: 6.1 ms of
                // walk + 7.3 ms here per 100 M reads spread over the 3.1 Gbp of a real header (bench.py whole_genome).
                const bool far1 = i0 + m >= WIN; // (then far2 as well: i1 >= i0 + m)
                own = shape && m && (uint64_t)a + m + z == r.l && holds && r.ref == win_ref && (uint32_t)r.pos >= a + win_base && e <= meta_E &&
                      (!far1 || (NGSQ_EDR_FAR && i0 < (1u << 14) - 1024u));
                if (own) {
                    // the record's descriptor for the window lanes: where its base 0 lies in the packed reference (a byte offset
                    // from ref_bases: the copy that starts at base P & 1), compared bases [v0, v1), window entry of base 0
                    P = (uint32_t)r.pos - a;
                    vv = a | (a + m) << 9 | (P - win_base) << 18;
                    P = (uint32_t)(meta_boff + (P >> 1)) + (P & 1u ? odd_delta : 0u);
                    if (EDITS_EXP != 2 && EDITS_EXP != 4) {
                        if (!far1) {
                            atomicAdd(&win[i0], 1u);
                            atomicAdd(&win[i0 + m], 0xFFFFFFFFu);
                        } else {
                            uint32_t *const diff = st.edits + meta_eoff + win_base;
                            atomicAdd(&diff[i0], 1u);
                            atomicAdd(&diff[i0 + m], 0xFFFFFFFFu);
                        }
                        if (m2 && !far2) {
                            atomicAdd(&win[i0 + m + del], 1u);
                            atomicAdd(&win[i1], 0xFFFFFFFFu);
                        } else if (m2) {
                            uint32_t *const diff = st.edits + meta_eoff + win_base;
                            atomicAdd(&diff[i0 + m + del], 1u);
                            atomicAdd(&diff[i1], 0xFFFFFFFFu);
                        }
                    }
                    top = max(top, (uint32_t)(far2 || far1 ? WIN - 1 : i1)); // (far: the mismatches below the window's end are tallied in it)
                    // what step 2c needs of the record: first M's length, insertion, parity of P; the gap in an array of its own
                    if (m2) {
                        info = m | ins << 9 | ((uint32_t)r.pos & 1u) << 13; // (m >= 1: never 0; 14 bits, kept above the 10 bits of the edit count)
                        gapl[lane] = del;
                    }
                } else {
                    deferred = true;
                }
            }
            bool gc_take = false;
            if (GC) { // gc_content.rs:41-45 (duplicate | secondary are ignored), :59-62 (shorter than the window), :68-74 (the offset)
                const bool live = r0 + lane < n;
                const bool ign_f = live && (r.flag & 0x500u), ign_s = live && !ign_f && r.l < NGSQ_GC_WINDOW;
                gc_take = live && !ign_f && !ign_s;
                goff[lane] = (uint16_t)(gc_take && r.l <= 2 * stride ? ngsq_gc_offset_fn(st.gc_seed, rid, r.l) : GC_NONE);
                if (!(lane & 1u)) gacc[lane >> 1] = 0; // (read by step 3 of the previous pass: LDS operations of a wave execute in order)
                const u64 mf = __ballot(ign_f), ms = __ballot(ign_s);
                if (lane == 0 && (mf | ms)) {
                    if (mf) atomicAdd(&s_gign[0], (uint32_t)__popcll(mf));
                    if (ms) atomicAdd(&s_gign[1], (uint32_t)__popcll(ms));
                }
            }
            const u64 imask = __ballot(info != 0u); // the pass's records with a second M
            if (info) ilist[__builtin_amdgcn_mbcnt_hi((uint32_t)(imask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)imask, 0u))] = (uint8_t)lane;
            {   // (a record the fast path does not take is marked in the launch's bitmap: k_edits_walk does it afterwards)
                const u64 dm = __ballot(deferred);
                if (lane == 0) defer_bits[r0 >> 6] = dm;
            }
            desc[lane] = make_uint2(P, vv);
            red[lane] = info << 10;
            if (RAGGED) { // the pass's windows: a prefix sum of the records' counts, each record's run of the map
                const uint32_t wi = own ? (r.sb + 15u) >> 4 : 0u;
                uint32_t incl = wi;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const uint32_t up = (uint32_t)__shfl_up(incl, o, 64);
                    if (lane >= (uint32_t)o) incl += up;
                }
                const uint32_t first = incl - wi;
                n_win = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                rg[lane] = make_uint2((uint32_t)(r.so - so0), first | wi << 16);
                for (uint32_t k = 0; k < wi; k++) wmap[first + k] = (uint8_t)lane;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const uint32_t n_steps = RAGGED ? (n_win + 63u) >> 6 : R; // steps of 64 windows
            if (RAGGED) {
                wa = begin_win(), wb = wa, wc = wa;
                if (n_steps > 1) wb = begin_win();
            }
            // ---- 2b. lane = window: 16 bytes of sequence XOR 16 of the reference
            auto compare_win = [&](const Win &w, const bool far = false) { // far (2c only): the window may lie beyond the wave's LDS window
                if (GC) { // the window's share of its record's GC window
                    const uint32_t glo = (w.lohi >> 16) & 0xFFu, ghi = w.lohi >> 24;
                    if (ghi > glo) {
                        constexpr uint32_t M1 = 0x11111111u;
                        const uint32_t v[4] = {w.sv.x, w.sv.y, w.sv.z, w.sv.w};
                        uint32_t gcw = 0, atw = 0; // bit 4 q + d <-> nibble q of dword d, the order of s_tmask
#pragma unroll
                        for (uint32_t d = 0; d < 4; d++) {
                            const uint32_t x = v[d], y = x >> 1, z = x >> 2, u = x >> 3;
                            gcw |= ((y ^ z) & ~(x | u) & M1) << d; // nibble == 0010 (C) or 0100 (G)
                            atw |= ((x ^ u) & ~(y | z) & M1) << d; // nibble == 0001 (A) or 1000 (T)
                        }
                        const uint32_t gm = s_tmask[ghi] & ~s_tmask[glo];
                        const uint32_t add = (uint32_t)__popc(gcw & gm) | (uint32_t)__popc(atw & gm) << 7;
                        atomicAdd(&gacc[w.slot >> 1], add << (16u * (w.slot & 1u)));
                    }
                }
                // the mismatching nibbles of the four dwords in one word: bit 4 q + d <-> nibble q of dword d = base 8 d + (q ^ 1) of
                // the window; the window's compared bases are [lo, hi): two table masks in that bit order
                const uint32_t x0 = w.sv.x ^ w.rv.x, x1 = w.sv.y ^ w.rv.y, x2 = w.sv.z ^ w.rv.z, x3 = w.sv.w ^ w.rv.w;
                const uint32_t n0 = (((x0 & 0x77777777u) + 0x77777777u) | x0) & 0x88888888u, n1 = (((x1 & 0x77777777u) + 0x77777777u) | x1) & 0x88888888u,
                               n2 = (((x2 & 0x77777777u) + 0x77777777u) | x2) & 0x88888888u, n3 = (((x3 & 0x77777777u) + 0x77777777u) | x3) & 0x88888888u;
                uint32_t t = ((n0 >> 3) | (n1 >> 2) | (n2 >> 1) | n3) & s_tmask[(w.lohi >> 8) & 0xFFu] & ~s_tmask[w.lohi & 0xFFu];
                if (t) {
                    atomicAdd(&red[w.slot], (uint32_t)__popc(t));
                    while (t && EDITS_EXP != 1 && EDITS_EXP != 4) {
                        const uint32_t bit = (uint32_t)__builtin_ctz(t);
                        t &= t - 1;
                        const uint32_t x0e = w.x0 + 8 * (bit & 3u) + ((bit >> 2) ^ 1u); // window entry of the 0-based position
                        if (far && x0e >= WIN) atomicAdd(st.edits + meta_eoff + win_base + ((uint64_t)meta_L + 1) + 1 + x0e, 1u); // alts[1 + position]
                        else atomicAdd(&altw[x0e >> 1], 1u << (16u * (x0e & 1u)));
                    }
                }
            };
            // two windows' bytes are in flight while one is compared (three slots used in turn; the kernel's waves are limited
            // by LDS to four per SIMD, which leaves 128 registers each)
            finish_win(wa);
            if (n_steps > 1) finish_win(wb);
            if (it + 1 < ppw) cur = load_cols(r0 + 64); // in flight while this pass is compared (behind the end: the last record again)
#pragma unroll 1
            for (uint32_t k = 0; k < n_steps && EDITS_EXP != 3; k += 3) {
                if (k + 2 < n_steps) wc = load_win();
                compare_win(wa, NGSQ_EDR_FAR != 0);
                if (k + 1 >= n_steps) break;
                if (k + 3 < n_steps) wa = load_win();
                compare_win(wb, NGSQ_EDR_FAR != 0);
                if (k + 2 >= n_steps) break;
                if (k + 4 < n_steps) wb = load_win();
                compare_win(wc, NGSQ_EDR_FAR != 0);
            }
            // ---- 2c. the second M of the records that have one: lane = window ww of the e-th such record (its row once more, the
            // reference del - ins bases further on -- the other packed copy when that is odd)
            if (imask && EDITS_EXP != 3) {
                const uint32_t n_ind = (uint32_t)__popcll(imask);
                for (uint32_t g = lane; g < n_ind * (RAGGED ? EDG_MAXW : R); g += 64) {
                    const uint32_t ei = RAGGED ? g / EDG_MAXW : (g * recip) >> 16, ww = RAGGED ? g % EDG_MAXW : g - ei * R, rr = ilist[ei];
                    if (RAGGED && ww >= rg[rr].y >> 16) continue; // (a window behind the record's last)
                    const uint2 d = desc[rr];
                    const uint32_t inf = red[rr] >> 10;
                    const uint32_t m1 = inf & 0x1FFu, ins = (inf >> 9) & 15u, del = gapl[rr], ppar = (inf >> 13) & 1u;
                    const uint32_t l_rd = RAGGED ? b.l_seq[r0 + rr] : min(b.l_seq[r0 + rr], 2u * stride);
                    const int32_t shift = (int32_t)del - (int32_t)ins, t = (int32_t)ppar + shift; // P2 = P + shift
                    const uint32_t off2 = (uint32_t)((int32_t)(d.x - (ppar ? odd_delta : 0u)) + (t >> 1)) + ((t & 1) ? odd_delta : 0u);
                    const uint32_t b0 = 32u * ww, v0 = m1 + ins, v1 = l_rd;
                    Win w;
                    __builtin_memcpy(&w.sv, rows + (RAGGED ? rg[rr].x : rr * stride) + 16u * ww, 16);
                    __builtin_memcpy(&w.rv, st.ref_bases + (off2 + 16u * ww), 16);
                    w.slot = rr, w.ww = ww;
                    w.x0 = (uint32_t)((int32_t)(d.y >> 18) + shift + (int32_t)b0);
                    const uint32_t lo = min(v0 > b0 ? v0 - b0 : 0u, 32u), hi = min(v1 > b0 ? v1 - b0 : 0u, 32u);
                    w.lohi = lo | hi << 8;
                    compare_win(w, true);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // ---- 3. lane = record: gc_content.rs:91-96 (the bin is the count itself), then edits.rs:296-300
            if (GC && gc_take) {
                const uint32_t v = (gacc[lane >> 1] >> (16u * (lane & 1u))) & 0x3FFFu, gc = v & 0x7Fu, at = v >> 7;
                atomicAdd(&s_ghist[gc >> 1], 1u << (16u * (gc & 1u)));
                atomicAdd(&s_ahist[at >> 1], 1u << (16u * (at & 1u)));
            }
            if (own) {
                const uint32_t edits = red[lane] & 0x3FFu;
                if (edits > 512u) c_too_many += 1;
                else atomicAdd(((r.flag & 0x40u) ? s_h1 : s_h2) + (edits >> 1), 1u << (16u * (edits & 1u)));
            }
        }
    }
    flush_windows();
    flush_hist();
    {
        const uint32_t r = ed_wave_sum(c_too_many);
        if (lane == 0 && r) atomicAdd(&s_acc[3], (u64)r);
    }
    __syncthreads();
    if (tid == 3 && s_acc[3]) atomicAdd(&st.counters[C_ERR + E_EDITS_TOO_MANY], s_acc[3]);
}

// The records the fast paths left.  A wave takes 32 words of the bitmap (2048 records) at a time, lists the marked ones in LDS
// (a word per lane, a prefix sum of the population counts) and walks them 64 at a time, a lane per marked record.  (Until the
// end of round 4 a wave took ONE word and a lane each of its marked records: with an aligner's 6 % of reads with an insertion or
// a deletion four lanes in 64 worked, and those reads cost three times what all the others did: 12.9 ms per 100 M reads
// against 3.4 for reads without them.)
constexpr uint32_t EDW_WORDS = 32, EDW_SPAN = EDW_WORDS * 64; // bitmap words / records per wave and round (4 KB of list per wave: eight blocks per CU)
__global__ __launch_bounds__(256) void k_edits_walk(DeviceState st, DeviceBatch b, const u64 *__restrict__ defer_bits) {
    __shared__ uint32_t s_h1[NGSQ_EDITS_BINS], s_h2[NGSQ_EDITS_BINS];
    __shared__ uint16_t s_list[4][EDW_SPAN];
    __shared__ u64 s_acc[4];
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    for (uint32_t k = tid; k < NGSQ_EDITS_BINS; k += 256) s_h1[k] = s_h2[k] = 0;
    if (tid < 4) s_acc[tid] = 0;
    __syncthreads();
    uint32_t c[4] = {0, 0, 0, 0};
    uint16_t *const list = s_list[tid >> 6];
    const uint64_t n_words = (b.n + 63) / 64, n_spans = (n_words + EDW_WORDS - 1) / EDW_WORDS, waves = (uint64_t)gridDim.x * 4;
    for (uint64_t sp = (uint64_t)blockIdx.x * 4 + (tid >> 6); sp < n_spans; sp += waves) {
        const uint64_t w = sp * EDW_WORDS + lane;
        u64 word = lane < EDW_WORDS && w < n_words ? defer_bits[w] : 0ull;
        if (w + 1 == n_words && (b.n & 63ull)) word &= (1ull << (b.n & 63ull)) - 1ull; // (bits behind the last record)
        const uint32_t cnt = (uint32_t)__popcll(word);
        uint32_t incl = cnt; // inclusive prefix sum over the lanes
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)incl, o, 64);
            if ((int)lane >= o) incl += up;
        }
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        if (!total) continue;
        uint32_t at = incl - cnt;
        while (word) { // this lane's marked records: their offsets in the span
            list[at++] = (uint16_t)(lane * 64u + (uint32_t)__builtin_ctzll(word));
            word &= word - 1;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (uint32_t k = lane; k < total; k += 64) {
            const uint64_t i = sp * EDW_SPAN + list[k];
            uint32_t edits = 0;
            const uint32_t err = ed_walk_record(st, b, i, &edits);
            if (err) c[err - 1] += 1;
            else if (edits != 0xFFFFFFFFu) atomicAdd(edits >> 31 ? &s_h1[edits & 0x7FFFFFFFu] : &s_h2[edits & 0x7FFFFFFFu], 1u);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); // the list is rewritten for the next span
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    for (uint32_t k = tid; k < NGSQ_EDITS_BINS; k += 256) {
        uint32_t v = s_h1[k];
        if (v) atomicAdd(&st.counters[st.off_edits1 + k], (u64)v);
        v = s_h2[k];
        if (v) atomicAdd(&st.counters[st.off_edits2 + k], (u64)v);
    }
    const uint32_t idx[4] = {C_ERR + E_EDITS_BAD_REF, C_ERR + E_EDITS_SHORT, C_ERR + E_EDITS_NOT_CONSUMED, C_ERR + E_EDITS_TOO_MANY};
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t r = ed_wave_sum(c[k]);
        if (lane == 0 && r) atomicAdd(&s_acc[k], (u64)r);
    }
    __syncthreads();
    if (tid < 4 && s_acc[tid]) atomicAdd(&st.counters[idx[tid]], s_acc[tid]);
}

// ---------------------------------------------------------------------------
// The reference FASTA on the device: one 4-bit code per byte in -> two packed copies out (see the head of this file).
// Copy E: byte k = code[2k] << 4 | code[2k+1]; copy O: byte k = code[2k+1] << 4 | code[2k+2]; codes behind the end are 0.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pack_reference(const uint8_t *__restrict__ codes, uint64_t len, uint8_t *__restrict__ even,
                                                         uint8_t *__restrict__ odd, uint64_t n_bytes, unsigned long long *bad) {
    for (uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x; k < n_bytes; k += (uint64_t)gridDim.x * 256) {
        const uint32_t a = 2 * k < len ? codes[2 * k] : 0u, m = 2 * k + 1 < len ? codes[2 * k + 1] : 0u, z = 2 * k + 2 < len ? codes[2 * k + 2] : 0u;
        if ((a | m | z) > 15u) atomicAdd(bad, 1ull); // not a 4-bit code (include/ngsq.h ngsq_config.ref_bases)
        even[k] = (uint8_t)((a & 15u) << 4 | (m & 15u));
        odd[k] = (uint8_t)((m & 15u) << 4 | (z & 15u));
    }
}

// ---------------------------------------------------------------------------
// Edits teardown (edits.rs:305-344).  The slot of refs holds the difference array of the `M` cover until here:
//   k_edits_chunk_sums   sum of the entries of every 4096-entry chunk
//   k_edits_super_sums   sum of every 256 consecutive chunk sums (a chunk's carry = the super sums and the chunk sums in front of it)
//   k_edits_refs         chunks [c0, c1): cover[p] = sum of the entries [0, p), refs[p] = cover[p] - alts[p] in place, and the
//                        VAF histogram of the positions with refs + alts > 0 -- f32 arithmetic exactly as edits.rs:331-335:
//                        alts as f32 / total as f32, * 100.0, truncated
// A sharded run converts and tallies a slice of every sequence's chunks per rank (ngsq_exchange); the rest of a sequence is
// converted, without the tally, when somebody asks for its positions (ngsq_get_edits_positions).
// ---------------------------------------------------------------------------
constexpr uint32_t EDC = 4096; // entries per teardown chunk

__device__ __forceinline__ void ed_chunk_sums_body(const uint32_t *__restrict__ diff, uint64_t n_entries, uint32_t *__restrict__ sums, const u64 *touched,
                                                   uint32_t bidx) {
    __shared__ uint32_t s_w[4];
    if (touched && !*touched) return; // (uniform) nothing was written for this sequence
    const uint64_t base = (uint64_t)bidx * EDC;
    uint32_t t = 0;
#pragma unroll
    for (uint32_t k = 0; k < EDC / 1024; k++) {
        const uint64_t i = base + (uint64_t)k * 1024 + threadIdx.x * 4;
        if (i + 4 <= n_entries) {
            const uint4 v = *reinterpret_cast<const uint4 *>(diff + i);
            t += v.x + v.y + v.z + v.w;
        } else {
            for (uint32_t q = 0; q < 4; q++)
                if (i + q < n_entries) t += diff[i + q];
        }
    }
    t = ed_wave_sum(t);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) sums[bidx] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}
__global__ __launch_bounds__(256) void k_edits_chunk_sums(const uint32_t *__restrict__ diff, uint64_t n_entries, uint32_t *__restrict__ sums, const u64 *touched) {
    ed_chunk_sums_body(diff, n_entries, sums, touched, blockIdx.x);
}

// The carry of a chunk (the sum of every entry in front of it) is never written down: behind the chunks' sums lie the sums of every
// EDS consecutive chunks, and a block of k_edits_refs adds up the (at most a few hundred) super sums in front of its chunk's group and
// the (at most EDS - 1) chunk sums in front of it inside the group.  (Until round 5 one block scanned the sixty thousand chunk sums of
// a chromosome in place, sixty sequential loads per thread: 0.09 ms of a 0.45 ms teardown.)
constexpr uint32_t EDS = 256; // chunks per super sum
__device__ __forceinline__ void ed_super_sums_body(const uint32_t *__restrict__ sums, uint32_t n, uint32_t *__restrict__ supers, const u64 *touched, uint32_t bidx) {
    __shared__ uint32_t s_w[4];
    if (touched && !*touched) return;
    const uint32_t i = bidx * EDS + threadIdx.x;
    uint32_t t = ed_wave_sum(i < n ? sums[i] : 0u);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) supers[bidx] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}
__global__ __launch_bounds__(256) void k_edits_super_sums(const uint32_t *__restrict__ sums, uint32_t n, uint32_t *__restrict__ supers, const u64 *touched) {
    ed_super_sums_body(sums, n, supers, touched, blockIdx.x);
}

template <bool WRITE>
__device__ __forceinline__ void ed_refs_body(uint32_t *__restrict__ refs, const uint32_t *__restrict__ alts, uint64_t n_entries,
                                             const uint32_t *__restrict__ carry, uint32_t chunk0, uint32_t chunk1, u64 *vaf_hist, const u64 *touched,
                                             uint32_t bidx, uint32_t gdim) {
    __shared__ uint32_t s_h[NGSQ_VAF_BINS];
    __shared__ uint32_t s_w[4], s_c[4];
    if (touched && !*touched) return;
    if (threadIdx.x < NGSQ_VAF_BINS) s_h[threadIdx.x] = 0;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t *const supers = carry + (uint32_t)((n_entries + EDC - 1) / EDC); // (`carry`: the chunks' sums, the super sums behind them)
    uint32_t zero_bin = 0;       // covered positions of this thread with alts == 0 (VAF 0.0 -> bin 0)
    __syncthreads();
    // a block takes every gridDim.x-th chunk and adds its histogram to the global one ONCE (a block per chunk meant sixty thousand
    // blocks per chromosome adding to the same bin: same-address atomics at the L2, ~9 ns each)
#pragma unroll 1
    for (uint32_t chunk = chunk0 + bidx; chunk < chunk1; chunk += gdim) {
    const uint64_t base = (uint64_t)chunk * EDC;
    uint32_t run; // sum of every entry in front of the chunk
    {
        const uint32_t grp = chunk / EDS, in_grp = chunk - grp * EDS;
        uint32_t t = threadIdx.x < in_grp ? carry[grp * EDS + threadIdx.x] : 0u;
        for (uint32_t i = threadIdx.x; i < grp; i += 256) t += supers[i];
        t = ed_wave_sum(t);
        if (lane == 0) s_c[wave] = t;
        __syncthreads(); // (s_c is rewritten for the next chunk behind the barriers of this chunk's steps)
        run = s_c[0] + s_c[1] + s_c[2] + s_c[3];
    }
#pragma unroll 1
    for (uint32_t k = 0; k < EDC / 1024; k++) {
        const uint64_t i = base + (uint64_t)k * 1024 + threadIdx.x * 4;
        uint32_t d[4] = {0, 0, 0, 0}, a[4] = {0, 0, 0, 0};
        if (i + 4 <= n_entries) {
            const uint4 dv = *reinterpret_cast<const uint4 *>(refs + i);
            uint4 av; // (alts starts L + 1 entries behind refs: any dword alignment)
            __builtin_memcpy(&av, alts + i, 16);
            d[0] = dv.x, d[1] = dv.y, d[2] = dv.z, d[3] = dv.w;
            a[0] = av.x, a[1] = av.y, a[2] = av.z, a[3] = av.w;
        } else {
            for (uint32_t q = 0; q < 4; q++)
                if (i + q < n_entries) d[q] = refs[i + q], a[q] = alts[i + q];
        }
        // exclusive prefix of the 1024 entries of this step: thread, wave (DPP-free shuffles), block
        const uint32_t mine = d[0] + d[1] + d[2] + d[3];
        uint32_t inc = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)inc, o, 64);
            if ((int)lane >= o) inc += up;
        }
        if (lane == 63) s_w[wave] = inc;
        __syncthreads();
        uint32_t before = run + inc - mine;
        for (uint32_t w = 0; w < wave; w++) before += s_w[w];
        const uint32_t step_total = s_w[0] + s_w[1] + s_w[2] + s_w[3];
        uint32_t out[4];
        uint32_t cov = before; // cover of position i = sum of the entries [0, i)
#pragma unroll
        for (uint32_t q = 0; q < 4; q++) {
            out[q] = cov - a[q];
            if (i + q < n_entries && cov && vaf_hist) {
                // (a covered position without a mismatch -- 99 in 100 -- is bin 0: counted in a register; 64 lanes adding to that
                // one LDS word serialise, and that, not the 8 bytes per position, was most of the teardown's time until round 5)
                if (a[q] == 0u) {
                    zero_bin += 1;
                } else {
                    const float vaf = __fdiv_rn((float)a[q], (float)cov); // total = refs + alts = cover
                    atomicAdd(&s_h[(uint32_t)__fmul_rn(vaf, 100.0f)], 1u);
                }
            }
            cov += d[q];
        }
        if (!WRITE) {
        } else if (i + 4 <= n_entries) {
            *reinterpret_cast<uint4 *>(refs + i) = make_uint4(out[0], out[1], out[2], out[3]);
        } else {
            for (uint32_t q = 0; q < 4; q++)
                if (i + q < n_entries) refs[i + q] = out[q];
        }
        run += step_total;
        __syncthreads();
    }
    }
    if (vaf_hist) {
        zero_bin = ed_wave_sum(zero_bin);
        if (lane == 0 && zero_bin) atomicAdd(&s_h[0], zero_bin);
        __syncthreads();
        if (threadIdx.x < NGSQ_VAF_BINS) {
            const uint32_t v = s_h[threadIdx.x];
            if (v) atomicAdd(&vaf_hist[threadIdx.x], (u64)v);
        }
    }
}
template <bool WRITE>
__global__ __launch_bounds__(256) void k_edits_refs(uint32_t *__restrict__ refs, const uint32_t *__restrict__ alts, uint64_t n_entries,
                                                     const uint32_t *__restrict__ carry, uint32_t chunk0, uint32_t chunk1, u64 *vaf_hist, const u64 *touched) {
    ed_refs_body<WRITE>(refs, alts, n_entries, carry, chunk0, chunk1, vaf_hist, touched, blockIdx.x, gridDim.x);
}

// ---- the teardown of EVERY sequence in three launches (round 6).  One launch per sequence and step was 585 launches for the 195
// sequences of a real header -- 9.6 ms of launch latency in a finalize whose work is 4 ms -- and two launches per sequence even for
// the sequences no read lay on.  A block finds its sequence in a table of first-block numbers.
__device__ __forceinline__ uint32_t ed_seq_of_block(const uint32_t *__restrict__ first, uint32_t n_seq, uint32_t b) {
    uint32_t lo = 0, hi = n_seq; // first[lo] <= b < first[hi]
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (first[mid] <= b) lo = mid;
        else hi = mid;
    }
    return lo;
}
__global__ __launch_bounds__(256) void k_edits_chunk_sums_all(const EditsSeq *__restrict__ seqs, uint32_t n_seq, const uint32_t *__restrict__ first,
                                                               uint32_t *edits, uint32_t *carry, const u64 *touched) {
    const uint32_t q = ed_seq_of_block(first, n_seq, blockIdx.x);
    const EditsSeq e = seqs[q];
    ed_chunk_sums_body(edits + e.edits_off, e.n_entries, carry + e.carry_off, touched + e.ref, blockIdx.x - first[q]);
}
__global__ __launch_bounds__(256) void k_edits_super_sums_all(const EditsSeq *__restrict__ seqs, uint32_t n_seq, const uint32_t *__restrict__ first,
                                                               uint32_t *carry, const u64 *touched) {
    const uint32_t q = ed_seq_of_block(first, n_seq, blockIdx.x);
    const EditsSeq e = seqs[q];
    const uint32_t nc = (uint32_t)((e.n_entries + EDC - 1) / EDC);
    ed_super_sums_body(carry + e.carry_off, nc, carry + e.carry_off + nc, touched + e.ref, blockIdx.x - first[q]);
}
__global__ __launch_bounds__(256) void k_edits_refs_all(const EditsSeq *__restrict__ seqs, uint32_t n_seq, const uint32_t *__restrict__ first,
                                                         uint32_t *edits, const uint32_t *carry, u64 *vaf_hist, const u64 *touched) {
    const uint32_t q = ed_seq_of_block(first, n_seq, blockIdx.x);
    const EditsSeq e = seqs[q];
    uint32_t *const refs = edits + e.edits_off;
    ed_refs_body<false>(refs, refs + e.n_entries, e.n_entries, carry + e.carry_off, e.chunk0, e.chunk1, vaf_hist, touched + e.ref, blockIdx.x - first[q],
                        first[q + 1] - first[q]);
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
// fixed-pitch rows of reads of up to 160 bases: a lane per 16-byte window (k_edits_rows); longer rows and the offsets layout: a lane
// per record (the window lanes address the packed reference by 32-bit byte offsets: 2 x 4 G bases; beyond that the lane-per-record kernel)
static bool edits_windows_ok(const DeviceState &st, const DeviceBatch &b) { // (what both window-lane variants need)
    static const bool per_record = getenv("NGSQ_EDITS_PER_RECORD") && atoi(getenv("NGSQ_EDITS_PER_RECORD")); // A/B measurements
    return !per_record && 2 * (uint64_t)(st.ref_bases_odd - st.ref_bases) + 256 < (1ull << 32) && (b.cigar_off || b.cigar_stride >= 1);
}
static bool edits_rows_ok(const DeviceState &st, const DeviceBatch &b) {
    const uint32_t R = (b.seq_stride + 15) / 16;
    return !b.seq_off && R >= 1 && R <= ED_NW && edits_windows_ok(st, b);
}
static bool edits_ragged_ok(const DeviceState &st, const DeviceBatch &b) { return b.seq_off && edits_windows_ok(st, b); }

bool edits_can_take_gc(const DeviceState &st, const DeviceBatch &b) {
    static const bool off = getenv("NGSQ_EDITS_NO_GC") && atoi(getenv("NGSQ_EDITS_NO_GC")); // A/B measurements: k_gc as a kernel of its own
    return !off && b.n && st.ref_bases && edits_rows_ok(st, b);
}

hipError_t launch_edits(const LaunchInfo &li, const DeviceState &st, const DeviceBatch &b, unsigned long long *defer_bits, bool with_gc, hipStream_t s) {
    if (!b.n) return hipSuccess;
    static int per_cu = -1;
    if (per_cu < 0) {
        const char *e = getenv("NGSQ_EDITS_BLOCKS_PER_CU"); // measurement aid
        per_cu = e && atoi(e) > 0 ? atoi(e) : 8;
    }
    uint64_t g = (b.n + ED_TILE - 1) / ED_TILE;
    const uint64_t cap = (uint64_t)li.n_cu * (uint32_t)per_cu;
    if (g > cap) g = cap;
    const uint32_t R = (b.seq_stride + 15) / 16;
    const uint32_t gr = (uint32_t)std::min<uint64_t>((b.n + EDR_TILE - 1) / EDR_TILE, cap);
    const bool rows_ok = edits_rows_ok(st, b);
    if (with_gc && !rows_ok) return hipErrorInvalidValue; // (the caller asked edits_can_take_gc)
    const uint32_t recip = R ? 65536u / R + 1u : 0u;
    if (rows_ok && b.cigar_off) {
        if (with_gc) hipLaunchKernelGGL((k_edits_rows<true, true, false>), dim3(gr), dim3(ED_THREADS), 0, s, st, b, R, recip, defer_bits);
        else hipLaunchKernelGGL((k_edits_rows<true, false, false>), dim3(gr), dim3(ED_THREADS), 0, s, st, b, R, recip, defer_bits);
    } else if (rows_ok) {
        if (with_gc) hipLaunchKernelGGL((k_edits_rows<false, true, false>), dim3(gr), dim3(ED_THREADS), 0, s, st, b, R, recip, defer_bits);
        else hipLaunchKernelGGL((k_edits_rows<false, false, false>), dim3(gr), dim3(ED_THREADS), 0, s, st, b, R, recip, defer_bits);
    } else if (edits_ragged_ok(st, b)) { // (R, recip: unused)
        if (b.cigar_off) hipLaunchKernelGGL((k_edits_rows<true, false, true>), dim3(gr), dim3(ED_THREADS), 0, s, st, b, 1u, 65536u, defer_bits);
        else hipLaunchKernelGGL((k_edits_rows<false, false, true>), dim3(gr), dim3(ED_THREADS), 0, s, st, b, 1u, 65536u, defer_bits);
    } else {
        hipLaunchKernelGGL(k_edits, dim3((uint32_t)g), dim3(ED_THREADS), 0, s, st, b, defer_bits);
    }
    hipLaunchKernelGGL(k_edits_walk, dim3((uint32_t)std::min<uint64_t>((b.n + 4 * EDW_SPAN - 1) / (4 * EDW_SPAN), (uint64_t)li.n_cu * 8)), dim3(256), 0, s, st, b, defer_bits);
    return hipGetLastError();
}

hipError_t launch_pack_reference(const LaunchInfo &li, const uint8_t *codes, uint64_t len, uint8_t *even, uint8_t *odd, uint64_t n_bytes,
                                 unsigned long long *bad, hipStream_t s) {
    if (!n_bytes) return hipSuccess;
    const uint32_t grid = (uint32_t)std::min<uint64_t>((n_bytes + 255) / 256, (uint64_t)li.n_cu * 16);
    hipLaunchKernelGGL(k_pack_reference, dim3(grid), dim3(256), 0, s, codes, len, even, odd, n_bytes, bad);
    return hipGetLastError();
}

// tables: seqs [n_seq], first_sums / first_supers / first_refs [n_seq + 1] each (device memory, filled by the host: context.cpp)
hipError_t launch_edits_teardown_all(const EditsSeq *seqs, uint32_t n_seq, const uint32_t *first_sums, uint32_t n_sums, const uint32_t *first_supers,
                                     uint32_t n_supers, const uint32_t *first_refs, uint32_t n_refs_blocks, uint32_t *edits, uint32_t *carry,
                                     unsigned long long *vaf_hist, const unsigned long long *touched, hipStream_t s) {
    if (!n_seq) return hipSuccess;
    if (n_sums) hipLaunchKernelGGL(k_edits_chunk_sums_all, dim3(n_sums), dim3(256), 0, s, seqs, n_seq, first_sums, edits, carry, touched);
    if (n_supers) hipLaunchKernelGGL(k_edits_super_sums_all, dim3(n_supers), dim3(256), 0, s, seqs, n_seq, first_supers, carry, touched);
    if (n_refs_blocks) hipLaunchKernelGGL(k_edits_refs_all, dim3(n_refs_blocks), dim3(256), 0, s, seqs, n_seq, first_refs, edits, carry, vaf_hist, touched);
    return hipGetLastError();
}

uint64_t edits_teardown_chunks(uint64_t n_entries) { return (n_entries + EDC - 1) / EDC; }
uint64_t edits_teardown_carry_words(uint64_t n_entries) { // the chunks' sums + the super sums behind them
    const uint64_t n = edits_teardown_chunks(n_entries);
    return n + (n + EDS - 1) / EDS;
}

hipError_t launch_edits_chunk_sums(const uint32_t *diff, uint64_t n_entries, uint32_t *sums, const unsigned long long *touched, hipStream_t s) {
    const uint64_t n = edits_teardown_chunks(n_entries);
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(k_edits_chunk_sums, dim3((uint32_t)n), dim3(256), 0, s, diff, n_entries, sums, touched);
    hipLaunchKernelGGL(k_edits_super_sums, dim3((uint32_t)((n + EDS - 1) / EDS)), dim3(256), 0, s, sums, (uint32_t)n, sums + n, touched);
    return hipGetLastError();
}

hipError_t launch_edits_refs(uint32_t *refs, const uint32_t *alts, uint64_t n_entries, const uint32_t *carry, uint64_t chunk0, uint64_t chunk1,
                             unsigned long long *vaf_hist, const unsigned long long *touched, bool write_refs, hipStream_t s) {
    if (chunk1 <= chunk0) return hipSuccess;
    const uint32_t grid = (uint32_t)std::min<uint64_t>(chunk1 - chunk0, 4096);
    if (write_refs)
        hipLaunchKernelGGL(k_edits_refs<true>, dim3(grid), dim3(256), 0, s, refs, alts, n_entries, carry, (uint32_t)chunk0, (uint32_t)chunk1, vaf_hist, touched);
    else
        hipLaunchKernelGGL(k_edits_refs<false>, dim3(grid), dim3(256), 0, s, refs, alts, n_entries, carry, (uint32_t)chunk0, (uint32_t)chunk1, vaf_hist, touched);
    return hipGetLastError();
}

} // namespace ngsq
