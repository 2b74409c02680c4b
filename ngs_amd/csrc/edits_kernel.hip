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

#include "kernels.h"

namespace ngsq {

typedef unsigned long long u64;

namespace {

constexpr uint32_t ED_THREADS = 256;
constexpr uint32_t ED_PASSES = 4;                      // records per thread and tile: lane + 64 * pass of the wave's 256 consecutive records
constexpr uint32_t ED_WAVE_TILE = 64 * ED_PASSES;
constexpr uint32_t ED_TILE = ED_THREADS * ED_PASSES;   // records per block and tile
constexpr uint32_t ED_WINDOW = 2048;                   // entries of the difference array in one wave's LDS window
constexpr uint32_t ED_CHUNKS = 5;                      // 16-byte pieces of packed sequence compared per round (160 bases)
constexpr uint32_t ED_FAST_MAX = 512;                  // longest read of the one-`M` path (a 64-bit map of its dwords)

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
__device__ __forceinline__ uint32_t ld32(const uint8_t *p) {
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}
// the eight nibbles that start at nibble offset `q` of the packed run at `base`, first nibble in bits 31..28
__device__ __forceinline__ uint32_t nibbles8(const uint8_t *base, uint64_t q) {
    const u64 v = ld64(base + (q >> 1));
    const uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    const u64 be = (u64)__builtin_bswap32(lo) << 32 | __builtin_bswap32(hi); // bytes in stream order from the top
    return (uint32_t)((be << ((q & 1u) * 4u)) >> 32);
}

} // namespace

__global__ __launch_bounds__(ED_THREADS) void k_edits(DeviceState st, DeviceBatch b) {
    NGSQ_FOREGROUND_WAVE();
    __shared__ uint32_t s_h1[NGSQ_EDITS_BINS], s_h2[NGSQ_EDITS_BINS];
    __shared__ uint32_t s_win[(ED_THREADS / 64) * ED_WINDOW];
    __shared__ u64 s_acc[4];
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    for (uint32_t i = tid; i < NGSQ_EDITS_BINS; i += ED_THREADS) s_h1[i] = s_h2[i] = 0;
    for (uint32_t i = tid; i < (ED_THREADS / 64) * ED_WINDOW; i += ED_THREADS) s_win[i] = 0;
    if (tid < 4) s_acc[tid] = 0;
    __syncthreads();
    uint32_t c[4] = {0, 0, 0, 0}; // bad_ref, record_short, not_consumed, too_many
    uint32_t *const win = s_win + (tid >> 6) * ED_WINDOW;

    // facts of the sequence the wave's window is anchored on (reloaded only when it changes: scalar registers)
    int32_t meta_ref = -1;
    uint64_t meta_eoff = NO_DEPTH, meta_boff = NO_DEPTH;
    uint32_t meta_L = 0;

    const uint64_t n_tiles = (b.n + ED_TILE - 1) / ED_TILE;
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint64_t w0 = tile * ED_TILE + (uint64_t)(tid >> 6) * ED_WAVE_TILE; // the wave's first record
        // ---- the wave's window: entries [win_base, win_base + ED_WINDOW) of the difference array of the sequence of its first record
        int32_t win_ref = -1;
        uint32_t win_base = 0, top = 0;
        if (w0 < b.n) {
            const int32_t fr = __builtin_amdgcn_readfirstlane(b.ref_id[w0]), fp = __builtin_amdgcn_readfirstlane(b.pos[w0]);
            if (fr >= 0 && (uint32_t)fr < st.n_refs && fp >= 0) {
                if (fr != meta_ref) {
                    const uint64_t eo = st.ref_edits_off[fr], bo = st.ref_bases_off[fr];
                    const uint32_t l = st.ref_len[fr];
                    meta_ref = fr;
                    meta_eoff = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(eo >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)eo);
                    meta_boff = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(bo >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)bo);
                    meta_L = (uint32_t)__builtin_amdgcn_readfirstlane((int)l);
                }
                if (meta_boff != NO_DEPTH) {
                    win_ref = fr;
                    win_base = (uint32_t)fp & ~3u; // entry index = position - 1 = 0-based position: <= the first record's
                }
            }
        }
#pragma unroll 1
        for (uint32_t pass = 0; pass < ED_PASSES; pass++) {
            const uint64_t i = w0 + (uint64_t)pass * 64 + lane;
            if (i >= b.n) continue;
            // ---- the record's placement (query() filter, flags, reference slice): edits.rs:227-261
            const uint32_t f = b.flag[i];
            const int32_t ref = b.ref_id[i], pos = b.pos[i];
            if (!(ref >= 0 && (uint32_t)ref < st.n_refs && pos >= 0)) continue;
            const uint32_t n_ops = b.n_cigar[i];
            const uint64_t cbase = b.cigar_off ? b.cigar_off[i] : i * (uint64_t)b.cigar_stride;
            const uint32_t cg0 = n_ops ? b.cigar[cbase] : 0u;
            uint64_t span = 0;
            if (n_ops == 1) {
                if ((cg0 & 0xFu) <= 8u && ((0x18Du >> (cg0 & 0xFu)) & 1u)) span = cg0 >> 4;
            } else {
                for (uint32_t k = 0; k < n_ops; k++) {
                    const uint32_t cg = b.cigar[cbase + k], op = cg & 0xFu;
                    if (op <= 8u && ((0x18Du >> op) & 1u)) span += cg >> 4;
                }
            }
            const bool on_win = ref == win_ref;
            const uint64_t L = on_win ? meta_L : st.ref_len[ref];
            const uint64_t s = (uint64_t)pos + 1, e = s + span - 1;
            if (e == 0 || s > L || (f & 0x404u)) continue; // not yielded by query(); unmapped | duplicate (edits.rs:227-229)
            const uint64_t boff = on_win ? meta_boff : st.ref_bases_off[ref];
            if (boff == NO_DEPTH || e > L) { // edits.rs:245-261: no such sequence in the FASTA / slice out of range
                c[0] += 1;
                continue;
            }
            const uint64_t eoff = on_win ? meta_eoff : st.ref_edits_off[ref];
            uint32_t *const diff = st.edits + eoff;  // entry p - 1 <-> position p (the slot that holds refs after the teardown)
            uint32_t *const alts = diff + (L + 1);   // alts[p]
            const uint8_t *const sq = b.seq + (b.seq_off ? b.seq_off[i] : i * (uint64_t)b.seq_stride);
            const uint32_t l = b.l_seq[i];
            // one `M` run [p0, p0 + len) of 0-based positions joins the difference array
            auto cover = [&](uint64_t p0, uint64_t len) {
                if (!len) return;
                const uint64_t i0 = p0 - win_base, i1 = p0 + len - win_base; // (wraps when p0 < win_base: then i0 is huge)
                if (on_win && p0 >= win_base && i1 < ED_WINDOW) {
                    atomicAdd(&win[i0], 1u);
                    atomicAdd(&win[i1], 0xFFFFFFFFu);
                    top = max(top, (uint32_t)i1);
                } else {
                    atomicAdd(&diff[p0], 1u);
                    atomicAdd(&diff[p0 + len], 0xFFFFFFFFu);
                }
            };
            uint32_t edits = 0;
            uint32_t qp = 0; // record_ptr
            int err = 0;
            if (n_ops == 1 && cg0 == (l << 4) && l <= ED_FAST_MAX) {
                // ---- the usual read: one M over all its bases.  Its reference bytes start at a byte of one of the two packed
                // copies (position parity); 16 bytes of each per step, five steps in flight together.
                const uint8_t *const rb = (pos & 1 ? st.ref_bases_odd : st.ref_bases) + boff + ((uint32_t)pos >> 1);
                const uint32_t nd = (l + 7) >> 3; // dwords that hold bases
                u64 bm = 0;                       // dwords with a mismatch
                for (uint32_t c0 = 0; c0 * 4 < nd; c0 += ED_CHUNKS) {
                    uint4 sv[ED_CHUNKS], rv[ED_CHUNKS];
#pragma unroll
                    for (uint32_t k = 0; k < ED_CHUNKS; k++) {
                        sv[k] = rv[k] = make_uint4(0, 0, 0, 0);
                        if ((c0 + k) * 4 < nd) {
                            __builtin_memcpy(&sv[k], sq + 16 * (c0 + k), 16);
                            __builtin_memcpy(&rv[k], rb + 16 * (c0 + k), 16);
                        }
                    }
#pragma unroll
                    for (uint32_t k = 0; k < ED_CHUNKS; k++) {
                        const uint32_t x[4] = {sv[k].x ^ rv[k].x, sv[k].y ^ rv[k].y, sv[k].z ^ rv[k].z, sv[k].w ^ rv[k].w};
#pragma unroll
                        for (uint32_t d = 0; d < 4; d++) {
                            const uint32_t j = (c0 + k) * 4 + d;
                            uint32_t xx = x[d];
                            if (8 * j + 8 > l) xx = 8 * j < l ? xx & lead_bases_mask(l - 8 * j) : 0u; // the read's last, partial dword; nothing behind it
                            const uint32_t t = nz_nibbles(xx);
                            edits += (uint32_t)__popc(t);
                            bm |= (u64)(t != 0) << (j & 63u);
                        }
                    }
                }
                // ---- the dwords that hold a mismatch, again, for the positions (fewer than one per read on real data)
                while (bm) {
                    const uint32_t j = (uint32_t)__builtin_ctzll(bm);
                    bm &= bm - 1;
                    uint32_t xx = ld32(sq + 4 * j) ^ ld32(rb + 4 * j);
                    if (8 * j + 8 > l) xx &= lead_bases_mask(l - 8 * j);
                    uint32_t t = nz_nibbles(xx);
                    while (t) {
                        const uint32_t k = (uint32_t)__builtin_ctz(t) >> 2; // nibble k: byte k >> 1, its high nibble (k odd) is the earlier base
                        t &= t - 1;
                        atomicAdd(&alts[s + 8 * j + (k & ~1u) + ((k & 1u) ^ 1u)], 1u);
                    }
                }
                cover((uint64_t)pos, l);
                qp = l;
            } else {
                // ---- any other CIGAR: walk the operations (utils/alignment.rs:48-107); only Kind::Match compares (edits.rs:277)
                const uint8_t *const rb = st.ref_bases + boff;
                uint64_t rp = 0; // reference_ptr
                for (uint32_t k = 0; k < n_ops && !err; k++) {
                    const uint32_t cg = k ? b.cigar[cbase + k] : cg0, op = cg & 0xFu, len = cg >> 4;
                    if (op > 8u) continue;
                    const bool c_ref = (0x18Du >> op) & 1u; // M D N = X
                    const bool c_seq = (0x193u >> op) & 1u; // M I S = X
                    if (op == 0u) {
                        // (a record that runs out of bases inside an M: alignment.rs:84-87; like the reference, the positions
                        // visited before the error stay counted -- the error aborts the run anyway)
                        const uint32_t m = (uint64_t)qp + len > l ? l - qp : len;
                        const uint64_t p0 = (uint64_t)pos + rp;
                        for (uint32_t j = 0; j < m; j += 8) {
                            uint32_t xx = nibbles8(sq, qp + j) ^ nibbles8(rb, p0 + j);
                            if (m - j < 8u) xx &= 0xFFFFFFFFu << (4u * (8u - (m - j)));
                            uint32_t t = nz_nibbles(xx);
                            edits += (uint32_t)__popc(t);
                            while (t) {
                                const uint32_t q = 7u - ((uint32_t)__builtin_ctz(t) >> 2); // first base of the window is the top nibble
                                t &= t - 1;
                                atomicAdd(&alts[p0 + 1 + j + q], 1u);
                            }
                        }
                        cover(p0, m);
                        qp += m;
                        rp += m;
                        if (m < len) err = 2;
                    } else {
                        if (c_seq) {
                            if ((uint64_t)qp + len > l) {
                                err = 2;
                                break;
                            }
                            qp += len;
                        }
                        if (c_ref) rp += len;
                    }
                }
            }
            if (err == 2) c[1] += 1;
            else if (qp != l) c[2] += 1;      // alignment.rs:102-103 (the reference side is consumed by construction)
            else if (edits > 512u) c[3] += 1; // edits.rs:296-300 unwrap()
            else if (f & 0x40u) atomicAdd(&s_h1[edits], 1u);
            else atomicAdd(&s_h2[edits], 1u);
        }
        // ---- the wave flushes the touched part of its window with coalesced global atomics and leaves it zeroed.
        // No barrier: LDS operations of one wave execute in order.
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) top = max(top, (uint32_t)__shfl_xor(top, o, 64));
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (win_ref >= 0 && top) {
            uint32_t *const dst = st.edits + meta_eoff + win_base;
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
    for (uint32_t i = tid; i < NGSQ_EDITS_BINS; i += ED_THREADS) {
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
//   k_edits_chunk_scan   exclusive prefix over a sequence's chunk sums (one block; a sequence has at most 2^20 chunks)
//   k_edits_refs         chunks [c0, c1): cover[p] = sum of the entries [0, p), refs[p] = cover[p] - alts[p] in place, and the
//                        VAF histogram of the positions with refs + alts > 0 -- f32 arithmetic exactly as edits.rs:331-335:
//                        alts as f32 / total as f32, * 100.0, truncated
// A sharded run converts and tallies a slice of every sequence's chunks per rank (ngsq_exchange); the rest of a sequence is
// converted, without the tally, when somebody asks for its positions (ngsq_get_edits_positions).
// ---------------------------------------------------------------------------
constexpr uint32_t EDC = 4096; // entries per teardown chunk

__global__ __launch_bounds__(256) void k_edits_chunk_sums(const uint32_t *__restrict__ diff, uint64_t n_entries, uint32_t *__restrict__ sums) {
    __shared__ uint32_t s_w[4];
    const uint64_t base = (uint64_t)blockIdx.x * EDC;
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
    if (threadIdx.x == 0) sums[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

__global__ __launch_bounds__(1024) void k_edits_chunk_scan(uint32_t *__restrict__ sums, uint32_t n) {
    __shared__ uint32_t s_part[1024];
    const uint32_t per = (n + 1023) / 1024, lo = threadIdx.x * per, hi = min(n, lo + per);
    uint32_t t = 0;
    for (uint32_t i = lo; i < hi; i++) t += sums[i];
    s_part[threadIdx.x] = t;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (uint32_t k = 0; k < 1024; k++) {
            const uint32_t v = s_part[k];
            s_part[k] = run;
            run += v;
        }
    }
    __syncthreads();
    uint32_t run = s_part[threadIdx.x];
    for (uint32_t i = lo; i < hi; i++) {
        const uint32_t v = sums[i];
        sums[i] = run;
        run += v;
    }
}

__global__ __launch_bounds__(256) void k_edits_refs(uint32_t *__restrict__ refs, const uint32_t *__restrict__ alts, uint64_t n_entries,
                                                     const uint32_t *__restrict__ carry, uint32_t chunk0, u64 *vaf_hist) {
    __shared__ uint32_t s_h[NGSQ_VAF_BINS];
    __shared__ uint32_t s_w[4];
    if (threadIdx.x < NGSQ_VAF_BINS) s_h[threadIdx.x] = 0;
    const uint32_t chunk = chunk0 + blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t base = (uint64_t)chunk * EDC;
    uint32_t run = carry[chunk]; // sum of every entry in front of the chunk
    __syncthreads();
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
                const float vaf = __fdiv_rn((float)a[q], (float)cov); // total = refs + alts = cover
                atomicAdd(&s_h[(uint32_t)__fmul_rn(vaf, 100.0f)], 1u);
            }
            cov += d[q];
        }
        if (i + 4 <= n_entries) {
            *reinterpret_cast<uint4 *>(refs + i) = make_uint4(out[0], out[1], out[2], out[3]);
        } else {
            for (uint32_t q = 0; q < 4; q++)
                if (i + q < n_entries) refs[i + q] = out[q];
        }
        run += step_total;
        __syncthreads();
    }
    if (vaf_hist && threadIdx.x < NGSQ_VAF_BINS) {
        const uint32_t v = s_h[threadIdx.x];
        if (v) atomicAdd(&vaf_hist[threadIdx.x], (u64)v);
    }
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
hipError_t launch_edits(const LaunchInfo &li, const DeviceState &st, const DeviceBatch &b, hipStream_t s) {
    if (!b.n) return hipSuccess;
    static int per_cu = -1;
    if (per_cu < 0) {
        const char *e = getenv("NGSQ_EDITS_BLOCKS_PER_CU"); // measurement aid
        per_cu = e && atoi(e) > 0 ? atoi(e) : 8;
    }
    uint64_t g = (b.n + ED_TILE - 1) / ED_TILE;
    const uint64_t cap = (uint64_t)li.n_cu * (uint32_t)per_cu;
    if (g > cap) g = cap;
    hipLaunchKernelGGL(k_edits, dim3((uint32_t)g), dim3(ED_THREADS), 0, s, st, b);
    return hipGetLastError();
}

hipError_t launch_pack_reference(const LaunchInfo &li, const uint8_t *codes, uint64_t len, uint8_t *even, uint8_t *odd, uint64_t n_bytes,
                                 unsigned long long *bad, hipStream_t s) {
    if (!n_bytes) return hipSuccess;
    const uint32_t grid = (uint32_t)std::min<uint64_t>((n_bytes + 255) / 256, (uint64_t)li.n_cu * 16);
    hipLaunchKernelGGL(k_pack_reference, dim3(grid), dim3(256), 0, s, codes, len, even, odd, n_bytes, bad);
    return hipGetLastError();
}

uint64_t edits_teardown_chunks(uint64_t n_entries) { return (n_entries + EDC - 1) / EDC; }

hipError_t launch_edits_chunk_sums(const uint32_t *diff, uint64_t n_entries, uint32_t *sums, hipStream_t s) {
    const uint64_t n = edits_teardown_chunks(n_entries);
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(k_edits_chunk_sums, dim3((uint32_t)n), dim3(256), 0, s, diff, n_entries, sums);
    hipLaunchKernelGGL(k_edits_chunk_scan, dim3(1), dim3(1024), 0, s, sums, (uint32_t)n);
    return hipGetLastError();
}

hipError_t launch_edits_refs(uint32_t *refs, const uint32_t *alts, uint64_t n_entries, const uint32_t *carry, uint64_t chunk0, uint64_t chunk1,
                             unsigned long long *vaf_hist, hipStream_t s) {
    if (chunk1 <= chunk0) return hipSuccess;
    hipLaunchKernelGGL(k_edits_refs, dim3((uint32_t)(chunk1 - chunk0)), dim3(256), 0, s, refs, alts, n_entries, carry, (uint32_t)chunk0, vaf_hist);
    return hipGetLastError();
}

} // namespace ngsq
