// kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the `ngs qc` record scan.
//
// Every kernel is an HBM-bound integer scan over SoA columns: coalesced column
// loads, per-thread register tallies, per-block LDS histograms, one flush of
// non-zero bins per block into the packed uint64 counter block with
// device-scope atomics.  No MFMA anywhere.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "../../include/ngsq_shared.h"
#include "kernels.h"

namespace ngsq {

typedef unsigned long long u64;

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ u64 wave_sum64(u64 v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// Sum N per-thread tallies over the block and add them to global counters
// dst[idx[k]].  s_acc: N u64 in LDS, zeroed before the call and barrier'd.
template <int N>
__device__ __forceinline__ void block_flush(const uint32_t (&v)[N], u64 *s_acc, u64 *dst,
                                            const uint32_t (&idx)[N]) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < N; k++) {
        uint32_t r = wave_sum(v[k]);
        if (lane == 0 && r) atomicAdd(&s_acc[k], (u64)r);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < N; k += blockDim.x) {
        u64 r = s_acc[k];
        if (r) atomicAdd(&dst[idx[k]], r);
    }
}

// contiguous slice of [0, n) owned by this block (keeps coordinate-sorted
// records of one block in one coordinate window)
__device__ __forceinline__ void block_slice(uint64_t n, uint64_t &lo, uint64_t &hi) {
    const uint64_t per = (n + gridDim.x - 1) / gridDim.x;
    lo = per * blockIdx.x;
    hi = lo + per;
    if (lo > n) lo = n;
    if (hi > n) hi = n;
}

// ---------------------------------------------------------------------------
// GC Content: one thread per record, gathering only the 100-base window.
// reference: gc_content.rs:38-100
//
// The window [off, off+100) of a record occupies 50 or 51 bytes of its packed
// row starting at byte off/2.  Each lane loads exactly those bytes (3 x
// unaligned dwordx4 + 1 x dword from its own row; neighbouring lanes' rows share
// cache lines, so every fetched line is used) and the window then sits at a
// FIXED place in 13 registers: only the parity of `off` (one leading nibble)
// varies.  Classification is nibble-parallel (8 bases per bit-op) with
// compile-time masks, ~150 VALU operations per record, no LDS staging and no
// dependence on the row layout: the same kernel serves fixed-pitch rows and the
// offsets layout.
// BAM base codes: A=1 C=2 G=4 T=8, anything else is "other" (gc_content.rs:79-86).
// ---------------------------------------------------------------------------
__device__ __forceinline__ void gc_dword(uint32_t x, uint32_t mask, uint32_t &gc, uint32_t &at) {
    const uint32_t y = x >> 1, z = x >> 2, u = x >> 3;
    gc += __popc((y ^ z) & ~(x | u) & mask); // nibble == 0010 or 0100
    at += __popc((x ^ u) & ~(y | z) & mask); // nibble == 0001 or 1000
}

__global__ __launch_bounds__(256) void k_gc(DeviceState st, DeviceBatch b, uint64_t seq_bytes) {
    NGSQ_FOREGROUND_WAVE();
    __shared__ uint32_t s_hist[NGSQ_GC_BINS];
    __shared__ u64 s_acc[6];
    if (threadIdx.x < NGSQ_GC_BINS) s_hist[threadIdx.x] = 0;
    if (threadIdx.x < 6) s_acc[threadIdx.x] = 0;
    __syncthreads();
    uint32_t c[6] = {0, 0, 0, 0, 0, 0}; // gc, at, other, processed, ign_flags, ign_short
    constexpr uint32_t M = 0x11111111u;

    uint64_t lo, hi;
    block_slice(b.n, lo, hi);
    for (uint64_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        const uint32_t f = b.flag[i];
        if (f & 0x500u) { // duplicate | secondary  gc_content.rs:41-45
            c[4] += 1;
            continue;
        }
        const uint32_t l = b.l_seq[i];
        if (l < NGSQ_GC_WINDOW) { // :59-62
            c[5] += 1;
            continue;
        }
        const uint32_t off = ngsq_gc_offset_fn(st.gc_seed, b.record_id ? b.record_id[i] : b.first_record_index + i, l); // :68-74
        const uint64_t row = b.seq_off ? b.seq_off[i] : i * (uint64_t)b.seq_stride;
        const uint64_t p = row + (off >> 1);
        const uint32_t odd = off & 1u;
        uint32_t x[13];
        if (p + 52 <= seq_bytes) {
            uint4 v0, v1, v2;
            __builtin_memcpy(&v0, b.seq + p, 16);
            __builtin_memcpy(&v1, b.seq + p + 16, 16);
            __builtin_memcpy(&v2, b.seq + p + 32, 16);
            __builtin_memcpy(&x[12], b.seq + p + 48, 4);
            x[0] = v0.x, x[1] = v0.y, x[2] = v0.z, x[3] = v0.w;
            x[4] = v1.x, x[5] = v1.y, x[6] = v1.z, x[7] = v1.w;
            x[8] = v2.x, x[9] = v2.y, x[10] = v2.z, x[11] = v2.w;
        } else { // the last rows of the buffer: never read past its end
#pragma unroll
            for (uint32_t d = 0; d < 13; d++) {
                uint32_t t = 0;
                for (uint32_t k = 0; k < 4; k++)
                    if (p + 4 * d + k < seq_bytes) t |= (uint32_t)b.seq[p + 4 * d + k] << (8 * k);
                x[d] = t;
            }
        }
        uint32_t gc = 0, at = 0;
        // byte j of the gathered run holds bases (high nibble, low nibble).  off odd: the run
        // starts one nibble early (drop the high nibble of byte 0) and ends in the high nibble of byte 50.
        gc_dword(x[0], odd ? (M & ~0x10u) : M, gc, at);
#pragma unroll
        for (uint32_t d = 1; d < 12; d++) gc_dword(x[d], M, gc, at);
        gc_dword(x[12], odd ? 0x00101111u : 0x00001111u, gc, at);
        c[0] += gc;
        c[1] += at;
        c[2] += NGSQ_GC_WINDOW - gc - at;
        c[3] += 1;
        atomicAdd(&s_hist[gc], 1u); // :91-96 round(gc/100*100) == gc
    }
    __syncthreads();
    if (threadIdx.x < NGSQ_GC_BINS) {
        uint32_t v = s_hist[threadIdx.x];
        if (v) atomicAdd(&st.counters[OFF_GC_HIST + threadIdx.x], (u64)v);
    }
    const uint32_t idx[6] = {C_GC_GC, C_GC_AT, C_GC_OTHER, C_GC_PROCESSED, C_GC_IGN_FLAGS, C_GC_IGN_SHORT};
    block_flush<6>(c, s_acc, st.counters, idx);
}

// ---------------------------------------------------------------------------
// Quality Score, general path: one wave per record, lane = cycle (mod 64)
// reference: quality_scores.rs:37-49
// LDS table [rows][94] u32; cycles >= rows go straight to global atomics.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_qual_general(DeviceState st, DeviceBatch b, uint32_t lds_rows) {
    NGSQ_FOREGROUND_WAVE();
    extern __shared__ uint32_t s_q[]; // lds_rows * 94
    __shared__ u64 s_acc[2];
    const uint32_t nbins = lds_rows * QUAL_BINS;
    for (uint32_t i = threadIdx.x; i < nbins; i += blockDim.x) s_q[i] = 0;
    if (threadIdx.x < 2) s_acc[threadIdx.x] = 0;
    __syncthreads();

    uint32_t c[2] = {0, 0}; // bad quality, read too long
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
    uint64_t lo, hi;
    block_slice(b.n, lo, hi);
    for (uint64_t i = lo + wave; i < hi; i += waves) {
        uint64_t q0;
        uint32_t nq;
        if (b.qual_off) {
            q0 = b.qual_off[i];
            nq = (uint32_t)(b.qual_off[i + 1] - q0);
        } else {
            q0 = i * (uint64_t)b.qual_stride;
            nq = b.qual_stride;
        }
        const uint8_t *q = b.qual + q0;
        const bool fixed_row = b.qual_off == nullptr;
        if (nq > st.max_read_len) {
            // reads longer than the table: an error unless the excess is row padding
            uint32_t over = 0;
            for (uint32_t cyc = st.max_read_len + lane; cyc < nq; cyc += 64)
                over |= (!fixed_row || q[cyc] != 0xFFu);
            if (__any(over) && lane == 0) c[1] += 1;
            nq = st.max_read_len;
        }
        for (uint32_t cyc = lane; cyc < nq; cyc += 64) {
            const uint32_t v = q[cyc];
            if (fixed_row && v == 0xFFu) continue; // no score at this cycle (ngsq.h)
            if (v > NGSQ_MAX_SCORE) {
                c[0] += 1;
            } else if (cyc < lds_rows) {
                atomicAdd(&s_q[cyc * QUAL_BINS + v], 1u);
            } else {
                atomicAdd(&st.counters[st.off_qual + (uint64_t)cyc * QUAL_BINS + v], 1ull);
            }
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < nbins; i += blockDim.x) {
        uint32_t v = s_q[i];
        if (v) atomicAdd(&st.counters[st.off_qual + i], (u64)v);
    }
    const uint32_t idx[2] = {C_ERR + E_BAD_QUAL, C_ERR + E_READ_TOO_LONG};
    block_flush<2>(c, s_acc, st.counters, idx);
}

// ---------------------------------------------------------------------------
// Edits process, one thread per record
// reference: edits.rs:217-303, utils/alignment.rs:48-107, utils/cigar.rs
//
// refs/alts per position are what the reference keeps per sequence (edits.rs:59-63).  One global atomic per
// compared base (150 per read, ~60 of them on every position at whole-genome depth) ran at the L2's atomic rate:
// 18 ms per 10 M reads.  A block works through consecutive tiles of 256 records; in a coordinate-sorted file they
// cover a few hundred positions, so the block tallies into an LDS window of EW positions anchored at the tile's
// first placed record and adds the window to the global arrays once per tile (positions outside the window, other
// sequences, unsorted input: straight to the global arrays -- always correct).
//
// Round 3: one LDS atomic per compared BASE was what bounded the kernel (150 per read).  The window's counters are 8 bits
// wide now, four positions to a dword -- a tile is 255 records, and a record adds at most one to a position, so a byte
// cannot overflow between two flushes -- and the usual read adds eight positions at a time: the eight match / mismatch
// outcomes of a step are a 64-bit word of 0/1 bytes (sequence nibbles spread to bytes with two v_perm, compared with the
// eight reference bytes by a carry-free "byte is non-zero"), shifted to the window's alignment and added with two or three
// atomics where there were eight.
// ---------------------------------------------------------------------------
#ifndef EDITS_EXP
#define EDITS_EXP 0 // measurement builds only: 1 = the flush adds nothing to the global arrays, 2 = no LDS atomics in the fast path
#endif
constexpr uint32_t EW = 4096;          // positions per LDS window: tile of 255 sorted reads (~650 positions at 60x) + the longest read / skip
constexpr uint32_t E_TILE = 255;       // records per tile: what an 8-bit counter holds
// inc: eight 0/1 bytes for positions o .. o+7 of the packed window w (4 positions per dword)
__device__ __forceinline__ void edits_add8(uint32_t *w, uint32_t o, uint64_t inc) {
    const uint32_t d = o >> 2, sh = (o & 3u) * 8u, i0 = (uint32_t)inc, i1 = (uint32_t)(inc >> 32);
    const uint32_t lo = i0 << sh;
    const uint32_t mid = sh ? __builtin_amdgcn_alignbit(i1, i0, 32u - sh) : i1;
    const uint32_t hi = sh ? i1 >> (32u - sh) : 0u;
    if (lo) atomicAdd(&w[d], lo);
    if (mid) atomicAdd(&w[d + 1], mid);
    if (hi) atomicAdd(&w[d + 2], hi);
}
__global__ __launch_bounds__(256) void k_edits(DeviceState st, DeviceBatch b) {
    NGSQ_FOREGROUND_WAVE();
    __shared__ uint32_t s_h1[NGSQ_EDITS_BINS], s_h2[NGSQ_EDITS_BINS];
    __shared__ uint32_t w_refs[EW / 4 + 4], w_alts[EW / 4 + 4]; // 8-bit counters, position o in byte o & 3 of dword o >> 2
    __shared__ u64 s_acc[4];
    __shared__ u64 s_key;      // (sequence << 32 | first position) of the tile's window
    __shared__ uint32_t s_whi; // one past the last window offset the tile touched
    for (uint32_t i = threadIdx.x; i < NGSQ_EDITS_BINS; i += blockDim.x) s_h1[i] = s_h2[i] = 0;
    for (uint32_t i = threadIdx.x; i < EW / 4 + 4; i += blockDim.x) w_refs[i] = w_alts[i] = 0;
    if (threadIdx.x < 4) s_acc[threadIdx.x] = 0;
    uint32_t c[4] = {0, 0, 0, 0}; // bad_ref, record_short, not_consumed, too_many

    uint64_t lo, hi;
    block_slice(b.n, lo, hi);
    for (uint64_t t0 = lo; t0 < hi; t0 += E_TILE) {
        const uint64_t i = t0 + threadIdx.x;
        const bool live = threadIdx.x < E_TILE && i < hi;
        if (threadIdx.x == 0) {
            s_key = ~0ull;
            s_whi = 0;
        }
        __syncthreads();
        // ---- the record's placement (query() filter, flags, reference slice): edits.rs:227-261
        uint32_t f = 0, n_ops = 0;
        int32_t ref = -1;
        uint64_t cbase = 0, L = 0, s = 0, span = 0;
        bool go = false;
        if (live) {
            f = b.flag[i];
            ref = b.ref_id[i];
            const int32_t pos = b.pos[i];
            if (ref >= 0 && (uint32_t)ref < st.n_refs && pos >= 0) {
                n_ops = b.n_cigar[i];
                cbase = b.cigar_off ? b.cigar_off[i] : i * (uint64_t)b.cigar_stride;
                for (uint32_t k = 0; k < n_ops; k++) {
                    const uint32_t cg = b.cigar[cbase + k], op = cg & 0xFu;
                    if (op <= 8u && ((0x18Du >> op) & 1u)) span += cg >> 4;
                }
                L = st.ref_len[ref];
                s = (uint64_t)pos + 1;
                const uint64_t e = s + span - 1;
                go = !(e == 0 || s > L)     // not yielded by query()
                     && !(f & 0x404u);      // unmapped | duplicate  edits.rs:227-229
                if (go && (st.ref_bases_off[ref] == NO_DEPTH || s + span - 1 > L)) { // edits.rs:245-261
                    c[0] += 1;
                    go = false;
                }
            }
        }
        if (go) atomicMin(&s_key, (u64)(uint32_t)ref << 32 | s);
        __syncthreads();
        const u64 key = s_key;
        const int32_t wref = key == ~0ull ? -1 : (int32_t)(key >> 32);
        const uint64_t wbase = key & 0xFFFFFFFFull;
        if (go) {
            const uint8_t *rb = st.ref_bases + st.ref_bases_off[ref] + (s - 1);
            uint32_t *refs = st.edits + st.ref_edits_off[ref];
            uint32_t *alts = refs + (L + 1);
            const bool in_seq = ref == wref;
            const uint8_t *sq = b.seq + (b.seq_off ? b.seq_off[i] : i * (uint64_t)b.seq_stride);
            const uint32_t l = b.l_seq[i];
            uint64_t rp = 0;  // reference_ptr
            uint32_t qp = 0;  // record_ptr
            uint32_t edits = 0, top = 0;
            int err = 0;
            uint32_t k0 = 0;
            const int64_t base_off = (int64_t)s - (int64_t)wbase;
            if (n_ops == 1 && b.cigar[cbase] == (l << 4) && in_seq && base_off >= 0 && base_off + (int64_t)l + 8 <= (int64_t)EW) {
                // The usual read: one M over all its bases, all of it inside the tile's window.  Eight bases per step: a dword
                // of packed sequence against eight reference bytes -> two dwords of 0/1 bytes (mismatch) and their complement
                // (match), shifted to the window's alignment -- the same for every step of a read, steps being eight positions
                // apart -- and added with two LDS atomics per array; the bytes a step pushes into a third dword are carried into
                // the next step's first.  64 bases per round trip to memory: the sixteen loads of a round are issued together
                // (one dependent global load per eight bases made the kernel wait out nineteen memory latencies per read).
                // (rows and the reference slices may be read a few bytes past their end: both buffers carry slack)
                k0 = 1;
                const uint32_t o = (uint32_t)base_off, sh = (o & 3u) * 8u;
                uint32_t *pr = &w_refs[o >> 2], *pa = &w_alts[o >> 2];
                uint32_t carry_r = 0, carry_a = 0;
                // (reads of one length -- the lanes of a wave that are here agree on l -- run the loops on a scalar bound: no
                // execution-mask bookkeeping around the loads and the steps)
                const uint32_t lu = __builtin_amdgcn_readfirstlane(l);
                const bool same_l = __ballot(l != lu) == 0;
                auto rounds = [&](const uint32_t ll) __attribute__((always_inline)) {
                for (uint32_t j1 = 0; j1 < ll; j1 += 64) {
                    uint32_t sw8[8];
                    uint64_t rw8[8];
#pragma unroll
                    for (uint32_t k = 0; k < 8; k++) {
                        const uint32_t jb = j1 + 8 * k;
                        sw8[k] = 0;
                        rw8[k] = 0;
                        if (jb < ll) {
                            __builtin_memcpy(&sw8[k], sq + (jb >> 1), 4);
                            __builtin_memcpy(&rw8[k], rb + jb, 8);
                        }
                    }
#pragma unroll
                    for (uint32_t k = 0; k < 8; k++) {
                        const uint32_t j0 = j1 + 8 * k;
                        if (j0 >= ll) break;
                        const uint32_t sw = sw8[k];
                        // the eight bases as bytes, in order: even positions are the high nibbles, odd ones the low nibbles
                        const uint32_t ev = (sw >> 4) & 0x0F0F0F0Fu, od = sw & 0x0F0F0F0Fu;
                        const uint32_t x0 = __builtin_amdgcn_perm(od, ev, 0x05010400u) ^ (uint32_t)rw8[k];
                        const uint32_t x1 = __builtin_amdgcn_perm(od, ev, 0x07030602u) ^ (uint32_t)(rw8[k] >> 32);
                        // a byte of x is non-zero where read and reference differ (any reference byte, also > 15): 1 there, 0 elsewhere
                        uint32_t a0 = ((((x0 & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x0) >> 7) & 0x01010101u;
                        uint32_t a1 = ((((x1 & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x1) >> 7) & 0x01010101u;
                        uint32_t r0 = a0 ^ 0x01010101u, r1 = a1 ^ 0x01010101u;
                        if (ll - j0 < 8u) { // the read's last, partial step
                            const uint64_t vm = (1ull << (8u * (ll - j0))) - 1ull;
                            a0 &= (uint32_t)vm, r0 &= (uint32_t)vm;
                            a1 &= (uint32_t)(vm >> 32), r1 &= (uint32_t)(vm >> 32);
                        }
                        edits += (uint32_t)__popc(a0) + (uint32_t)__popc(a1);
                        const uint32_t step = j0 >> 3;
                        if (EDITS_EXP == 2) {
                            carry_r ^= r0 + r1 + a0 + a1; // (keeps the values alive)
                            continue;
                        }
                        atomicAdd(pr + 2 * step, (r0 << sh) | carry_r);
                        atomicAdd(pr + 2 * step + 1, sh ? __builtin_amdgcn_alignbit(r1, r0, 32u - sh) : r1);
                        carry_r = sh ? r1 >> (32u - sh) : 0u;
                        atomicAdd(pa + 2 * step, (a0 << sh) | carry_a);
                        atomicAdd(pa + 2 * step + 1, sh ? __builtin_amdgcn_alignbit(a1, a0, 32u - sh) : a1);
                        carry_a = sh ? a1 >> (32u - sh) : 0u;
                    }
                }
                };
                if (same_l) rounds(lu);
                else rounds(l);
                const uint32_t steps = (l + 7) >> 3;
                if (carry_r) atomicAdd(pr + 2 * steps, carry_r);
                if (carry_a) atomicAdd(pa + 2 * steps, carry_a);
                top = o + l;
                qp = l;
            }
            for (uint32_t k = k0; k < n_ops && !err; k++) {
                const uint32_t cg = b.cigar[cbase + k], op = cg & 0xFu, len = cg >> 4;
                if (op > 8u) continue;
                const bool c_ref = (0x18Du >> op) & 1u;  // M D N = X
                const bool c_seq = (0x193u >> op) & 1u;  // M I S = X
                if (op == 0u) { // only Kind::Match compares (edits.rs:277)
                    for (uint32_t j = 0; j < len; j++) {
                        if (qp >= l) { // alignment.rs:84-87
                            err = 2;
                            break;
                        }
                        const uint32_t byte = sq[qp >> 1];
                        const uint32_t rec = (qp & 1u) ? (byte & 0xFu) : (byte >> 4);
                        const uint32_t rbase = rb[rp];
                        const uint64_t off = s + rp - wbase;
                        if (in_seq && off < EW) {
                            atomicAdd(rbase != rec ? &w_alts[off >> 2] : &w_refs[off >> 2], 1u << (8u * ((uint32_t)off & 3u)));
                            top = (uint32_t)off + 1;
                        } else {
                            atomicAdd(rbase != rec ? &alts[s + rp] : &refs[s + rp], 1u);
                        }
                        edits += rbase != rec;
                        rp += 1;
                        qp += 1;
                    }
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
            if (top) atomicMax(&s_whi, top);
            // NOTE: like the reference, positions visited before an error stay counted;
            // the error aborts the run anyway.
            if (err == 2) c[1] += 1;
            else if (qp != l) c[2] += 1;     // alignment.rs:102-103 (reference side is consumed by construction)
            else if (edits > 512u) c[3] += 1; // edits.rs:296-300 unwrap()
            else if (f & 0x40u) atomicAdd(&s_h1[edits], 1u);
            else atomicAdd(&s_h2[edits], 1u);
        }
        __syncthreads();
        // ---- the window goes to the global arrays (and is zero again for the next tile)
        const uint32_t whi = s_whi;
        if (whi) {
            uint32_t *refs = st.edits + st.ref_edits_off[wref];
            uint32_t *alts = refs + ((uint64_t)st.ref_len[wref] + 1);
            for (uint32_t d = threadIdx.x; d < (whi + 3) / 4; d += blockDim.x) {
                const uint32_t r = w_refs[d], a = w_alts[d];
                if (r) {
#pragma unroll
                    for (uint32_t k = 0; k < 4; k++)
                        if (((r >> (8 * k)) & 0xFFu) && EDITS_EXP != 1) atomicAdd(&refs[wbase + 4 * d + k], (r >> (8 * k)) & 0xFFu);
                    w_refs[d] = 0;
                }
                if (a) {
#pragma unroll
                    for (uint32_t k = 0; k < 4; k++)
                        if (((a >> (8 * k)) & 0xFFu) && EDITS_EXP != 1) atomicAdd(&alts[wbase + 4 * d + k], (a >> (8 * k)) & 0xFFu);
                    w_alts[d] = 0;
                }
            }
        }
        __syncthreads(); // the next tile resets the window's anchor and tallies into the same cells
    }
    for (uint32_t i = threadIdx.x; i < NGSQ_EDITS_BINS; i += blockDim.x) {
        uint32_t v = s_h1[i];
        if (v) atomicAdd(&st.counters[st.off_edits1 + i], (u64)v);
        v = s_h2[i];
        if (v) atomicAdd(&st.counters[st.off_edits2 + i], (u64)v);
    }
    const uint32_t idx[4] = {C_ERR + E_EDITS_BAD_REF, C_ERR + E_EDITS_SHORT, C_ERR + E_EDITS_NOT_CONSUMED,
                             C_ERR + E_EDITS_TOO_MANY};
    block_flush<4>(c, s_acc, st.counters, idx);
}

// ---------------------------------------------------------------------------
// Edits teardown (edits.rs:320-341): one VAF histogram increment per covered position.
// f32 arithmetic exactly as the reference: alts as f32 / total as f32, * 100.0, truncate.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_edits_vaf(const uint32_t *refs, const uint32_t *alts, uint32_t ref_len,
                                                   u64 *vaf_hist) {
    __shared__ uint32_t s_h[NGSQ_VAF_BINS];
    if (threadIdx.x < NGSQ_VAF_BINS) s_h[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t n = (uint64_t)ref_len + 1;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint32_t r = refs[i], a = alts[i];
        const uint64_t total = (uint64_t)r + a;
        if (total == 0) continue;
        const float vaf = __fdiv_rn((float)a, (float)total);
        const float scaled = __fmul_rn(vaf, 100.0f);
        atomicAdd(&s_h[(uint32_t)scaled], 1u);
    }
    __syncthreads();
    if (threadIdx.x < NGSQ_VAF_BINS) {
        uint32_t v = s_h[threadIdx.x];
        if (v) atomicAdd(&vaf_hist[threadIdx.x], (u64)v);
    }
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
static inline uint32_t grid_for(uint64_t n, uint32_t per_block, uint32_t max_blocks) {
    uint64_t g = (n + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > max_blocks) g = max_blocks;
    return (uint32_t)g;
}

hipError_t launch_gc(const LaunchInfo &li, const DeviceState &st, const DeviceBatch &b, uint64_t seq_bytes,
                     hipStream_t s) {
    if (!b.n) return hipSuccess;
    const uint32_t grid = grid_for(b.n, 256 * 4, li.n_cu * 8);
    hipLaunchKernelGGL(k_gc, dim3(grid), dim3(256), 0, s, st, b, seq_bytes);
    return hipGetLastError();
}

hipError_t launch_qual(const LaunchInfo &li, const DeviceState &st, const DeviceBatch &b, hipStream_t s) {
    if (!b.n) return hipSuccess;
    // fast path: fixed-pitch rows no longer than the table (qual_kernel.hip)
    if (qual_window_supported(st, b)) {
        static int nrot = -1;
        if (nrot < 0) {
            const char *e = getenv("NGSQ_QUAL_NROT"); // measurement knob; default 4
            nrot = e ? atoi(e) : 4;
        }
        return launch_qual_window(li, st, b, (uint32_t)nrot, s);
    }
    if (qual_ragged_supported(st, b)) return launch_qual_ragged(li, st, b, s);
    uint32_t rows = b.qual_off ? st.max_read_len : b.qual_stride;
    if (rows > st.max_read_len) rows = st.max_read_len;
    if (rows > QUAL_LDS_MAX_ROWS) rows = QUAL_LDS_MAX_ROWS;
    if (rows < 1) rows = 1;
    const size_t lds = (size_t)rows * QUAL_BINS * sizeof(uint32_t);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_qual_general),
                                           hipFuncAttributeMaxDynamicSharedMemorySize,
                                           QUAL_LDS_MAX_ROWS * QUAL_BINS * sizeof(uint32_t));
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const uint32_t per_cu = lds <= 72 * 1024 ? 2 : 1;
    const uint32_t grid = grid_for(b.n, 1024, li.n_cu * per_cu);
    hipLaunchKernelGGL(k_qual_general, dim3(grid), dim3(1024), lds, s, st, b, rows);
    return hipGetLastError();
}

hipError_t launch_edits(const LaunchInfo &li, const DeviceState &st, const DeviceBatch &b, hipStream_t s) {
    if (!b.n) return hipSuccess;
    // 73 VGPRs: six blocks per CU are resident.  Eight per CU (round 2) ran as one full round and a second one at a third of
    // the device; twelve per CU = two full rounds.  (NGSQ_EDITS_BLOCKS_PER_CU: measurement aid)
    static int per_cu = -1;
    if (per_cu < 0) {
        const char *e = getenv("NGSQ_EDITS_BLOCKS_PER_CU");
        per_cu = e && atoi(e) > 0 ? atoi(e) : 12;
    }
    const uint32_t grid = grid_for(b.n, 256, li.n_cu * (uint32_t)per_cu);
    hipLaunchKernelGGL(k_edits, dim3(grid), dim3(256), 0, s, st, b);
    return hipGetLastError();
}

hipError_t launch_edits_vaf(const LaunchInfo &li, const uint32_t *refs, const uint32_t *alts,
                            uint32_t ref_len, unsigned long long *vaf_hist, hipStream_t s) {
    const uint32_t grid = grid_for((uint64_t)ref_len + 1, 256 * 8, li.n_cu * 8);
    hipLaunchKernelGGL(k_edits_vaf, dim3(grid), dim3(256), 0, s, refs, alts, ref_len, vaf_hist);
    return hipGetLastError();
}

} // namespace ngsq
