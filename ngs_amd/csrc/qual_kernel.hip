// qual_kernel.hip -- Quality Score facet, fast path for fixed-pitch rows (gfx950).
// reference: src/qc/record_based/quality_scores.rs:37-49
//     for (i, val) in record.quality_scores() { scores[i + 1][val] += 1 }
//
// This is the dominant kernel of the `ngs qc` scan: 150 of the 254 algorithmic
// bytes per 150 bp record.  Two kernels (measurements: DESIGN.md section 4):
//
//  k_qual_perm  rows of up to 256 bytes.  thread = one 16-byte WINDOW (record, w) of
//               a row, w fixed per thread; ONE v_perm_b32 per byte builds the LDS
//               address of its (cycle, score) cell; see the comment above the kernel.
//  k_qual_win   rows of 257..320 bytes (and the measurement knob NGSQ_QUAL_NROT):
//               the same window-per-thread shape with a [q][kb * RP + w] table
//               and two VALU operations per byte.
//
// Shared design points:
//  * Consecutive lanes hold consecutive windows, so a wave's global_load_dwordx4
//    covers 1 KiB of contiguous bytes (rows are dense: the load is unaligned by
//    design, gfx950 serves it as one request per lane).
//  * The LDS bank of an update depends only on the cycle -- never on the score --
//    so skewed real-world score distributions cannot serialise the LDS atomics.
//  * Lanes of different records in one wave hold the same window w; taking the
//    bytes in an order rotated by the record keeps them off the same table word
//    in the same instruction.
//  * One OR-filter per 16 bytes finds windows that may hold 0xFF (absent score)
//    or an invalid score and sends only those lanes down the exact path.
//  * Next window(s) prefetched while the current one is tallied.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "kernels.h"

namespace ngsq {

typedef unsigned long long u64;

// row pitch (in words) between consecutive kb rows of one score: >= R, chosen so
// that the NROT rotated records of a 32-lane group land on disjoint bank ranges
__host__ __device__ constexpr uint32_t qw_rp(uint32_t R) {
    // 4*RP mod 32 is the bank distance between two consecutive rotation positions; take the
    // smallest RP >= R whose circular distance from 0 is at least min(R, 12) banks
    const uint32_t want = R < 12 ? R : 12;
    for (uint32_t rp = R; rp < R + 8; rp++) {
        const uint32_t s = (4 * rp) % 32, d = s < 32 - s ? s : 32 - s;
        if (d >= want) return rp;
    }
    return R;
}
__host__ __device__ constexpr uint32_t qw_cp(uint32_t R) { return (16 * qw_rp(R) + 31) / 32 * 32; }

uint32_t qual_window_lds_bytes(uint32_t R) { return QUAL_BINS * qw_cp(R) * 4; }

template <uint32_t R, uint32_t NROT>
__global__ __launch_bounds__(1024) void k_qual_win(DeviceState st, const uint8_t *__restrict__ qual, uint64_t n_rec,
                                                   uint32_t pitch) {
    NGSQ_FOREGROUND_WAVE();
    constexpr uint32_t RP = qw_rp(R), CP = qw_cp(R), CP4 = CP * 4, RP4 = RP * 4;
    constexpr uint32_t S4 = 4u * RP4; // table bytes between the first entries of two consecutive dwords of a window
    constexpr uint32_t magicR = R == 1 ? 0u : (uint32_t)(((1ull << 32) + R - 1) / R);
    extern __shared__ uint32_t s_q[]; // QUAL_BINS x CP words; the ONLY LDS object, so table offsets are LDS addresses
    constexpr uint32_t nb = QUAL_BINS * CP;
    for (uint32_t i = threadIdx.x; i < nb; i += blockDim.x) s_q[i] = 0;
    __syncthreads();

    const uint32_t rem = pitch - 16u * (R - 1); // bytes of the last window that belong to the row
    const uint64_t n_win = n_rec * R;
    const uint64_t per = (n_win + gridDim.x - 1) / gridDim.x;
    const uint64_t lo = min(per * blockIdx.x, n_win), hi = min(lo + per, n_win);
    const uint64_t rec_lo = lo / R;
    // block-local (record, window) of this thread's first window
    const uint32_t tl0 = (uint32_t)(lo - rec_lo * R) + threadIdx.x; // < R + 1024: magic division exact
    const uint32_t rl0 = R == 1 ? tl0 : __umulhi(tl0, magicR);
    uint32_t w = tl0 - rl0 * R;
    uint32_t rot = rl0 & 3u; // dword order of this record (advanced with the record index)
    constexpr uint32_t STEP_R = 1024 / R, STEP_W = 1024 % R; // blockDim.x == 1024
    // the thread's window address advances by a fixed number of bytes per pass (+ one row tail on a wrap)
    const uint32_t d0 = STEP_R * pitch + 16u * STEP_W, d1 = d0 + pitch - 16u * R;
    const uint8_t *p = qual + (rec_lo + rl0) * (uint64_t)pitch + 16u * w;
    uint32_t bad = 0;
    char *const tab = reinterpret_cast<char *>(s_q);

    auto exact = [&](const uint32_t (&x)[4], uint32_t w_) {
        // 0xFF = no score at this cycle (ngsq.h), 94..254 = decode error
        const uint32_t nvalid = (w_ == R - 1) ? rem : 16u;
        for (uint32_t j = 0; j < nvalid; j++) {
            const uint32_t q = (x[j >> 2] >> (8 * (j & 3))) & 0xFFu;
            if (q <= NGSQ_MAX_SCORE)
                atomicAdd(reinterpret_cast<uint32_t *>(tab + q * CP4 + 4u * w_ + j * RP4), 1u);
            else if (q != 0xFFu)
                bad += 1;
        }
    };
    // one 16-byte window: 2 VALU operations + one ds_add per byte on the fast path
    auto tally = [&](const uint4 &cur, uint32_t w_, uint32_t rot_) {
        const uint32_t ww[4] = {cur.x, cur.y, cur.z, cur.w};
        // any byte >= 64 (bit 6 or 7)?  Then it may be 0xFF / invalid: exact path.
        const uint32_t any = (ww[0] | ww[1] | ww[2] | ww[3]) & 0xC0C0C0C0u;
        if (__builtin_expect(any == 0u, 1)) {
            // every byte is a score < 64.  Bytes of the last window that lie beyond the row are the
            // next row's leading scores: they fall into cells of cycles >= pitch, which exist in the
            // table but are never read back.
            const uint32_t wbase = 4u * w_;
            uint32_t x[4], bd[4];
            if (NROT == 1) {
#pragma unroll
                for (uint32_t d = 0; d < 4; d++) {
                    x[d] = ww[d];
                    bd[d] = wbase + d * S4;
                }
            } else {
                // step d takes dword d ^ rot of the window (a bijection in d and in rot): lanes of the
                // up to four records of a 32-lane group never update the same table word in one instruction
                const bool r1 = rot_ & 1u, r2 = (NROT == 4) && (rot_ & 2u);
                const uint32_t y0 = r2 ? ww[2] : ww[0], y1 = r2 ? ww[3] : ww[1], y2 = r2 ? ww[0] : ww[2],
                               y3 = r2 ? ww[1] : ww[3];
                x[0] = r1 ? y1 : y0;
                x[1] = r1 ? y0 : y1;
                x[2] = r1 ? y3 : y2;
                x[3] = r1 ? y2 : y3;
                const uint32_t h = r2 ? 2u * S4 : 0u, l = r1 ? S4 : 0u;
                bd[0] = wbase + h + l;
                bd[1] = wbase + h + (S4 - l);
                bd[2] = wbase + (2u * S4 - h) + l;
                bd[3] = wbase + (2u * S4 - h) + (S4 - l);
            }
#pragma unroll
            for (uint32_t d = 0; d < 4; d++) {
#pragma unroll
                for (uint32_t k = 0; k < 4; k++) {
                    const uint32_t q = (x[d] >> (8 * k)) & 0xFFu;
                    atomicAdd(reinterpret_cast<uint32_t *>(tab + (__umul24(q, CP4) + bd[d]) + k * RP4), 1u);
                }
            }
        } else {
            exact(ww, w_);
        }
    };
    auto load = [&](const uint8_t *q) -> uint4 {
        uint4 v;
        __builtin_memcpy(&v, q, 16); // one unaligned global_load_dwordx4
        return v;
    };

    // The hot loop never touches the last window of the whole buffer (its 16-byte load would run
    // past the allocation); that single window is tallied exactly below.
    const uint64_t hi_fast = min(hi, n_win - 1);
    uint64_t t = lo + threadIdx.x;
    if (t < hi_fast) {
        uint4 cur = load(p);
        // main loop: the next window of this thread exists, so the prefetch needs no guard
        while (t + 1024 < hi_fast) {
            uint32_t w_n = w + STEP_W;
            const bool wrap = w_n >= R;
            w_n = wrap ? w_n - R : w_n;
            p += wrap ? d1 : d0;
            const uint4 nxt = load(p);
            tally(cur, w, rot);
            cur = nxt;
            w = w_n;
            rot = (rot + STEP_R + (wrap ? 1u : 0u)) & 3u;
            t += 1024;
        }
        tally(cur, w, rot);
        t += 1024;
    }
    // the last window of the buffer, byte by byte, by the thread that owns it
    if (hi == n_win && n_win > 0 && (n_win - 1 - lo) % 1024 == threadIdx.x && n_win - 1 >= lo) {
        uint32_t x[4] = {0, 0, 0, 0};
        const uint64_t off = (n_rec - 1) * (uint64_t)pitch + 16u * (R - 1);
        for (uint32_t k = 0; k < rem; k++) x[k >> 2] |= (uint32_t)qual[off + k] << (8 * (k & 3));
        exact(x, R - 1);
    }
    __syncthreads();
    // flush the cells of real cycles: cycle c = 16 w + kb < pitch
    const uint32_t n_out = pitch * QUAL_BINS;
    for (uint32_t i = threadIdx.x; i < n_out; i += blockDim.x) {
        const uint32_t c = i / QUAL_BINS, q = i - c * QUAL_BINS;
        const uint32_t v = s_q[q * CP + (c & 15u) * RP + (c >> 4)];
        if (v) atomicAdd(&st.counters[st.off_qual + i], (u64)v);
    }
    uint32_t r = bad;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) r += __shfl_down(r, o, 64);
    if ((threadIdx.x & 63) == 0 && r) atomicAdd(&st.counters[C_ERR + E_BAD_QUAL], (u64)r);
}

// ---------------------------------------------------------------------------------------------
// k_qual_ragged: the offsets layout (reads of different lengths packed back to back).
// Same window-per-lane tally as k_qual_win, but the (record, window) of a lane comes from a
// per-wave schedule: a wave takes 64 records, prefix-sums their window counts (DPP), and writes the
// record (lane) of every window into a byte map in LDS (each record lane fills its own run: at most R
// byte stores for the 64 records); for every batch of 64 consecutive windows a lane then reads its
// record from the map and fetches that record's first window and byte offset from the record's lane
// (two ds_bpermute).  Consecutive lanes therefore still read consecutive 16-byte pieces of the byte
// stream.  (The first version searched the prefix with six dependent ds_bpermute steps per window and
// gathered five values: eleven LDS-pipe operations beside the sixteen atomics of a window.)  The last window of a record (fewer than 16 bytes)
// and windows with a score >= 64 take the exact path.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t qr_wave_inclusive_sum(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false); // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false); // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false); // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false); // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false); // row_bcast:15
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false); // row_bcast:31
    return v;
}
// The table of k_qual_ragged (round 4): 16-BIT counters, two to a word -- the cells of cycle bytes kb = 2 p and 2 p + 1 of a
// window share the word (q, p, w), so that one ds_add instruction (all its lanes on the same byte k of their dword) never has
// two lanes on the two halves of a word -- for the scores 0..63 and a trash row: 65 x 8 x RP words = 41 KB for reads of up to
// 320 bases instead of 122 KB, TWO blocks of sixteen waves per CU instead of one.  A record adds at most one to a cell, so
// the table is flushed to the global counters every 63 x 1024 records of the block at the latest; scores 64..93 (rare) go to
// the global counters directly.
constexpr uint32_t QR_ROWS = 64;         // scores the LDS table holds
constexpr uint32_t QR_TRASH = QR_ROWS;   // the score the bytes behind a record's end are given: a row of the table nobody reads
constexpr uint32_t QR_FLUSH_EVERY = 63;  // iterations of 1024 records between two flushes (63 x 1024 < 65536)
__host__ __device__ constexpr uint32_t qr_cp(uint32_t R) { return (8 * qw_rp(R) + 31) / 32 * 32; } // words per score
__host__ __device__ constexpr uint32_t qr_table_bytes(uint32_t R) { return (QR_ROWS + 1) * qr_cp(R) * 4; }
__device__ __forceinline__ uint32_t qr_from_lane(uint32_t v, uint32_t src_lane) {
    return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src_lane << 2), (int)v);
}

template <uint32_t R, uint32_t NROT>
__global__ __launch_bounds__(1024) void k_qual_ragged(DeviceState st, const uint8_t *__restrict__ qual,
                                                      const uint64_t *__restrict__ qual_off, uint64_t n_rec) {
    NGSQ_FOREGROUND_WAVE();
    constexpr uint32_t RP = qw_rp(R), CP = qr_cp(R), CP4 = CP * 4, RP4 = RP * 4;
    constexpr uint32_t S4 = 2u * RP4; // a dword of a window = two pair rows
    extern __shared__ uint32_t s_q[]; // (QR_ROWS + 1) x CP words (the last row: QR_TRASH, never flushed): the only LDS object
    constexpr uint32_t nb = (QR_ROWS + 1) * CP;
    for (uint32_t i = threadIdx.x; i < nb; i += blockDim.x) s_q[i] = 0;
    char *const tab = reinterpret_cast<char *>(s_q);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint8_t *const map = reinterpret_cast<uint8_t *>(s_q + nb) + wave * (64u * R); // window -> record lane, this wave's
    // keep[n] (n = 0..16): 0xFF in the first n bytes of a 16-byte window -- the bytes of a window that belong to its record
    uint4 *const keep_tab = reinterpret_cast<uint4 *>(reinterpret_cast<uint8_t *>(s_q + nb) + 16u * 64u * R);
    if (threadIdx.x <= 16u) {
        uint32_t k[4];
        for (uint32_t d = 0; d < 4; d++) {
            const uint32_t nvd = threadIdx.x > 4u * d ? min(threadIdx.x - 4u * d, 4u) : 0u;
            k[d] = nvd >= 4u ? 0xFFFFFFFFu : (1u << (8u * nvd)) - 1u;
        }
        keep_tab[threadIdx.x] = make_uint4(k[0], k[1], k[2], k[3]);
    }
    __syncthreads();
    const uint64_t per = (n_rec + gridDim.x - 1) / gridDim.x;
    const uint64_t lo = min(per * blockIdx.x, n_rec), hi = min(lo + per, n_rec);
    const uint64_t end_bytes = n_rec ? qual_off[n_rec] : 0; // a 16-byte load must not run past this (>= 16: launch_qual_ragged)
    uint32_t bad = 0, too_long = 0;
    u64 *const gq = st.counters + st.off_qual;
    const uint32_t n_cyc = min(st.max_read_len, 16u * R);
    // the table's cells of real cycles (cycle c = 16 w + kb < max_read_len) to the global counters; the whole block
    auto flush = [&]() {
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < n_cyc * QR_ROWS; i += blockDim.x) {
            const uint32_t c = i / QR_ROWS, q = i - c * QR_ROWS, kb = c & 15u;
            const uint32_t v = (s_q[q * CP + (kb >> 1) * RP + (c >> 4)] >> (16u * (kb & 1u))) & 0xFFFFu;
            if (v) atomicAdd(&gq[(uint64_t)c * QUAL_BINS + q], (u64)v);
        }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < QR_ROWS * CP; i += blockDim.x) s_q[i] = 0;
        __syncthreads();
    };

    // offsets of a wave's 64 records, loaded one group ahead (branch-free: past the end the last entry twice)
    auto load_offs = [&](uint64_t i0, uint64_t &o0, uint64_t &o1) {
        const uint64_t rec = i0 + lane;
        o0 = qual_off[rec < n_rec ? rec : n_rec];
        o1 = qual_off[rec + 1 < n_rec ? rec + 1 : n_rec];
    };
    uint64_t nx_o0 = 0, nx_o1 = 0;
    if (lo + 64ull * wave < hi) load_offs(lo + 64ull * wave, nx_o0, nx_o1);
    const uint64_t n_iter = (hi - lo + 1023) / 1024; // the same for every wave of the block (the flush is a block's)
    for (uint64_t it = 0; it < n_iter; it++) {
        if (it && it % QR_FLUSH_EVERY == 0) flush();
        const uint64_t i0 = lo + it * 1024 + 64ull * wave;
        if (i0 >= hi) continue;
        // this wave's 64 records: offset, length (clamped to the table), windows
        const uint64_t rec = i0 + lane;
        const uint64_t cur_o0 = nx_o0, cur_o1 = nx_o1;
        load_offs(i0 + 64ull * 16, nx_o0, nx_o1);
        uint64_t off = 0;
        uint32_t len = 0;
        if (rec < hi) {
            off = cur_o0;
            const uint64_t l64 = cur_o1 - off;
            len = (uint32_t)min(l64, (uint64_t)st.max_read_len);
            too_long += l64 > st.max_read_len; // quality_scores.rs: the table is sized for max_read_len
        }
        // Every 16-byte window of the 64 records, the last, partial one of a record included, 64 consecutive windows per
        // step.  (Until round 4 the partial windows were a second pass over the 64 records: by then the lines that hold them
        // -- three lines in four of the records' bytes -- had left the L2, a block's waves having 180 KB in flight per CU:
        // TCC_EA0_RDREQ said 1.39 x the bytes.)
        const uint32_t nwin = (len + 15u) >> 4;
        const uint32_t P = qr_wave_inclusive_sum(nwin); // windows of records 0..lane
        const uint32_t T = __builtin_amdgcn_readlane(P, 63);
        // byte offsets relative to the wave's first record (64 records of at most 16 R bytes: 32 bits)
        const uint64_t off0 = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(off >> 32)) << 32) |
                              (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)off);
        const uint32_t rel = rec < hi ? (uint32_t)(off - off0) : 0u;
        const uint32_t first_win = P - nwin;
        const uint32_t fw_len = first_win | len << 16; // first window (< 64 R <= 1280) and length (<= 16 R) of the record, fetched together
        for (uint32_t j = 0; __ballot(j < nwin); j++) // nwin <= R
            if (j < nwin) map[first_win + j] = (uint8_t)lane;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // the window of the NEXT step is in flight while this one is tallied (every lane loads: an idle lane
        // re-reads the first window of the step, which is inside the buffer)
        struct Sched {
            uint32_t r, w, nv, sh; // record lane, window of the record, its bytes that count, bytes the load was moved back by
            bool active;
            uint4 v;
        };
        // Addresses as 32-bit offsets from a wave-uniform base g0 <= every address of the group.  The partial window of the
        // buffer's last record would end behind the buffer: the 16 bytes that END with the buffer are read instead
        // (offset `lim`) and shifted when used.
        const uint64_t g0 = min(off0, end_bytes - 16);
        const uint32_t delta = (uint32_t)(off0 - g0), lim = (uint32_t)min(end_bytes - 16 - g0, (uint64_t)0xFFFFFFFFu);
        const uint8_t *const gbase = qual + g0;
        auto fetch = [&](uint32_t t0) -> Sched {
            Sched sc;
            const uint32_t t = t0 + lane;
            sc.active = t < T;
            const uint32_t tt = sc.active ? t : (t0 < T ? t0 : 0u);
            sc.r = map[tt]; // record of window tt
            const uint32_t fl = qr_from_lane(fw_len, sc.r);
            sc.w = tt - (fl & 0xFFFFu);
            const uint32_t l = fl >> 16;
            sc.nv = sc.w == (l >> 4) ? (l & 15u) : 16u;
            const uint32_t at = qr_from_lane(rel, sc.r) + 16u * sc.w + delta, at_c = min(at, lim);
            sc.sh = at - at_c;
            __builtin_memcpy(&sc.v, gbase + at_c, 16);
            return sc;
        };
        Sched nx{};
        if (T) nx = fetch(0);
        for (uint32_t t0 = 0; t0 < T; t0 += 64) {
            const Sched cu = nx;
            nx = fetch(t0 + 64 < T ? t0 + 64 : t0); // past the end: a re-read
            const bool active = cu.active;
            const uint32_t r = cu.r, w = cu.w, nv = cu.nv;
            if (!active) continue;
            uint32_t ww[4] = {cu.v.x, cu.v.y, cu.v.z, cu.v.w};
            if (__builtin_expect(cu.sh != 0u, 0)) { // (one or two windows per launch)
                u64 l64 = (u64)ww[1] << 32 | ww[0], h64 = (u64)ww[3] << 32 | ww[2];
                const uint32_t sb = 8u * cu.sh;
                if (sb >= 64u) {
                    l64 = h64 >> (sb - 64u);
                    h64 = 0;
                } else {
                    l64 = l64 >> sb | h64 << (64u - sb);
                    h64 >>= sb;
                }
                ww[0] = (uint32_t)l64, ww[1] = (uint32_t)(l64 >> 32), ww[2] = (uint32_t)h64, ww[3] = (uint32_t)(h64 >> 32);
            }
            // bytes behind the record's end (they are the next record's) count in the trash row
            const uint4 kp = keep_tab[nv];
            const uint32_t keep[4] = {kp.x, kp.y, kp.z, kp.w};
            const uint32_t any = ((ww[0] & keep[0]) | (ww[1] & keep[1]) | (ww[2] & keep[2]) | (ww[3] & keep[3])) & 0xC0C0C0C0u;
            if (__builtin_expect(any == 0u, 1)) {
                const uint32_t wbase = 4u * w, rot = r & 3u;
                uint32_t x[4], bd[4];
#pragma unroll
                for (uint32_t d = 0; d < 4; d++) ww[d] = (ww[d] & keep[d]) | (~keep[d] & (QR_TRASH * 0x01010101u)); // (v_bfi_b32)
                if (NROT == 1) {
#pragma unroll
                    for (uint32_t d = 0; d < 4; d++) {
                        x[d] = ww[d];
                        bd[d] = wbase + d * S4;
                    }
                } else {
                    const bool r1 = rot & 1u, r2 = (NROT == 4) && (rot & 2u);
                    const uint32_t y0 = r2 ? ww[2] : ww[0], y1 = r2 ? ww[3] : ww[1], y2 = r2 ? ww[0] : ww[2],
                                   y3 = r2 ? ww[1] : ww[3];
                    x[0] = r1 ? y1 : y0;
                    x[1] = r1 ? y0 : y1;
                    x[2] = r1 ? y3 : y2;
                    x[3] = r1 ? y2 : y3;
                    const uint32_t h = r2 ? 2u * S4 : 0u, l = r1 ? S4 : 0u;
                    bd[0] = wbase + h + l;
                    bd[1] = wbase + h + (S4 - l);
                    bd[2] = wbase + (2u * S4 - h) + l;
                    bd[3] = wbase + (2u * S4 - h) + (S4 - l);
                }
#pragma unroll
                for (uint32_t d = 0; d < 4; d++) {
#pragma unroll
                    for (uint32_t k = 0; k < 4; k++) {
                        const uint32_t q = (x[d] >> (8 * k)) & 0xFFu;
                        atomicAdd(reinterpret_cast<uint32_t *>(tab + (__umul24(q, CP4) + bd[d]) + (k >> 1) * RP4), 1u << (16u * (k & 1u)));
                    }
                }
            } else {
                // every byte is a score in this layout: 94..255 are decode errors
                for (uint32_t j = 0; j < nv; j++) {
                    const uint32_t q = (ww[j >> 2] >> (8 * (j & 3))) & 0xFFu;
                    if (q < QR_ROWS)
                        atomicAdd(reinterpret_cast<uint32_t *>(tab + q * CP4 + 4u * w + (j >> 1) * RP4), 1u << (16u * (j & 1u)));
                    else if (q <= NGSQ_MAX_SCORE)
                        atomicAdd(&gq[(uint64_t)(16u * w + j) * QUAL_BINS + q], (u64)1);
                    else
                        bad += 1;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); // the map is rewritten for the next 64 records
        __builtin_amdgcn_wave_barrier();
    }
    flush();
    uint32_t r0 = bad, r1 = too_long;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        r0 += __shfl_down(r0, o, 64);
        r1 += __shfl_down(r1, o, 64);
    }
    if (lane == 0 && r0) atomicAdd(&st.counters[C_ERR + E_BAD_QUAL], (u64)r0);
    if (lane == 0 && r1) atomicAdd(&st.counters[C_ERR + E_READ_TOO_LONG], (u64)r1);
}

#ifndef NGSQ_QR_NROT
#define NGSQ_QR_NROT 4 // dword orders the records of a wave rotate through (1, 2 or 4: measurement builds)
#endif
template <uint32_t R>
static hipError_t launch_ragged_r(const LaunchInfo &li, const DeviceState &st, const DeviceBatch &b, hipStream_t s) {
    const uint32_t lds = qr_table_bytes(R) + 16u * 64u * R + 17u * 16u; // table + one window->record byte map per wave + the keep masks
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_qual_ragged<R, NGSQ_QR_NROT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr = true;
    }
    const uint32_t per_cu = lds <= 78 * 1024 ? 2 : 1;
    uint64_t g = (b.n + 1023) / 1024;
    if (g > (uint64_t)li.n_cu * per_cu) g = (uint64_t)li.n_cu * per_cu;
    if (g < 1) g = 1;
    hipLaunchKernelGGL((k_qual_ragged<R, NGSQ_QR_NROT>), dim3((uint32_t)g), dim3(1024), lds, s, st, b.qual, b.qual_off, b.n);
    return hipGetLastError();
}

bool qual_ragged_supported(const DeviceState &st, const DeviceBatch &b) {
    // (fewer than 16 bytes of qualities in the whole batch: the general kernel -- k_qual_ragged reads 16 bytes at a time)
    return b.qual_off != nullptr && st.max_read_len <= 16 * QUAL_WIN_MAX_R && b.qual_bytes >= 16;
}

hipError_t launch_qual_ragged(const LaunchInfo &li, const DeviceState &st, const DeviceBatch &b, hipStream_t s) {
    const uint32_t R = (st.max_read_len + 15) / 16;
    if (R <= 5) return launch_ragged_r<5>(li, st, b, s);
    if (R <= 10) return launch_ragged_r<10>(li, st, b, s);
    if (R <= 16) return launch_ragged_r<16>(li, st, b, s);
    return launch_ragged_r<20>(li, st, b, s);
}

// ---------------------------------------------------------------------------------------------
// k_qual_perm: rows of up to 256 bytes (R <= 16 windows).  One VALU operation per byte.
//
// LDS table of 64 scores x 4 dword planes x 64 words (64 KiB, two blocks per CU):
//     byte address(q, cycle c = 16 w + 4 d + k) = d * 16384 + q * 256 + w * 16 + k * 4
// so the address of byte k of dword d is  { byte1 = the score, byte0 = w*16 + k*4 }  + immediate
// d*16384: ONE v_perm_b32 assembles it from the data dword and a per-lane register holding the
// four byte-0 values.  The bank, (w*4 + k) mod 32, never depends on the score.  Scores 64..93 (never
// seen on this path's short-read rows in practice) and 0xFF cells take the exact path: LDS for
// q < 64, a global atomic otherwise.
// Lanes that hold the same window w in one wave are R lanes apart; each takes the four bytes of a
// dword in an order rotated by (lane / R) & 3, a per-lane constant folded into the v_perm selectors,
// so they never update the same table word in one instruction.
// ---------------------------------------------------------------------------------------------
constexpr uint32_t QP_MAX_R = 16;
constexpr uint32_t QP_LDS_BYTES = 64 * 1024;
typedef __attribute__((address_space(3))) uint32_t lds_u32;

// threads per block: the largest multiple of R up to 1024, so a thread keeps its window index w
__host__ __device__ constexpr uint32_t qp_threads(uint32_t R) { return R * (1024u / R); }

// exact tally of one window (rare: a 0xFF cell, a score >= 64 or an invalid byte in it)
static __device__ __noinline__ uint32_t qp_exact(u64 *__restrict__ qual_counters, uint32_t x0, uint32_t x1, uint32_t x2,
                                                 uint32_t x3, uint32_t w, uint32_t nvalid) {
    const uint32_t x[4] = {x0, x1, x2, x3};
    uint32_t bad = 0;
    for (uint32_t j = 0; j < nvalid; j++) {
        const uint32_t q = (x[j >> 2] >> (8 * (j & 3))) & 0xFFu;
        if (q < 64u) {
            lds_u32 *cell = reinterpret_cast<lds_u32 *>((j >> 2) * 16384u + q * 256u + w * 16u + (j & 3u) * 4u);
            __hip_atomic_fetch_add(cell, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else if (q <= NGSQ_MAX_SCORE) {
            atomicAdd(&qual_counters[(uint64_t)(16u * w + j) * QUAL_BINS + q], (u64)1);
        } else if (q != 0xFFu) { // 0xFF = no score at this cycle (ngsq.h), 94..254 = decode error
            bad += 1;
        }
    }
    return bad;
}

template <uint32_t R, uint32_t D>
__global__ __launch_bounds__(1024) void k_qual_perm(DeviceState st, const uint8_t *__restrict__ qual, uint64_t n_rec,
                                                    uint32_t pitch) {
    NGSQ_FOREGROUND_WAVE();
    static_assert(R >= 1 && R <= QP_MAX_R, "window index must fit the low byte of a table address");
    constexpr uint32_t T = qp_threads(R), RPB = T / R; // records per pass of the block
    constexpr uint32_t magicR = R == 1 ? 0u : (uint32_t)(((1ull << 32) + R - 1) / R);
    extern __shared__ uint32_t s_q[]; // the ONLY LDS object: table offsets are LDS addresses
    if (reinterpret_cast<uintptr_t>((lds_u32 *)s_q) != 0) __builtin_trap();
    for (uint32_t i = threadIdx.x; i < QP_LDS_BYTES / 4; i += T) s_q[i] = 0;
    __syncthreads();

    const uint32_t rem = pitch - 16u * (R - 1); // bytes of the last window that belong to the row
    const uint32_t r0 = R == 1 ? threadIdx.x : __umulhi(threadIdx.x, magicR); // < RPB
    const uint32_t w = threadIdx.x - r0 * R;
    // each block streams one contiguous run of rows, RPB rows per pass
    const uint64_t per = (n_rec + gridDim.x - 1) / gridDim.x;
    const uint64_t lo = min(per * blockIdx.x, n_rec), hi = min(lo + per, n_rec);
    const uint64_t step = (uint64_t)RPB * pitch;
    const uint8_t *pf = qual + (lo + r0) * (uint64_t)pitch + 16u * w;
    u64 *const qc = st.counters + st.off_qual;

    // Lanes with the same w in a wave belong to consecutive records: rotate the byte order by the
    // record.  Selector k = { 0, 0, data byte kk, cvec byte kk }.
    uint32_t sel[4];
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) {
        const uint32_t kk = (k + r0) & 3u;
        sel[k] = 0x0C0C0000u | ((4u + kk) << 8) | kk;
    }
    // byte j = low address byte of byte j of a dword of window w
    const uint32_t cvec = w * 0x10101010u + 0x0C080400u;

    // passes of this thread; the very last window of the buffer is never loaded as 16 bytes (the
    // load would run past the allocation): its owner stops one pass early and tallies it by bytes
    uint32_t n_it = lo + r0 < hi ? (uint32_t)((hi - lo - r0 + RPB - 1) / RPB) : 0u;
    const bool owns_last = n_it > 0 && w == R - 1 && lo + r0 + (uint64_t)(n_it - 1) * RPB == n_rec - 1;
    if (owns_last) n_it -= 1;

    uint32_t bad = 0;
    auto tally = [&](const uint4 &v) {
        // any byte >= 64 (bit 6 or 7)?  Then it may be 0xFF / a high or invalid score: exact path.
        const uint32_t any = (v.x | v.y | v.z | v.w) & 0xC0C0C0C0u;
        if (__builtin_expect(any == 0u, 1)) {
            // Bytes of the last window that lie beyond the row are the next row's leading scores:
            // they fall into cells of cycles >= pitch, which exist in the table but are never read back.
            const uint32_t x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (uint32_t d = 0; d < 4; d++) {
#pragma unroll
                for (uint32_t k = 0; k < 4; k++) {
                    // LDS address straight from the integer (the table starts at LDS offset 0)
                    const uint32_t a = __builtin_amdgcn_perm(x[d], cvec, sel[k]);
                    lds_u32 *cell = reinterpret_cast<lds_u32 *>(a) + d * 4096u;
                    __hip_atomic_fetch_add(cell, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        } else {
            bad += qp_exact(qc, v.x, v.y, v.z, v.w, w, w == R - 1 ? rem : 16u);
        }
    };
    auto load = [&]() -> uint4 {
        uint4 v;
        __builtin_memcpy(&v, pf, 16); // one unaligned global_load_dwordx4
        pf += step;
        return v;
    };

    // D windows in flight per thread.  Pass i lives in register slot i % (D+1): a load never targets
    // the slot being tallied, so the compiler needs no copies (and no vmcnt(0)) to rotate them.
    constexpr uint32_t NB = D + 1;
    uint4 buf[NB];
#pragma unroll
    for (uint32_t j = 0; j < D; j++)
        if (j < n_it) buf[j] = load();
    uint32_t done = 0;
    for (; done + NB + D <= n_it; done += NB) {
#pragma unroll
        for (uint32_t j = 0; j < NB; j++) {
            buf[(j + D) % NB] = load(); // pass done + j + D exists: no guard, nothing drains the queue
            tally(buf[j]);
        }
    }
#pragma unroll
    for (uint32_t j = 0; j < NB + D; j++) {
        if (done + j < n_it) {
            if (done + j + D < n_it) buf[(j + D) % NB] = load();
            tally(buf[j % NB]);
        }
    }

    if (owns_last) {
        uint32_t x[4] = {0, 0, 0, 0};
        const uint64_t off = (n_rec - 1) * (uint64_t)pitch + 16u * (R - 1);
        for (uint32_t k = 0; k < rem; k++) x[k >> 2] |= (uint32_t)qual[off + k] << (8 * (k & 3));
        bad += qp_exact(qc, x[0], x[1], x[2], x[3], R - 1, rem);
    }
    __syncthreads();
    // flush in table order (conflict-free reads): word i = d*4096 + q*64 + w*4 + k
    for (uint32_t i = threadIdx.x; i < QP_LDS_BYTES / 4; i += T) {
        const uint32_t v = s_q[i];
        if (v) {
            const uint32_t d = i >> 12, q = (i >> 6) & 63u, c = ((i >> 2) & 15u) * 16u + d * 4u + (i & 3u);
            if (c < pitch) atomicAdd(&qc[(uint64_t)c * QUAL_BINS + q], (u64)v);
        }
    }
    uint32_t r = bad;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) r += __shfl_down(r, o, 64);
    if ((threadIdx.x & 63) == 0 && r) atomicAdd(&st.counters[C_ERR + E_BAD_QUAL], (u64)r);
}

template <uint32_t R, uint32_t D>
static hipError_t launch_perm(const LaunchInfo &li, const DeviceState &st, const DeviceBatch &b, hipStream_t s) {
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_qual_perm<R, D>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)QP_LDS_BYTES);
        if (e != hipSuccess) return e;
        attr = true;
    }
    constexpr uint32_t RPB = qp_threads(R) / R;
    uint64_t g = (b.n + RPB - 1) / RPB;
    if (g > (uint64_t)li.n_cu * 2) g = (uint64_t)li.n_cu * 2; // 64 KiB of LDS each: two blocks per CU
    if (g < 1) g = 1;
    hipLaunchKernelGGL((k_qual_perm<R, D>), dim3((uint32_t)g), dim3(qp_threads(R)), QP_LDS_BYTES, s, st, b.qual, b.n,
                       b.qual_stride);
    return hipGetLastError();
}

template <uint32_t D>
static hipError_t dispatch_perm(uint32_t R, const LaunchInfo &li, const DeviceState &st, const DeviceBatch &b,
                                hipStream_t s) {
    switch (R) {
#define CASE(r) \
    case r: return launch_perm<r, D>(li, st, b, s);
        CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9) CASE(10) CASE(11) CASE(12) CASE(13)
        CASE(14) CASE(15) CASE(16)
#undef CASE
    default: return hipErrorInvalidValue;
    }
}

template <uint32_t R, uint32_t NROT>
static hipError_t launch_r(const LaunchInfo &li, const DeviceState &st, const DeviceBatch &b, hipStream_t s) {
    const uint32_t lds = qual_window_lds_bytes(R);
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_qual_win<R, NROT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr = true;
    }
    const uint32_t per_cu = lds <= 78 * 1024 ? 2 : 1;
    uint64_t g = (b.n * R + 1023) / 1024;
    if (g > (uint64_t)li.n_cu * per_cu) g = (uint64_t)li.n_cu * per_cu;
    if (g < 1) g = 1;
    hipLaunchKernelGGL((k_qual_win<R, NROT>), dim3((uint32_t)g), dim3(1024), lds, s, st, b.qual, b.n, b.qual_stride);
    return hipGetLastError();
}

template <uint32_t NROT>
static hipError_t dispatch(uint32_t R, const LaunchInfo &li, const DeviceState &st, const DeviceBatch &b,
                           hipStream_t s) {
    switch (R) {
#define CASE(r) \
    case r: return launch_r<r, NROT>(li, st, b, s);
        CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9) CASE(10) CASE(11) CASE(12) CASE(13)
        CASE(14) CASE(15) CASE(16) CASE(17) CASE(18) CASE(19) CASE(20)
#undef CASE
    default: return hipErrorInvalidValue;
    }
}

bool qual_window_supported(const DeviceState &st, const DeviceBatch &b) {
    return !b.qual_off && b.qual_stride >= 1 && b.qual_stride <= st.max_read_len && b.qual_stride <= 16 * QUAL_WIN_MAX_R;
}

hipError_t launch_qual_window(const LaunchInfo &li, const DeviceState &st, const DeviceBatch &b, uint32_t nrot,
                              hipStream_t s) {
    const uint32_t R = (b.qual_stride + 15) / 16;
    // two windows in flight per thread: depths 1..3, one or two blocks per CU, nontemporal loads and a
    // block-interleaved row order all measured within 2 % of each other
    if (R <= QP_MAX_R) return dispatch_perm<2>(R, li, st, b, s);
    switch (nrot) {
    case 1: return dispatch<1>(R, li, st, b, s);
    case 2: return dispatch<2>(R, li, st, b, s);
    default: return dispatch<4>(R, li, st, b, s);
    }
}

} // namespace ngsq
