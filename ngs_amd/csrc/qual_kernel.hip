// qual_kernel.hip -- Quality Score facet, fast path for fixed-pitch rows (gfx950).
// reference: src/qc/record_based/quality_scores.rs:37-49
//     for (i, val) in record.quality_scores() { scores[i + 1][val] += 1 }
//
// This is the dominant kernel of the `ngs qc` scan: 150 of the 254 algorithmic
// bytes per 150 bp record.  Design (measurements: tools/micro_qual.hip):
//
//  * thread = one 16-byte WINDOW of one row: (record, w), cycles 16w .. 16w+15.
//    Consecutive lanes hold consecutive windows, so a wave's global_load_dwordx4
//    covers 1 KiB of contiguous bytes (rows are dense: the load is unaligned by
//    design, gfx950 serves it as one request per lane).
//  * per-block LDS table [q][kb * RP + w] (q score, kb byte inside the window,
//    w window; row pitch CP = multiple of 32 words).  The bank of an update
//    depends only on (kb, w) -- never on the score -- so skewed real-world score
//    distributions cannot serialise the LDS atomics.  Address = q*CP4 + lane
//    base + immediate: two VALU operations per byte.
//  * lanes of different records in one wave hold the same window w; taking the
//    four dwords in an order rotated by the record index (NROT positions) keeps
//    them off the same table word in the same instruction.
//  * one OR-filter per 16 bytes finds windows that may hold 0xFF (absent score)
//    or an invalid score and sends only those lanes down the exact path.
//  * next window prefetched while the current one is tallied.
#include <hip/hip_runtime.h>

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
    constexpr uint32_t RP = qw_rp(R), CP = qw_cp(R), CP4 = CP * 4, RP4 = RP * 4;
    constexpr uint32_t magicR = R == 1 ? 0u : (uint32_t)(((1ull << 32) + R - 1) / R);
    extern __shared__ uint32_t s_q[]; // QUAL_BINS x CP words
    __shared__ u64 s_acc[1];
    constexpr uint32_t nb = QUAL_BINS * CP;
    for (uint32_t i = threadIdx.x; i < nb; i += blockDim.x) s_q[i] = 0;
    if (threadIdx.x == 0) s_acc[0] = 0;
    __syncthreads();

    const uint32_t rem = pitch - 16u * (R - 1); // bytes of the last window that belong to the row
    const uint64_t n_win = n_rec * R;
    const uint64_t per = (n_win + gridDim.x - 1) / gridDim.x;
    const uint64_t lo = min(per * blockIdx.x, n_win), hi = min(lo + per, n_win);
    const uint64_t rec_lo = lo / R;
    // block-local (record, window) of this thread's first window; advanced incrementally
    const uint32_t tl0 = (uint32_t)(lo - rec_lo * R) + threadIdx.x; // < R + 1024: magic division exact
    uint32_t rl = R == 1 ? tl0 : __umulhi(tl0, magicR);
    uint32_t w = tl0 - rl * R;
    constexpr uint32_t STEP_R = 1024 / R, STEP_W = 1024 % R; // blockDim.x == 1024
    uint32_t bad[1] = {0};
    char *const tab = reinterpret_cast<char *>(s_q);

    // The hot loop never touches the last window of the whole buffer (its 16-byte load
    // would run past the allocation); that single window is tallied exactly below.
    const uint64_t hi_fast = min(hi, n_win - 1);
    auto load = [&](uint32_t rl_, uint32_t w_) -> uint4 {
        uint4 v;
        __builtin_memcpy(&v, qual + (rec_lo + rl_) * (uint64_t)pitch + 16u * w_, 16); // one unaligned dwordx4
        return v;
    };
    auto exact = [&](const uint32_t (&x)[4], uint32_t w_) {
        // 0xFF = no score at this cycle (ngsq.h), 94..254 = decode error
        const uint32_t nvalid = (w_ == R - 1) ? rem : 16u;
        for (uint32_t j = 0; j < nvalid; j++) {
            const uint32_t q = (x[j >> 2] >> (8 * (j & 3))) & 0xFFu;
            if (q <= NGSQ_MAX_SCORE)
                atomicAdd(reinterpret_cast<uint32_t *>(tab + q * CP4 + 4u * w_ + j * RP4), 1u);
            else if (q != 0xFFu)
                bad[0] += 1;
        }
    };

    uint64_t t = lo + threadIdx.x;
    uint4 cur = make_uint4(0, 0, 0, 0);
    if (t < hi_fast) cur = load(rl, w);
    while (t < hi_fast) {
        // ---- prefetch the next window of this thread
        uint32_t rl_n = rl + STEP_R, w_n = w + STEP_W;
        if (w_n >= R) {
            w_n -= R;
            rl_n += 1;
        }
        const uint64_t t_n = t + 1024;
        // unconditional (branch-free) prefetch: past the end re-read the current window
        const bool more = t_n < hi_fast;
        const uint4 nxt = load(more ? rl_n : rl, more ? w_n : w);

        const uint32_t ww[4] = {cur.x, cur.y, cur.z, cur.w};
        const uint32_t wbase = 4u * w;
        // any byte >= 64 (bit 6 or 7)?  Then it may be 0xFF / invalid: exact path.
        const uint32_t any = (ww[0] | ww[1] | ww[2] | ww[3]) & 0xC0C0C0C0u;
        if (__builtin_expect(any == 0u, 1)) {
            // every byte is a score < 64.  Bytes of the last window that lie beyond the row
            // are the next row's leading scores: they fall into cells of cycles >= pitch,
            // which exist in the table but are never read back.
            uint32_t x[4], bd[4];
            if (NROT == 1) {
#pragma unroll
                for (uint32_t d = 0; d < 4; d++) {
                    x[d] = ww[d];
                    bd[d] = wbase + d * (4u * RP4);
                }
            } else if (NROT == 2) {
                const bool sw = rl & 1u;
                x[0] = sw ? ww[2] : ww[0];
                x[1] = sw ? ww[3] : ww[1];
                x[2] = sw ? ww[0] : ww[2];
                x[3] = sw ? ww[1] : ww[3];
                const uint32_t b0 = wbase + (sw ? 8u * RP4 : 0u), b2 = wbase + (sw ? 0u : 8u * RP4);
                bd[0] = b0;
                bd[1] = b0 + 4u * RP4;
                bd[2] = b2;
                bd[3] = b2 + 4u * RP4;
            } else {
                // step d takes dword (d + rot) & 3 of the window: two-level barrel rotate
                const uint32_t rot = rl & 3u;
                const bool by2 = rot & 2u, by1 = rot & 1u;
                const uint32_t y0 = by2 ? ww[2] : ww[0], y1 = by2 ? ww[3] : ww[1], y2 = by2 ? ww[0] : ww[2],
                               y3 = by2 ? ww[1] : ww[3];
                x[0] = by1 ? y1 : y0;
                x[1] = by1 ? y2 : y1;
                x[2] = by1 ? y3 : y2;
                x[3] = by1 ? y0 : y3;
                const uint32_t b_rot = __umul24(rot, 4u * RP4) + wbase, b_wrap = b_rot - 16u * RP4;
#pragma unroll
                for (uint32_t d = 0; d < 4; d++) bd[d] = (rot + d >= 4u ? b_wrap : b_rot) + d * (4u * RP4);
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
            exact(ww, w);
        }
        cur = nxt;
        rl = rl_n;
        w = w_n;
        t = t_n;
    }
    // the last window of the buffer, byte by byte, by the thread that owns it
    if (hi == n_win && n_win > 0 && t == n_win - 1) {
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
    const uint32_t lane = threadIdx.x & 63;
    uint32_t r = bad[0];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) r += __shfl_down(r, o, 64);
    if (lane == 0 && r) atomicAdd(&s_acc[0], (u64)r);
    __syncthreads();
    if (threadIdx.x == 0 && s_acc[0]) atomicAdd(&st.counters[C_ERR + E_BAD_QUAL], s_acc[0]);
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
    switch (nrot) {
    case 1: return dispatch<1>(R, li, st, b, s);
    case 2: return dispatch<2>(R, li, st, b, s);
    default: return dispatch<4>(R, li, st, b, s);
    }
}

} // namespace ngsq
