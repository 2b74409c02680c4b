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

#ifndef NGSQ_GC_AHEAD
#define NGSQ_GC_AHEAD 1 // measurement builds: 0 = every load when its turn comes
#endif
// Round 6: what says WHERE a record's window lies -- its flag, its length, its row's offset, its identity -- is requested an
// iteration ahead, unconditionally and from a clamped index (a load under a branch gets a wait of its own from this compiler,
// DESIGN 5.8; for the same reason OFFS / RID -- the batch has seq_off / record_id -- are compile-time).  An iteration used to
// wait three or four times in a row: flag, then length, then offset / identity, then the window's bytes, each a round trip
// to HBM behind the test that needs it; now once, for the window.  Same box, alternating builds: 1.40-1.41 -> 1.34-1.37 ms per
// 100 M reads of 150 bases, 1.69-1.70 -> 1.62-1.64 on the 50-300 base reads of the offsets layout.  (The window's bytes an
// iteration ahead as well -- a two-stage pipeline, 66-75 registers -- measured 1.62 on the offsets layout and 2.65 on
// fixed-pitch rows: not kept.)
template <bool OFFS, bool RID>
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
    struct Info {
        uint32_t f, l;
        uint64_t row, rid;
    };
    auto fetch = [&](uint64_t j) -> Info { // (j < n)
        Info r;
        r.f = b.flag[j];
        r.l = b.l_seq[j];
        r.row = OFFS ? b.seq_off[j] : j * (uint64_t)b.seq_stride;
        r.rid = RID ? b.record_id[j] : b.first_record_index + j;
        return r;
    };
    const uint64_t step = blockDim.x, i0 = lo + threadIdx.x;
    Info nx{0, 0, 0, 0};
    if (NGSQ_GC_AHEAD && i0 < hi) nx = fetch(i0);
    for (uint64_t i = i0; i < hi; i += step) {
        Info r;
        if (NGSQ_GC_AHEAD) {
            r = nx;
            nx = fetch(i + step < hi ? i + step : i); // (behind the slice: this record's columns again)
        } else {
            r = fetch(i);
        }
        if (r.f & 0x500u) { // duplicate | secondary  gc_content.rs:41-45
            c[4] += 1;
            continue;
        }
        if (r.l < NGSQ_GC_WINDOW) { // :59-62
            c[5] += 1;
            continue;
        }
        const uint32_t off = ngsq_gc_offset_fn(st.gc_seed, r.rid, r.l); // :68-74
        const uint64_t p = r.row + (off >> 1);
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
    if (b.seq_off && b.record_id) hipLaunchKernelGGL((k_gc<true, true>), dim3(grid), dim3(256), 0, s, st, b, seq_bytes);
    else if (b.seq_off) hipLaunchKernelGGL((k_gc<true, false>), dim3(grid), dim3(256), 0, s, st, b, seq_bytes);
    else if (b.record_id) hipLaunchKernelGGL((k_gc<false, true>), dim3(grid), dim3(256), 0, s, st, b, seq_bytes);
    else hipLaunchKernelGGL((k_gc<false, false>), dim3(grid), dim3(256), 0, s, st, b, seq_bytes);
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

// ---------------------------------------------------------------------------
// Small state blocks in ONE launch (kernels.h StateSpans): ngsq_reset used to be nine hipMemsetAsync + one 16-byte
// host-to-device copy, ngsq_finalize five device-to-host copies into pageable memory -- two dozen operations of a few
// microseconds each, every one a gap on the stream (and a host round trip for every pageable copy): 0.24-0.41 ms of a
// 5.6 ms step that no kernel accounted for.  A thread takes four words of the concatenated spans.
__global__ __launch_bounds__(256) void k_state_spans(StateSpans a) {
    const uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; // quad index over the concatenation
    uint32_t k = 0;
    uint64_t base = 0;
#pragma unroll 1
    for (; k < a.n; k++) {
        const uint64_t quads = (a.span[k].n_words + 3) / 4;
        if (q < base + quads) break;
        base += quads;
    }
    if (k >= a.n) return;
    const StateSpans::Span sp = a.span[k];
    const uint64_t w0 = (q - base) * 4;
    if (sp.src) { // copy (the destination may be pinned host memory the device addresses)
        if (w0 + 4 <= sp.n_words && (((uintptr_t)sp.src | (uintptr_t)sp.dst) & 15) == 0) {
            *reinterpret_cast<uint4 *>(sp.dst + w0) = *reinterpret_cast<const uint4 *>(sp.src + w0);
        } else {
            for (uint64_t w = w0; w < sp.n_words && w < w0 + 4; w++) sp.dst[w] = sp.src[w];
        }
    } else {
        if (w0 + 4 <= sp.n_words && ((uintptr_t)sp.dst & 15) == 0) {
            *reinterpret_cast<uint4 *>(sp.dst + w0) = make_uint4(sp.value, sp.value, sp.value, sp.value);
        } else {
            for (uint64_t w = w0; w < sp.n_words && w < w0 + 4; w++) sp.dst[w] = sp.value;
        }
    }
}

hipError_t launch_state_spans(const StateSpans &a, hipStream_t s) {
    if (a.overflow) return hipErrorInvalidValue; // more blocks queued than STATE_SPANS_MAX: a block would be left un-reset / un-copied
    uint64_t quads = 0;
    for (uint32_t k = 0; k < a.n; k++) quads += (a.span[k].n_words + 3) / 4;
    if (!quads) return hipSuccess;
    hipLaunchKernelGGL(k_state_spans, dim3((uint32_t)((quads + 255) / 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}

} // namespace ngsq
