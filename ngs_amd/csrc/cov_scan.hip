// cov_scan.hip -- Coverage teardown (coverage.rs:182-246) for all sequences in one launch.
//
// The range-add kernel leaves, per covered sequence, a difference array
// (positions 0..L, sentinel L+1, zero padding to a chunk multiple) and keeps a
// running sum per 4096-position chunk (summed per 256 chunks by a tiny pre-kernel).  Every sequence's
// differences sum to zero (each +1 has its -1 inside the same array), so the
// prefix over the WHOLE block restarts at zero at each sequence by itself.
//
// A persistent WAVE owns a contiguous range of chunks: its first carry is the
// sum of the super-chunk and chunk sums in front of it, later carries chain
// from step to step.  Per step of 1024 positions: 16 consecutive positions per
// lane, wave prefix sum, depth histogram into the wave's own LDS histogram,
// integer bin totals, and the array is zeroed behind the read.
// HBM traffic: 4 B read + 4 B written per position.
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace ngsq {

typedef unsigned long long u64;

constexpr uint32_t CS_THREADS = 256;
constexpr uint32_t CS_PER_THREAD = COV_CHUNK / CS_THREADS; // 16

__device__ __forceinline__ u64 cs_wave_sum64(u64 v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// one block per super-chunk: sum of its COV_SUPER chunk sums
__global__ __launch_bounds__(COV_SUPER) void k_cov_super_sums(const uint32_t *chunk_sums, uint64_t n_chunks,
                                                             uint32_t *super_sums) {
    __shared__ uint32_t s_w[COV_SUPER / 64];
    const uint64_t i = (uint64_t)blockIdx.x * COV_SUPER + threadIdx.x;
    uint32_t v = i < n_chunks ? chunk_sums[i] : 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (uint32_t w = 0; w < COV_SUPER / 64; w++) t += s_w[w];
        super_sums[blockIdx.x] = t;
    }
}

// Every WAVE is an independent scan unit: it owns a contiguous range of chunks, its own LDS
// histogram, and chains its carry from step to step through a lane broadcast -- the main loop
// has no block barrier.  One step = 1024 positions (16 consecutive positions per lane).
constexpr uint32_t CS_STEP = 64 * CS_PER_THREAD; // positions per wave step

// SKIP: chunks flagged in a.chunk_flags were finished by the streaming pass (cov_stream.hip): they are
// not positions of this scan, hold no entries, and only their chunk sum joins the running carry.
template <bool RESET, bool SKIP>
__global__ __launch_bounds__(CS_THREADS) void k_cov_scan(CovScanArgs a) {
    extern __shared__ uint32_t s_hist[]; // one histogram of cov_cap + 2 bins per wave
    const uint32_t nb = a.cov_cap + 2;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (uint32_t i = tid; i < (CS_THREADS / 64) * nb; i += CS_THREADS) s_hist[i] = 0;
    __syncthreads();
    uint32_t *const my_hist = s_hist + wave * nb;

    const uint64_t n_units = (uint64_t)gridDim.x * (CS_THREADS / 64);
    const uint64_t unit = (uint64_t)blockIdx.x * (CS_THREADS / 64) + wave;
    const uint64_t n_range = a.c_end - a.c_begin;
    const uint64_t per = (n_range + n_units - 1) / n_units;
    const uint64_t c_lo = a.c_begin + (per * unit < n_range ? per * unit : n_range);
    const uint64_t c_hi = c_lo + per < a.c_end ? c_lo + per : a.c_end;
    if (c_lo >= c_hi) return;

    // ---- carry in front of the wave: carry_in (everything before c_begin) + the sums of
    // [c_begin, c_lo): whole super-chunks through super_sums, the ragged ends chunk by chunk
    uint32_t carry = 0;
    {
        uint32_t part = 0;
        const uint64_t s0 = (a.c_begin + COV_SUPER - 1) / COV_SUPER, s1 = c_lo / COV_SUPER; // whole supers [s0, s1)
        if (s0 < s1) {
            for (uint64_t i = a.c_begin + lane; i < s0 * COV_SUPER; i += 64) part += a.chunk_sums[i];
            for (uint64_t i = s0 + lane; i < s1; i += 64) part += a.super_sums[i];
            for (uint64_t i = s1 * COV_SUPER + lane; i < c_lo; i += 64) part += a.chunk_sums[i];
        } else {
            for (uint64_t i = a.c_begin + lane; i < c_lo; i += 64) part += a.chunk_sums[i];
        }
        if (a.carry_words && ((a.carry_mask >> lane) & 1)) part += a.carry_words[2 * lane]; // owners in front (exchange.cpp)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
        carry = a.carry_in + part;
    }

    // ---- sequence of the first chunk (wave-uniform binary search)
    uint32_t ref = 0;
    {
        uint32_t lo = 0, hi = a.n_refs; // last r with first_chunk[r] <= c_lo
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (a.ref_first_chunk[mid] <= c_lo) lo = mid; else hi = mid;
        }
        ref = lo;
    }
    bool dirty = false; // the wave's histogram holds counts of `ref`
    // per-sequence facts, reloaded only when the wave moves to another sequence (the loads sit
    // outside the step loop: in-order vmcnt would otherwise drain the prefetch every step)
    uint64_t L = 0, ref_e0 = 0, ref_e1 = 0;
    u64 *bins = a.bin_totals;
    auto load_ref = [&](uint32_t r) {
        L = a.ref_len[r];
        ref_e0 = (uint64_t)a.ref_first_chunk[r] * COV_CHUNK;
        ref_e1 = (uint64_t)a.ref_first_chunk[r + 1] * COV_CHUNK;
        bins = a.bin_totals + a.bin_off[r];
    };

    uint32_t zeros = 0; // positions of depth 0 this lane met on the current sequence
    auto flush_hist = [&](uint32_t r) { // wave-private: no barrier needed
        u64 *dst = a.hist + (u64)r * nb;
        {
            u64 z = zeros;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) z += __shfl_xor(z, o, 64);
            if (lane == 0 && z) atomicAdd(&dst[0], z);
            zeros = 0;
        }
        for (uint32_t i = lane; i < nb; i += 64) {
            const uint32_t v = my_hist[i];
            if (v) {
                atomicAdd(&dst[i], (u64)v);
                my_hist[i] = 0;
            }
        }
    };
    auto load_step = [&](uint64_t e0, uint4 (&v)[CS_PER_THREAD / 4]) { // e0: element index of the step
        const uint4 *src = reinterpret_cast<const uint4 *>(a.depth + e0 + (uint64_t)lane * CS_PER_THREAD);
#pragma unroll
        for (uint32_t r = 0; r < CS_PER_THREAD / 4; r++) v[r] = src[r];
    };
    // first chunk >= c this wave really has to read (skips sequences without an entry)
    auto next_live = [&](uint64_t c) -> uint64_t {
        while (c < c_hi) {
            while (ref + 1 < a.n_refs && c >= a.ref_first_chunk[ref + 1]) {
                if (dirty) flush_hist(ref);
                dirty = false;
                ref += 1;
            }
            if (a.seen[ref] != 0) { // coverage.rs:187-193: only sequences with an entry
                if (SKIP && a.chunk_flags[c]) {
                    carry += a.chunk_sums[c];
                    c += 1;
                    continue;
                }
                load_ref(ref);
                return c;
            }
            const uint64_t ref_end = a.ref_first_chunk[ref + 1];
            c = ref_end < c_hi ? ref_end : c_hi;
        }
        return c_hi;
    };

    uint64_t c = next_live(c_lo);
    uint64_t e = c * (uint64_t)COV_CHUNK; // element index of the current step
    uint4 cur[CS_PER_THREAD / 4];
    if (c < c_hi) load_step(e, cur);
    while (c < c_hi) {
        const uint64_t i0 = e - ref_e0 + (uint64_t)lane * CS_PER_THREAD; // position of this lane's first entry
        if (RESET) { // compile-time: a conditional store would force a full vmcnt drain per step
            uint4 *dstz = reinterpret_cast<uint4 *>(a.depth + e + (uint64_t)lane * CS_PER_THREAD);
#pragma unroll
            for (uint32_t r = 0; r < CS_PER_THREAD / 4; r++) dstz[r] = make_uint4(0, 0, 0, 0);
        }
        // ---- prefetch the next step while this one is tallied (same sequence only: a sequence
        // change flushes the histogram first)
        const uint64_t en = e + CS_STEP;
        bool same_ref = en < c_hi * (uint64_t)COV_CHUNK && en < ref_e1;
        if (SKIP && same_ref && en % COV_CHUNK == 0 && a.chunk_flags[en / COV_CHUNK]) same_ref = false;
        uint4 nxt[CS_PER_THREAD / 4];
        load_step(same_ref ? en : e, nxt); // branch-free: past the end re-read this step (zeros by now)

        uint32_t d[CS_PER_THREAD];
#pragma unroll
        for (uint32_t r = 0; r < CS_PER_THREAD / 4; r++)
            d[4 * r + 0] = cur[r].x, d[4 * r + 1] = cur[r].y, d[4 * r + 2] = cur[r].z, d[4 * r + 3] = cur[r].w;
        uint32_t tsum = 0;
#pragma unroll
        for (uint32_t t = 0; t < CS_PER_THREAD; t++) tsum += d[t];
        uint32_t inc = tsum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = __shfl_up(inc, o, 64);
            if ((int)lane >= o) inc += t;
        }
        uint32_t run = carry + inc - tsum;
        carry += __shfl(inc, 63, 64);

        // ---- depth of each position -> histogram + integer bin totals.
        // Position 0 is bin 0 on its own (its depth is always 0: starts are >= 1);
        // position i >= 1 is in bin 1 + (i-1)/bin_size  (coverage.rs:206-230).
        // The sentinel (L+1) and the padding are not positions.
        const uint32_t nvalid = i0 > L ? 0u : (L - i0 + 1 < CS_PER_THREAD ? (uint32_t)(L - i0 + 1) : CS_PER_THREAD);
        uint32_t q = 0, rem = 0; // (i - 1) divmod bin_size of the next position, kept incrementally
        if (i0 >= 1) {
            q = (uint32_t)((i0 - 1) / a.bin_size);
            rem = (uint32_t)((i0 - 1) - (uint64_t)q * a.bin_size);
        }
        u64 bin_sum = 0;
        bool split = false; // this lane crossed a bin boundary
#pragma unroll
        for (uint32_t t = 0; t < CS_PER_THREAD; t++) {
            run += d[t];
            if (t < nvalid) {
                const uint32_t depth = run;
                // depth 0 is counted in a register: in targeted data and on the untouched stretches of a shard
                // every lane would queue on bin 0 (measured: the scan of a mostly empty axis took 1.8x as long)
                zeros += depth == 0;
                if (depth) atomicAdd(&my_hist[depth <= a.cov_cap ? depth : a.cov_cap + 1], 1u);
                if (i0 + t != 0) {
                    bin_sum += depth;
                    rem += 1;
                    if (rem == a.bin_size) { // last position of bin q+1
                        if (bin_sum) atomicAdd(&bins[(u64)q + 1], bin_sum);
                        bin_sum = 0;
                        rem = 0;
                        q += 1;
                        split = true;
                    }
                }
            }
        }
        // wave-aggregate the common case: no lane crossed a boundary and all are in one bin
        const uint32_t q0 = __shfl(q, 0, 64);
        if (__all(!split && q == q0)) {
            const u64 sum = cs_wave_sum64(bin_sum);
            if (lane == 0 && sum) atomicAdd(&bins[(u64)q0 + 1], sum);
        } else if (bin_sum) {
            atomicAdd(&bins[(u64)q + 1], bin_sum);
        }
        dirty = true;
        if (same_ref) {
#pragma unroll
            for (uint32_t r = 0; r < CS_PER_THREAD / 4; r++) cur[r] = nxt[r];
            e = en;
            c = e / COV_CHUNK;
        } else {
            c = next_live((en + COV_CHUNK - 1) / COV_CHUNK);
            e = c * (uint64_t)COV_CHUNK;
            if (c < c_hi) load_step(e, cur);
        }
    }
    if (dirty) flush_hist(ref);
}

hipError_t launch_cov_scan(const LaunchInfo &li, const CovScanArgs &a, hipStream_t s) {
    if (!a.n_chunks) return hipSuccess;
    const size_t lds = (size_t)(CS_THREADS / 64) * (a.cov_cap + 2) * sizeof(uint32_t);
    static bool attr = false;
    if (!attr) {
        const void *fns[4] = {reinterpret_cast<const void *>(k_cov_scan<true, false>), reinterpret_cast<const void *>(k_cov_scan<false, false>),
                              reinterpret_cast<const void *>(k_cov_scan<true, true>), reinterpret_cast<const void *>(k_cov_scan<false, true>)};
        for (const void *f : fns) {
            hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
            if (e != hipSuccess) return e;
        }
        attr = true;
    }
    const uint32_t n_super = (uint32_t)((a.n_chunks + COV_SUPER - 1) / COV_SUPER);
    hipLaunchKernelGGL(k_cov_super_sums, dim3(n_super), dim3(COV_SUPER), 0, s, a.chunk_sums, a.n_chunks, a.super_sums);
    if (a.c_end <= a.c_begin) return hipSuccess;
    uint64_t g = (a.c_end - a.c_begin + CS_THREADS / 64 - 1) / (CS_THREADS / 64); // one chunk per wave at least
    const uint64_t cap = (uint64_t)li.n_cu * 4;
    if (g > cap) g = cap;
    if (a.reset && a.chunk_flags)
        hipLaunchKernelGGL((k_cov_scan<true, true>), dim3((uint32_t)g), dim3(CS_THREADS), lds, s, a);
    else if (a.reset)
        hipLaunchKernelGGL((k_cov_scan<true, false>), dim3((uint32_t)g), dim3(CS_THREADS), lds, s, a);
    else if (a.chunk_flags)
        hipLaunchKernelGGL((k_cov_scan<false, true>), dim3((uint32_t)g), dim3(CS_THREADS), lds, s, a);
    else
        hipLaunchKernelGGL((k_cov_scan<false, false>), dim3((uint32_t)g), dim3(CS_THREADS), lds, s, a);
    return hipGetLastError();
}

} // namespace ngsq
