// exchange_kernels.hip -- the three small device operations of the shard exchange (exchange.cpp):
// add a received halo into the difference arrays, and summarise an owned chunk range (its sum, and
// whether any incoming halo range touches a chunk the streaming pass already finished).
#include <hip/hip_runtime.h>

#include "../../include/ngsq_comm.h"
#include "kernels.h"

namespace ngsq {

__global__ __launch_bounds__(256) void k_halo_add(uint32_t *dst, const uint32_t *src, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        const uint32_t v = src[i];
        if (v) dst[i] += v; // ranges of different senders are added by successive launches: no atomics needed
    }
}

hipError_t launch_halo_add(uint32_t *dst, const uint32_t *src, uint64_t n, hipStream_t s) {
    if (!n) return hipSuccess;
    const uint64_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(k_halo_add, dim3((uint32_t)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, s, dst, src, n);
    return hipGetLastError();
}

struct InRanges {
    uint64_t r[2 * NGSQ_COMM_MAX_WORLD];
    uint32_t n;
};

// one block: out2[0] = sum of chunk_sums[b0, b1) mod 2^32, out2[1] = any flag set inside the incoming ranges
__global__ __launch_bounds__(1024) void k_range_summary(const uint32_t *chunk_sums, uint64_t b0, uint64_t b1, const uint8_t *flags,
                                                        InRanges in, uint32_t *out2) {
    __shared__ uint32_t s_sum[16], s_bad[16];
    uint32_t sum = 0, bad = 0;
    for (uint64_t i = b0 + threadIdx.x; i < b1; i += 1024) sum += chunk_sums[i];
    if (flags)
        for (uint32_t k = 0; k < in.n; k++)
            for (uint64_t i = in.r[2 * k] + threadIdx.x; i < in.r[2 * k + 1]; i += 1024) bad |= flags[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        sum += __shfl_xor(sum, o, 64);
        bad |= __shfl_xor(bad, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        s_sum[threadIdx.x >> 6] = sum;
        s_bad[threadIdx.x >> 6] = bad;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0, b = 0;
        for (int w = 0; w < 16; w++) {
            t += s_sum[w];
            b |= s_bad[w];
        }
        out2[0] = t;
        out2[1] = b ? 1u : 0u;
    }
}

hipError_t launch_range_summary(const uint32_t *chunk_sums, uint64_t b0, uint64_t b1, const uint8_t *flags, const uint64_t *in_ranges,
                                uint32_t n_in, uint32_t *out2, hipStream_t s) {
    InRanges in{};
    if (n_in > NGSQ_COMM_MAX_WORLD) return hipErrorInvalidValue;
    in.n = n_in;
    for (uint32_t k = 0; k < 2 * n_in; k++) in.r[k] = in_ranges[k];
    hipLaunchKernelGGL(k_range_summary, dim3(1), dim3(1024), 0, s, chunk_sums, b0, b1, flags, in, out2);
    return hipGetLastError();
}

} // namespace ngsq
