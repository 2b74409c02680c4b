// micro_qual.hip -- measurement harness (not product): what bounds the Quality Score
// kernel on gfx950?  Variants of the dense-stream LDS-histogram loop.
//   hipcc --offload-arch=gfx950 -O3 tools/micro_qual.hip -o /tmp/micro_qual && /tmp/micro_qual
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

constexpr uint32_t PITCH = 95;

// MODE 0: full (table address from cycle), 1: no LDS op (checksum), 2: lane-private address (no conflicts, no addr math),
// 3: mad24 address math, 4: natural [cycle][94] layout
template <int MODE>
__global__ __launch_bounds__(1024) void k(const uint8_t *__restrict__ qual, uint64_t n_bytes, uint32_t l, uint32_t R,
                                          unsigned long long *out) {
    extern __shared__ uint32_t s_q[];
    const uint32_t nb = 16u * R * PITCH;
    for (uint32_t i = threadIdx.x; i < nb; i += blockDim.x) s_q[i] = 0;
    __syncthreads();
    const uint64_t n_chunks = n_bytes / 16;
    const uint64_t per = (n_chunks + gridDim.x - 1) / gridDim.x;
    const uint64_t lo = min(per * blockIdx.x, n_chunks), hi = min(lo + per, n_chunks);
    uint32_t cyc = (uint32_t)(((lo + threadIdx.x) * 16) % l);
    const uint32_t step = (16u * blockDim.x) % l;
    const uint4 *src = reinterpret_cast<const uint4 *>(qual);
    uint32_t acc = 0;
    const uint32_t RP = R * PITCH;
    for (uint64_t g = lo + threadIdx.x; g < hi; g += blockDim.x) {
        const uint4 w = src[g];
        const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (uint32_t d = 0; d < 4; d++) {
#pragma unroll
            for (uint32_t k = 0; k < 4; k++) {
                const uint32_t q = (ww[d] >> (8 * k)) & 0xFFu;
                uint32_t c = cyc + 4 * d + k;
                c = min(c, c - l);
                if (MODE == 0) {
                    const uint32_t row = (c & 15u) * R + (c >> 4);
                    atomicAdd(&s_q[row * PITCH + q], 1u);
                } else if (MODE == 1) {
                    const uint32_t row = (c & 15u) * R + (c >> 4);
                    acc ^= row * PITCH + q;
                } else if (MODE == 2) {
                    atomicAdd(&s_q[threadIdx.x + 1024 * (q & 7)], 1u);
                } else if (MODE == 3) {
                    const uint32_t a = __umul24(c & 15u, RP) + __umul24(c >> 4, PITCH) + q;
                    atomicAdd(&s_q[a], 1u);
                } else if (MODE == 4) {
                    atomicAdd(&s_q[c * 94 + q], 1u);
                }
            }
        }
        cyc += step;
        cyc = min(cyc, cyc - l);
    }
    __syncthreads();
    unsigned long long t = acc;
    for (uint32_t i = threadIdx.x; i < nb; i += blockDim.x) t += s_q[i];
    if (t == 0xFFFFFFFFFFFFull) out[0] = t;
    if (threadIdx.x == 0) atomicAdd(&out[1], t);
}


// ---- window-per-lane variants: thread = (record, 16-cycle window w); unaligned 16-byte loads;
// table [q][CP] (CP = 16R rounded up to a multiple of 32 words): the bank depends only on the
// lane's (k, w), never on q.  ROT: records take their four dwords in a rotated order so that
// lanes of different records in one wave do not hit the same (cycle, q) word.
template <int ROT, int CHECK = 0, int COPIES = 1>
__global__ __launch_bounds__(1024) void kw(const uint8_t *__restrict__ qual, uint64_t n_rec, uint32_t l, uint32_t R,
                                           uint32_t CP, uint32_t magicR, unsigned long long *out, uint32_t RP = 0) {
    extern __shared__ uint32_t s_q[];
    if (RP == 0) RP = R;
    const uint32_t nb = 95u * CP * COPIES;
    for (uint32_t i = threadIdx.x; i < nb; i += blockDim.x) s_q[i] = 0;
    __syncthreads();
    const uint64_t n_win = n_rec * R;
    const uint64_t per = (n_win + gridDim.x - 1) / gridDim.x;
    const uint64_t lo = min(per * blockIdx.x, n_win), hi = min(lo + per, n_win);
    const uint32_t rem = l - 16u * (R - 1); // valid bytes of the last window
    const uint64_t rec_lo = lo / R;
    const uint32_t CP4 = CP * 4;
    for (uint64_t t = lo + threadIdx.x; t < hi; t += blockDim.x) {
        const uint32_t tl = (uint32_t)(t - rec_lo * R);      // block-local window index
        const uint32_t rl = __umulhi(tl, magicR);            // tl / R
        const uint32_t w = tl - rl * R;
        const uint64_t rec = rec_lo + rl;
        const uint8_t *p = qual + rec * (uint64_t)l + 16u * w;
        uint4 v;
        __builtin_memcpy(&v, p, 16);
        uint32_t ww[4] = {v.x, v.y, v.z, v.w};
        if (w == R - 1) { // bytes >= rem belong to the next record: send them to the trash column 94
#pragma unroll
            for (uint32_t d = 0; d < 4; d++) {
                const int nv = (int)rem - 4 * (int)d; // valid bytes in this dword
                const uint32_t keep = nv >= 4 ? 0xFFFFFFFFu : (nv <= 0 ? 0u : ((1u << (8 * nv)) - 1u));
                ww[d] = (ww[d] & keep) | (0x5E5E5E5Eu & ~keep);
            }
        }
        uint32_t base = 4u * w;
        if (COPIES > 1) base += ((uint32_t)(rec >> 2) & (COPIES - 1)) * (95u * CP * 4u);
        const uint32_t rot = ROT ? ((uint32_t)rec & 3u) : 0u;
        if (CHECK) {
            uint32_t hi = 0;
#pragma unroll
            for (uint32_t d = 0; d < 4; d++) hi |= ((ww[d] & 0x7F7F7F7Fu) + 0x21212121u) | ww[d];
            if (hi & 0x80808080u) { atomicAdd(&out[0], 1ull); continue; }
        }
#pragma unroll
        for (uint32_t dd = 0; dd < 4; dd++) {
            uint32_t x, kd;
            if (ROT) {
                const uint32_t d = (dd + rot) & 3u;
                x = d == 0 ? ww[0] : d == 1 ? ww[1] : d == 2 ? ww[2] : ww[3];
                kd = d * 4;
#pragma unroll
                for (uint32_t k = 0; k < 4; k++) {
                    const uint32_t q = (x >> (8 * k)) & 0xFFu;
                    const uint32_t a = __umul24(q, CP4) + base + (kd + k) * (4u * RP);
                    atomicAdd((uint32_t *)((char *)s_q + a), 1u);
                }
            } else {
                x = ww[dd];
#pragma unroll
                for (uint32_t k = 0; k < 4; k++) {
                    const uint32_t q = (x >> (8 * k)) & 0xFFu;
                    const uint32_t a = __umul24(q, CP4) + base;
                    atomicAdd((uint32_t *)((char *)s_q + a + (dd * 4 + k) * (4u * R)), 1u);
                }
            }
        }
    }
    __syncthreads();
    unsigned long long tsum = 0;
    for (uint32_t i = threadIdx.x; i < nb; i += blockDim.x) tsum += s_q[i];
    if (threadIdx.x == 0) atomicAdd(&out[1], tsum);
}

// pure streaming read of the same bytes (HBM ceiling for this access pattern)
__global__ __launch_bounds__(1024) void k_read(const uint8_t *__restrict__ qual, uint64_t n_bytes, unsigned long long *out) {
    const uint64_t n_chunks = n_bytes / 16;
    const uint64_t per = (n_chunks + gridDim.x - 1) / gridDim.x;
    const uint64_t lo = min(per * blockIdx.x, n_chunks), hi = min(lo + per, n_chunks);
    const uint4 *src = reinterpret_cast<const uint4 *>(qual);
    uint32_t acc = 0;
    for (uint64_t g = lo + threadIdx.x; g < hi; g += blockDim.x) {
        const uint4 w = src[g];
        acc ^= w.x ^ w.y ^ w.z ^ w.w;
    }
    if (acc == 0x12345) out[0] = acc;
}

int main(int argc, char **argv) {
    const uint64_t n_rec = argc > 1 ? strtoull(argv[1], 0, 10) : 40000000ull;
    const uint32_t l = 150, R = (l + 15) / 16;
    const uint64_t n_bytes = n_rec * l;
    uint8_t *d;
    unsigned long long *out;
    CK(hipMalloc((void **)&d, n_bytes));
    CK(hipMalloc((void **)&out, 16));
    CK(hipMemset(out, 0, 16));
    const size_t lds = 16 * R * PITCH * 4;
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for (int dist = 0; dist < 3; dist++) {
        // 0: all 37, 1: binned {2,11,25,37} 75% 37, 2: uniform 0..40
        std::vector<uint8_t> h(1 << 24);
        srand(1);
        for (auto &x : h) {
            int r = rand() % 100;
            x = dist == 0 ? 37 : dist == 1 ? (r < 75 ? 37 : r < 88 ? 25 : r < 95 ? 11 : 2) : (uint8_t)(rand() % 41);
        }
        for (uint64_t off = 0; off < n_bytes; off += h.size())
            CK(hipMemcpy(d + off, h.data(), std::min<uint64_t>(h.size(), n_bytes - off), hipMemcpyHostToDevice));
        printf("dist %d (%s)\n", dist, dist == 0 ? "all 37" : dist == 1 ? "binned 4 values" : "uniform 0..40");
        auto run = [&](const char *name, auto kern, int grid, size_t shm) {
            float best = 1e9;
            for (int it = 0; it < 4; it++) {
                CK(hipEventRecord(a));
                kern(grid, shm);
                CK(hipEventRecord(b));
                CK(hipEventSynchronize(b));
                float ms;
                CK(hipEventElapsedTime(&ms, a, b));
                if (ms < best) best = ms;
            }
            printf("  %-28s %8.3f ms  %8.1f GB/s\n", name, best, n_bytes / best / 1e6);
        };
        for (int per_cu = 1; per_cu <= 2; per_cu++) {
            int grid = 256 * per_cu;
            printf(" grid %d x 1024\n", grid);
            if (dist == 0 && per_cu == 2)
                run("read only", [&](int g, size_t) { hipLaunchKernelGGL(k_read, dim3(g), dim3(1024), 0, 0, d, n_bytes, out); }, grid, 0);
            hipFuncSetAttribute(reinterpret_cast<const void *>(k<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            hipFuncSetAttribute(reinterpret_cast<const void *>(k<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            hipFuncSetAttribute(reinterpret_cast<const void *>(k<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            hipFuncSetAttribute(reinterpret_cast<const void *>(k<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            hipFuncSetAttribute(reinterpret_cast<const void *>(k<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            run("0 rho layout (current)", [&](int g, size_t s) { hipLaunchKernelGGL(k<0>, dim3(g), dim3(1024), s, 0, d, n_bytes, l, R, out); }, grid, lds);
            run("1 no LDS op", [&](int g, size_t s) { hipLaunchKernelGGL(k<1>, dim3(g), dim3(1024), s, 0, d, n_bytes, l, R, out); }, grid, lds);
            run("2 lane-private atomics", [&](int g, size_t s) { hipLaunchKernelGGL(k<2>, dim3(g), dim3(1024), s, 0, d, n_bytes, l, R, out); }, grid, lds);
            run("3 rho layout, mad24", [&](int g, size_t s) { hipLaunchKernelGGL(k<3>, dim3(g), dim3(1024), s, 0, d, n_bytes, l, R, out); }, grid, lds);
            run("4 natural [cycle][94]", [&](int g, size_t s) { hipLaunchKernelGGL(k<4>, dim3(g), dim3(1024), s, 0, d, n_bytes, l, R, out); }, grid, lds);
            {
                const uint32_t CP = (16 * R + 31) / 32 * 32;
                const uint32_t magicR = (uint32_t)(((1ull << 32) + R - 1) / R);
                const size_t ldsw = 95 * CP * 4;
                hipFuncSetAttribute(reinterpret_cast<const void *>(kw<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                hipFuncSetAttribute(reinterpret_cast<const void *>(kw<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                run("5 window/lane [q][CP]", [&](int g, size_t s) { hipLaunchKernelGGL(kw<0>, dim3(g), dim3(1024), s, 0, d, n_rec, l, R, CP, magicR, out, 0); }, grid, ldsw);
                run("6 window/lane + dword rot", [&](int g, size_t s) { hipLaunchKernelGGL(kw<1>, dim3(g), dim3(1024), s, 0, d, n_rec, l, R, CP, magicR, out, 0); }, grid, ldsw);
                run("7 = 6 + validity check", [&](int g, size_t s) { hipLaunchKernelGGL((kw<1, 1>), dim3(g), dim3(1024), s, 0, d, n_rec, l, R, CP, magicR, out, 0); }, grid, ldsw);
                for (uint32_t RP : {11u, 13u}) {
                    const uint32_t CP2 = (16 * RP + 31) / 32 * 32;
                    hipFuncSetAttribute(reinterpret_cast<const void *>(kw<1, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                    char nm[64]; snprintf(nm, sizeof nm, "8 = 7 with RP=%u", RP);
                    run(nm, [&](int g, size_t s) { hipLaunchKernelGGL((kw<1, 1>), dim3(g), dim3(1024), s, 0, d, n_rec, l, R, CP2, magicR, out, RP); }, grid, 95 * CP2 * 4);
                }
                if (per_cu == 1) {
                    hipFuncSetAttribute(reinterpret_cast<const void *>(kw<1, 1, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                    run("9 = 7 with 2 table copies", [&](int g, size_t s) { hipLaunchKernelGGL((kw<1, 1, 2>), dim3(g), dim3(1024), s, 0, d, n_rec, l, R, CP, magicR, out, 0); }, grid, 2 * ldsw);
                }
            }
        }
    }
    return 0;
}
