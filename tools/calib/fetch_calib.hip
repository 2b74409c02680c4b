// fetch_calib.hip -- what FETCH_SIZE says about 16-byte-per-lane reads that are NOT aligned to the cache line, against a byte count
// that is known (MI355X_MICROARCH.md, HBM: "Other access widths are uncalibrated: calibrate on a known byte count in your own
// access pattern").  Four patterns over the same buffer, each kernel reads every byte of its span once:
//   aligned     lane l of step t reads 16 bytes at 1024 t + 16 l                      (the fixed-pitch QUAL kernel's pattern)
//   shifted     the same + 8                                                          (every fourth window straddles a 64-byte sector)
//   ragged      pseudo-records of W windows + a tail of G bytes that is skipped: the window of lane l is record-relative, so
//               the wave's 64 windows cover 1024 + (about 64 / W) G bytes at a changing alignment (k_qual_ragged's pass A)
//   ragged_tail the same, and afterwards one 16-byte load per record at its tail (k_qual_ragged's pass B)
// Build: hipcc -O3 --offload-arch=gfx950 tools/calib/fetch_calib.hip -o /tmp/fetch_calib ; run under rocprofv3 --pmc FETCH_SIZE
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

template <int MODE>
__global__ __launch_bounds__(1024) void k_calib(const uint8_t *__restrict__ p, uint64_t n_win, uint32_t W, uint32_t G, uint32_t *sink) {
    const uint64_t wave = (blockIdx.x * 1024ull + threadIdx.x) >> 6, n_waves = gridDim.x * 16ull;
    const uint32_t lane = threadIdx.x & 63;
    uint32_t acc = 0;
    const uint64_t rec_bytes = 16ull * W + G;
    for (uint64_t t = wave; t * 64 < n_win; t += n_waves) {
        const uint64_t w = t * 64 + lane;
        uint64_t at;
        if (MODE == 0) at = 16 * w;
        else if (MODE == 1) at = 16 * w + 8;
        else at = (w / W) * rec_bytes + 16 * (w % W);
        uint4 v;
        __builtin_memcpy(&v, p + at, 16);
        acc += v.x ^ v.y ^ v.z ^ v.w;
        if (MODE == 3) {   // the tails of the records whose last window this step holds: lane = record, as pass B
            const uint64_t r0 = (t * 64) / W, r1 = (t * 64 + 63) / W;
            if (r0 + lane <= r1 && ((r0 + lane) * W + W - 1) / 64 == t) {
                uint4 u;
                __builtin_memcpy(&u, p + (r0 + lane) * rec_bytes + 16ull * W, 16);
                acc += u.x;
            }
        }
    }
    if (acc == 0x12345678u) *sink = acc;
}

int main(int argc, char **argv) {
    const uint64_t n_bytes = (argc > 1 ? strtoull(argv[1], 0, 10) : 4096ull) << 20;
    const uint32_t W = argc > 2 ? atoi(argv[2]) : 10, G = argc > 3 ? atoi(argv[3]) : 15;
    uint8_t *d;
    uint32_t *sink;
    if (hipMalloc(&d, n_bytes + 4096) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) return 1;
    hipMemset(d, 1, n_bytes + 4096);
    hipEvent_t a, b;
    hipEventCreate(&a), hipEventCreate(&b);
    for (int mode = 0; mode < 4; mode++) {
        const uint64_t n_win = mode < 2 ? n_bytes / 16 : n_bytes / (16ull * W + G) * W;
        const uint64_t span = mode < 2 ? n_win * 16 : n_win / W * (16ull * W + G);
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(a, 0);
            switch (mode) {
            case 0: hipLaunchKernelGGL(k_calib<0>, dim3(512), dim3(1024), 0, 0, d, n_win, W, G, sink); break;
            case 1: hipLaunchKernelGGL(k_calib<1>, dim3(512), dim3(1024), 0, 0, d, n_win, W, G, sink); break;
            case 2: hipLaunchKernelGGL(k_calib<2>, dim3(512), dim3(1024), 0, 0, d, n_win, W, G, sink); break;
            default: hipLaunchKernelGGL(k_calib<3>, dim3(512), dim3(1024), 0, 0, d, n_win, W, G, sink); break;
            }
            hipEventRecord(b, 0);
            hipEventSynchronize(b);
            float ms = 0;
            hipEventElapsedTime(&ms, a, b);
            if (rep == 2) printf("mode %d  span %llu bytes  windows %llu (%llu bytes loaded)  %.3f ms  %.0f GB/s of span\n", mode, (unsigned long long)span,
                                 (unsigned long long)n_win, (unsigned long long)n_win * 16, ms, span / ms / 1e6);
        }
    }
    return 0;
}
