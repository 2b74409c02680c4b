// Do two kernels on two HIP streams run at the same time on this box?  A spinning kernel of few small workgroups on
// stream A, the same on stream B: wall time of both together against one alone, for several shapes.
//   hipcc --offload-arch=gfx950 -O2 tools/two_stream_probe.hip -o /tmp/two_stream_probe && /tmp/two_stream_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

__global__ void spin(unsigned long long cycles, unsigned *sink) {
    const unsigned long long t0 = __builtin_readcyclecounter();
    unsigned x = threadIdx.x;
    while (__builtin_readcyclecounter() - t0 < cycles) x = x * 1664525u + 1013904223u;
    if (x == 0xDEADBEEF) *sink = x;
}
template <int LDS> __global__ void spin_lds(unsigned long long cycles, unsigned *sink) {
    __shared__ unsigned s[LDS / 4];
    s[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    unsigned x = s[(threadIdx.x + 1) & 63];
    while (__builtin_readcyclecounter() - t0 < cycles) x = x * 1664525u + 1013904223u;
    if (x == 0xDEADBEEF) *sink = x;
}
// the inflate kernel's footprint: 72 VGPRs, 6400 bytes of LDS, one wave per workgroup
template <int LDS> __global__ __attribute__((amdgpu_num_vgpr(72))) void spin_fat(unsigned long long cycles, unsigned *sink) {
    __shared__ unsigned s[LDS / 4];
    s[threadIdx.x] = threadIdx.x;
    __syncthreads();
    unsigned r[60];
#pragma unroll
    for (int k = 0; k < 60; k++) r[k] = s[(threadIdx.x + k) & 63] + k;
    const unsigned long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < cycles) {
#pragma unroll
        for (int k = 0; k < 60; k++) r[k] = r[k] * 1664525u + r[(k + 1) % 60];
    }
    unsigned x = 0;
#pragma unroll
    for (int k = 0; k < 60; k++) x ^= r[k];
    if (x == 0xDEADBEEF) *sink = x;
}
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(e_)); exit(1); } } while (0)

int main() {
    unsigned *sink;
    CK(hipMalloc(&sink, 4));
    hipStream_t a, b, lowp;
    CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    int lo = 0, hi = 0;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    CK(hipStreamCreateWithPriority(&lowp, hipStreamNonBlocking, lo));
    printf("priority range: least %d greatest %d\n", lo, hi);
    const unsigned long long cyc = 200000000ull / 100; // ~2 M cycles of the 100 MHz counter?  calibrated below
    auto run = [&](const char *what, auto launchA, auto launchB) {
        launchA(a); CK(hipDeviceSynchronize());
        double t = now(); launchA(a); CK(hipDeviceSynchronize()); const double ta = now() - t;
        t = now(); launchB(b); CK(hipDeviceSynchronize()); const double tb = now() - t;
        t = now(); launchA(a); launchB(b); CK(hipDeviceSynchronize()); const double tab = now() - t;
        t = now(); launchA(lowp); launchB(b); CK(hipDeviceSynchronize()); const double tlow = now() - t;
        printf("%-58s A %.2f ms, B %.2f ms, A|B %.2f ms, A(low priority)|B %.2f ms  -> %s\n", what, ta, tb, tab, tlow,
               tab < 0.75 * (ta + tb) ? "CONCURRENT" : "serial");
    };
    run("A: 64 WGs x 64 thr, B: the same", [&](hipStream_t s) { hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, s, cyc, sink); },
        [&](hipStream_t s) { hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, s, cyc, sink); });
    run("A: 4096 WGs x 64 thr (16 per CU), B: 256 WGs x 256 thr", [&](hipStream_t s) { hipLaunchKernelGGL(spin, dim3(4096), dim3(64), 0, s, cyc, sink); },
        [&](hipStream_t s) { hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s, cyc, sink); });
    run("A: 6144 WGs x 64 thr, 6400 B LDS each (24 per CU), B: 256 x 256", [&](hipStream_t s) { hipLaunchKernelGGL(spin_lds<6400>, dim3(6144), dim3(64), 0, s, cyc, sink); },
        [&](hipStream_t s) { hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s, cyc, sink); });
    run("A: 20000 WGs x 64 thr, 6400 B LDS each (3 rounds), B: 256 x 256", [&](hipStream_t s) { hipLaunchKernelGGL(spin_lds<6400>, dim3(20000), dim3(64), 0, s, cyc / 3, sink); },
        [&](hipStream_t s) { hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s, cyc, sink); });
    run("A: 6400 x 64 thr, 72 VGPRs + 6400 B LDS (25 per CU), B: 256 x 256", [&](hipStream_t s) { hipLaunchKernelGGL(spin_fat<6400>, dim3(6400), dim3(64), 0, s, cyc, sink); },
        [&](hipStream_t s) { hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s, cyc, sink); });
    run("A: 6400 x 64 thr, 72 VGPRs + 6400 B LDS (25 per CU), B: 1024 x 64", [&](hipStream_t s) { hipLaunchKernelGGL(spin_fat<6400>, dim3(6400), dim3(64), 0, s, cyc, sink); },
        [&](hipStream_t s) { hipLaunchKernelGGL(spin, dim3(1024), dim3(64), 0, s, cyc, sink); });
    run("A: 20000 x 64 thr, 72 VGPRs + 6400 B LDS (3+ rounds), B: 1024 x 64", [&](hipStream_t s) { hipLaunchKernelGGL(spin_fat<6400>, dim3(20000), dim3(64), 0, s, cyc / 3, sink); },
        [&](hipStream_t s) { hipLaunchKernelGGL(spin, dim3(1024), dim3(64), 0, s, cyc, sink); });
    run("A: 4096 x 64 thr, 72 VGPRs + 6400 B LDS (16 per CU), B: 256 x 256", [&](hipStream_t s) { hipLaunchKernelGGL(spin_fat<6400>, dim3(4096), dim3(64), 0, s, cyc, sink); },
        [&](hipStream_t s) { hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s, cyc, sink); });
    return 0;
}
