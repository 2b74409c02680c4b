// init_probe.cpp -- where the ~250 ms between main() and the first usable HIP context go.
//   hipcc -O2 tools/init_probe.cpp -o /tmp/init_probe && /tmp/init_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k_nop(int *p) { if (p) *p = 1; }
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    double t0 = now(), t = t0;
    auto lap = [&](const char *what) { double n = now(); printf("%-34s %8.1f ms (at %8.1f)\n", what, n - t, n - t0); t = n; };
    (void)hipInit(0); lap("hipInit");
    int n = 0; (void)hipGetDeviceCount(&n); lap("hipGetDeviceCount");
    (void)hipSetDevice(0); lap("hipSetDevice");
    (void)hipFree(nullptr); lap("hipFree(0) (context)");
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0); lap("hipGetDeviceProperties");
    hipStream_t s; (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking); lap("hipStreamCreate");
    void *d = nullptr; (void)hipMalloc(&d, 1 << 20); lap("hipMalloc 1 MiB");
    void *d2 = nullptr; (void)hipMalloc(&d2, (size_t)1 << 30); lap("hipMalloc 1 GiB");
    hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, s, (int *)d); (void)hipStreamSynchronize(s); lap("first kernel launch + sync");
    void *h = nullptr; (void)hipHostMalloc(&h, 64 << 20, hipHostMallocDefault); lap("hipHostMalloc 64 MiB");
    (void)hipMemcpyAsync(d2, h, 64 << 20, hipMemcpyHostToDevice, s); (void)hipStreamSynchronize(s); lap("first H2D 64 MiB");
    return 0;
}
