// Which element of the ingest pipeline's call sequence makes a short kernel on stream `st` wait for a long kernel that was
// queued on stream `sb` just before it?  The sequence of bam_device_reader.cpp's load_chunk()/issue_inflate() with a spin
// kernel for the inflate, elements switched off one at a time.
//   hipcc --offload-arch=gfx950 -O2 tools/stream_order_probe.hip -o /tmp/sop && /tmp/sop
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void spin(unsigned long long cycles, unsigned *sink) {
    __shared__ unsigned s[1600];
    s[threadIdx.x] = threadIdx.x;
    const unsigned long long t0 = __builtin_readcyclecounter();
    unsigned x = s[threadIdx.x ^ 1];
    while (__builtin_readcyclecounter() - t0 < cycles) x = x * 1664525u + 1013904223u;
    if (x == 0xDEADBEEF) *sink = x;
}
__global__ void shortk(unsigned *sink) {
    if (threadIdx.x == 9999) *sink = 1;
}
__global__ void write_host(unsigned *mapped, unsigned n) { // results straight into host memory the device can address
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) mapped[i] = i;
}
__global__ void read_host(const unsigned *mapped, unsigned *dst, unsigned n) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = mapped[i];
}
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(e_)); exit(1); } } while (0)

int main() {
    unsigned *sink, *dbuf, *dsmall;
    CK(hipMalloc(&sink, 4));
    CK(hipMalloc(&dbuf, 128 << 20));
    CK(hipMalloc(&dsmall, 1 << 20));
    void *pinned, *pinned_small;
    CK(hipHostMalloc(&pinned, 96 << 20, hipHostMallocDefault));
    CK(hipHostMalloc(&pinned_small, 1 << 20, hipHostMallocDefault));
    std::vector<unsigned> pageable(1 << 18);
    hipStream_t st, sb, cs;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
    int lo = 0, hi = 0;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    CK(hipStreamCreateWithPriority(&sb, hipStreamNonBlocking, lo));
    hipEvent_t h2d_done, raw_free, inf_done;
    CK(hipEventCreateWithFlags(&h2d_done, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&raw_free, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&inf_done, hipEventDisableTiming));
    const unsigned long long cyc = 300000; // ~3 ms of the 100 MHz counter
    // bit k of `mask` switches element k ON
    auto trial = [&](unsigned mask, const char *what) {
        CK(hipDeviceSynchronize());
        double worst = 0, sum = 0;
        for (int it = 0; it < 5; it++) {
            if (mask & 1) { CK(hipMemcpyAsync(dbuf, pinned, 90 << 20, hipMemcpyHostToDevice, cs)); CK(hipEventRecord(h2d_done, cs)); }
            if (mask & 2) { CK(hipMemcpyAsync(dbuf + (100 << 18), dbuf + (101 << 18), 400, hipMemcpyDeviceToDevice, st)); }
            if (mask & 4) { CK(hipEventRecord(raw_free, st)); }
            if (mask & 1) CK(hipStreamWaitEvent(sb, h2d_done, 0));
            if (mask & 4) CK(hipStreamWaitEvent(sb, raw_free, 0));
            if (mask & 8) { CK(hipMemcpyAsync(dsmall, pinned_small, 200000, hipMemcpyHostToDevice, sb)); }
            if (mask & 16) CK(hipMemsetAsync(dsmall + 1000, 0, 4, sb));
            const double t0 = now();
            hipLaunchKernelGGL(spin, dim3(6144), dim3(64), 0, sb, cyc, sink); // "inflate": 24 waves per CU, 6400 B LDS each
            if (mask & 32) hipLaunchKernelGGL(shortk, dim3(1024), dim3(512), 0, sb, sink); // "crc"
            if (mask & 64) CK(hipMemcpyAsync(pinned_small, dsmall, 36000, hipMemcpyDeviceToHost, sb));
            if (mask & 128) CK(hipMemcpyAsync(pageable.data(), dsmall, 36000, hipMemcpyDeviceToHost, sb));
            CK(hipEventRecord(inf_done, sb));
            const double t1 = now();
            hipLaunchKernelGGL(shortk, dim3(4096), dim3(64), 0, st, sink); // "candidates"
            if (mask & 256) CK(hipMemcpyAsync(pageable.data(), dsmall, 196000, hipMemcpyDeviceToHost, st));
            if (mask & 512) CK(hipMemcpyAsync(pinned_small, dsmall, 196000, hipMemcpyDeviceToHost, st));
            if (mask & 1024) hipLaunchKernelGGL(write_host, dim3(192), dim3(256), 0, st, (unsigned *)pinned_small, 49000u);
            if (mask & 2048) CK(hipMemcpyAsync(dsmall, pageable.data(), 65536, hipMemcpyHostToDevice, st));
            if (mask & 4096) CK(hipMemcpyAsync(dsmall, pinned_small, 65536, hipMemcpyHostToDevice, st));
            if (mask & 8192) hipLaunchKernelGGL(read_host, dim3(64), dim3(256), 0, st, (const unsigned *)pinned_small, dsmall, 16384u);
            CK(hipStreamSynchronize(st));
            const double t2 = now();
            CK(hipEventSynchronize(inf_done));
            const double t3 = now();
            if (it) { sum += t2 - t1; worst = t2 - t1 > worst ? t2 - t1 : worst; }
            (void)t0; (void)t3;
        }
        printf("%-70s short kernel on st done after %.2f ms (worst %.2f)\n", what, sum / 4, worst);
    };
    trial(0, "nothing but the two kernels");
    trial(1, "+ big H2D on the copy stream, sb waits for it");
    trial(2, "+ small D2D copy on st in front");
    trial(4, "+ event recorded on st, sb waits for it");
    trial(8, "+ small pinned H2D on sb in front of the long kernel");
    trial(16, "+ 4-byte memset on sb in front of the long kernel");
    trial(32, "+ short kernel on sb behind the long one");
    trial(64, "+ pinned D2H on sb behind the long kernel");
    trial(128, "+ PAGEABLE D2H on sb behind the long kernel");
    trial(256, "+ pageable D2H on st behind the short kernel");
    const unsigned all = 1 | 2 | 4 | 8 | 16 | 32 | 64 | 256;
    trial(all, "everything (pinned status)");
    const char *names[] = {"big H2D + wait", "D2D on st", "event st -> sb", "pinned H2D on sb", "memset on sb", "short kernel on sb", "pinned D2H on sb", "", "pageable D2H on st"};
    for (int k = 0; k < 9; k++) {
        if (!(all >> k & 1)) continue;
        char buf[128];
        snprintf(buf, sizeof buf, "everything but: %s", names[k]);
        trial(all & ~(1u << k), buf);
    }
    const unsigned base = all & ~256u;
    trial(base | 512, "everything, the D2H on st into PINNED memory");
    trial(base | 1024, "everything, the results written to mapped host memory by a kernel");
    trial(base | 2048, "everything, no D2H but a PAGEABLE H2D (64 KB) on st");
    trial(base | 4096, "everything, no D2H but a PINNED H2D (64 KB) on st");
    trial(base | 8192, "everything, no D2H but a kernel reading 64 KB of mapped host memory");
    trial(2 | 4, "only: D2D on st + event st -> sb");
    trial(1 | 4, "only: big H2D + wait, event st -> sb");
    trial(4 | 8, "only: event st -> sb + pinned H2D on sb");
    trial(2 | 4 | 8, "only: D2D on st + event st -> sb + pinned H2D on sb");
    trial(4 | 16, "only: event st -> sb + memset on sb");
    return 0;
}
