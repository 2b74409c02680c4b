// reader_paths_probe.cpp -- what does a byte of a file cost the host on its way to the GPU?  
// Four ways of getting FILE's first `--mb` MiB into device memory, each reported as GB/s and as GB/s PER HOST CORE-SECOND
// (user + system CPU time of the whole process over the leg: the cores the path keeps busy are what a node of eight
// GPUs under a CPU quota runs out of):
//   pread    NT threads pread() 4 MiB pieces into pinned buffers, hipMemcpyAsync from there   (the library's path)
//   mmapcpy  hipMemcpy straight from a read-only mapping of the file (the runtime stages it through its own pinned buffer)
//   mmapreg  hipHostRegister of the mapping piece by piece, hipMemcpyAsync from the registered pages (no CPU copy at all)
//   odirect  O_DIRECT pread() into the pinned buffers (no page cache, no kernel copy), hipMemcpyAsync from there
// `--cold` drops the file from the page cache before every leg (posix_fadvise DONTNEED after fsync; works without root
// on the boxes' overlay filesystem for files the process wrote itself).
//   hipcc -O2 -pthread tools/reader_paths_probe.cpp -o /tmp/rpp && /tmp/rpp FILE [--mb 4096] [--threads 4] [--cold]
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <sys/resource.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static double cpu_s() {
    struct rusage u;
    getrusage(RUSAGE_SELF, &u);
    return u.ru_utime.tv_sec + u.ru_utime.tv_usec * 1e-6 + u.ru_stime.tv_sec + u.ru_stime.tv_usec * 1e-6;
}
#define CK(x)                                                                    \
    do {                                                                         \
        hipError_t e_ = (x);                                                     \
        if (e_ != hipSuccess) {                                                  \
            printf("  %s: %s\n", #x, hipGetErrorString(e_));                     \
            (void)hipGetLastError();                                             \
            return false;                                                        \
        }                                                                        \
    } while (0)

static const size_t PIECE = (size_t)4 << 20, SLOT = (size_t)64 << 20;
static size_t g_total = 0;
static int g_nt = 4;
static uint8_t *g_dev = nullptr;
static uint8_t *g_pin[2] = {nullptr, nullptr};

static void drop(int fd) {
    fsync(fd);
    posix_fadvise(fd, 0, 0, POSIX_FADV_DONTNEED);
}

// NT threads fill slot after slot (64 MiB) of two pinned buffers; each slot is copied to the device when it is full
static bool leg_pread(const char *path, bool direct) {
    const int fd = open(path, O_RDONLY | (direct ? O_DIRECT : 0));
    if (fd < 0) { printf("  open(%s) failed\n", direct ? "O_DIRECT" : ""); return false; }
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t done[2];
    CK(hipEventCreateWithFlags(&done[0], hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&done[1], hipEventDisableTiming));
    bool used[2] = {false, false};
    bool ok = true;
    size_t k = 0;
    for (size_t pos = 0; pos < g_total && ok; pos += SLOT, k++) {
        const size_t want = std::min(SLOT, g_total - pos);
        uint8_t *buf = g_pin[k & 1];
        if (used[k & 1]) CK(hipEventSynchronize(done[k & 1]));
        std::atomic<size_t> next{0};
        std::atomic<bool> bad{false};
        std::vector<std::thread> w;
        for (int t = 0; t < g_nt; t++)
            w.emplace_back([&] {
                for (;;) {
                    const size_t lo = next.fetch_add(PIECE);
                    if (lo >= want) return;
                    size_t hi = std::min(want, lo + PIECE), d = 0;
                    if (direct) hi = lo + ((hi - lo + 4095) & ~(size_t)4095); // O_DIRECT: whole sectors (the buffer has room)
                    while (lo + d < hi) {
                        const ssize_t r = pread(fd, buf + lo + d, hi - lo - d, (off_t)(pos + lo + d));
                        if (r < 0) { bad = true; return; }
                        if (r == 0) break;
                        d += (size_t)r;
                    }
                }
            });
        for (auto &x : w) x.join();
        if (bad) { printf("  pread failed (errno %d)\n", errno); ok = false; break; }
        CK(hipMemcpyAsync(g_dev + pos, buf, want, hipMemcpyHostToDevice, s));
        CK(hipEventRecord(done[k & 1], s));
        used[k & 1] = true;
    }
    CK(hipStreamSynchronize(s));
    (void)hipEventDestroy(done[0]);
    (void)hipEventDestroy(done[1]);
    (void)hipStreamDestroy(s);
    close(fd);
    return ok;
}

static bool leg_mmapcpy(const char *path) {
    const int fd = open(path, O_RDONLY);
    void *m = mmap(nullptr, g_total, PROT_READ, MAP_PRIVATE, fd, 0);
    if (m == MAP_FAILED) { printf("  mmap failed\n"); return false; }
    madvise(m, g_total, MADV_SEQUENTIAL);
    for (size_t pos = 0; pos < g_total; pos += SLOT) CK(hipMemcpy(g_dev + pos, (uint8_t *)m + pos, std::min(SLOT, g_total - pos), hipMemcpyHostToDevice));
    munmap(m, g_total);
    close(fd);
    return true;
}

static bool leg_mmapreg(const char *path, bool shared, unsigned flags) {
    const int fd = open(path, O_RDONLY);
    void *m = mmap(nullptr, g_total, PROT_READ, (shared ? MAP_SHARED : MAP_PRIVATE) | MAP_POPULATE, fd, 0);
    if (m == MAP_FAILED) { printf("  mmap failed\n"); return false; }
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    bool ok = true;
    for (size_t pos = 0; pos < g_total; pos += SLOT) {
        const size_t want = std::min(SLOT, g_total - pos);
        hipError_t e = hipHostRegister((uint8_t *)m + pos, want, flags);
        if (e != hipSuccess) {
            printf("  hipHostRegister(%s mapping, flags 0x%x): %s\n", shared ? "shared" : "private", flags, hipGetErrorString(e));
            (void)hipGetLastError();
            ok = false;
            break;
        }
        CK(hipMemcpyAsync(g_dev + pos, (uint8_t *)m + pos, want, hipMemcpyHostToDevice, s));
        CK(hipStreamSynchronize(s));
        CK(hipHostUnregister((uint8_t *)m + pos));
    }
    (void)hipStreamDestroy(s);
    munmap(m, g_total);
    close(fd);
    return ok;
}

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    const char *path = argv[1];
    size_t mb = 4096;
    bool cold = false;
    for (int i = 2; i < argc; i++) {
        if (!strcmp(argv[i], "--mb") && i + 1 < argc) mb = (size_t)atol(argv[++i]);
        else if (!strcmp(argv[i], "--threads") && i + 1 < argc) g_nt = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--cold")) cold = true;
    }
    struct stat st;
    if (stat(path, &st) != 0) { printf("no such file\n"); return 1; }
    g_total = std::min<size_t>((size_t)st.st_size, mb << 20) & ~(SLOT - 1);
    if (hipMalloc((void **)&g_dev, g_total) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    for (auto &p : g_pin)
        if (hipHostMalloc((void **)&p, SLOT + 4096, hipHostMallocDefault) != hipSuccess) { printf("hipHostMalloc failed\n"); return 1; }
    memset(g_pin[0], 1, SLOT);
    memset(g_pin[1], 1, SLOT);
    const int fd0 = open(path, O_RDONLY);
    // a checksum of the device copy: every leg must deliver the same bytes
    std::vector<uint8_t> back(1 << 20);
    auto sample = [&]() {
        unsigned long long h = 1469598103934665603ull;
        for (size_t pos = 0; pos < g_total; pos += g_total / 16) {
            (void)hipMemcpy(back.data(), g_dev + pos, std::min(back.size(), g_total - pos), hipMemcpyDeviceToHost);
            for (size_t i = 0; i < std::min(back.size(), g_total - pos); i += 97) h = (h ^ back[i]) * 1099511628211ull;
        }
        return h;
    };
    printf("%s: %.2f GB per leg, %d read threads, %s page cache\n", path, g_total / 1e9, g_nt, cold ? "COLD" : "warm");
    struct Leg { const char *name; int kind; };
    const Leg legs[] = {{"pread", 0}, {"mmapcpy", 1}, {"mmapreg(private,default)", 2}, {"mmapreg(shared,default)", 3}, {"mmapreg(shared,readonly)", 4}, {"odirect", 5}};
    for (int rep = 0; rep < 2; rep++)
        for (const Leg &l : legs) {
            if (cold) drop(fd0);
            (void)hipMemset(g_dev, 0, g_total);
            (void)hipDeviceSynchronize();
            const double t0 = now(), c0 = cpu_s();
            bool ok = false;
            switch (l.kind) {
            case 0: ok = leg_pread(path, false); break;
            case 1: ok = leg_mmapcpy(path); break;
            case 2: ok = leg_mmapreg(path, false, hipHostRegisterDefault); break;
            case 3: ok = leg_mmapreg(path, true, hipHostRegisterDefault); break;
            case 4: ok = leg_mmapreg(path, true, hipHostRegisterReadOnly); break;
            case 5: ok = leg_pread(path, true); break;
            }
            const double t = now() - t0, c = cpu_s() - c0;
            if (ok) printf("%-26s %6.3f s = %5.1f GB/s wall   cpu %6.3f s = %5.1f GB per core-second   sum %016llx\n", l.name, t, g_total / t / 1e9, c,
                           g_total / c / 1e9, sample());
            else printf("%-26s failed\n", l.name);
            fflush(stdout);
        }
    return 0;
}
