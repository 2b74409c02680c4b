// reader_probe.cpp -- how fast can N threads copy a file out of the page cache into (pinned) memory?
// Measurement aid for the device ingest's reader thread; not part of the library.
//   hipcc -O2 -pthread tools/reader_probe.cpp -o /tmp/reader_probe && /tmp/reader_probe FILE
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <sched.h>
#include <sys/mman.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static bool pin_node(int node) {
    char p[128];
    snprintf(p, sizeof p, "/sys/devices/system/node/node%d/cpulist", node);
    FILE *f = fopen(p, "r");
    if (!f) return false;
    char line[4096];
    if (!fgets(line, sizeof line, f)) { fclose(f); return false; }
    fclose(f);
    cpu_set_t set;
    CPU_ZERO(&set);
    for (char *s = line; *s;) {
        int a = strtol(s, &s, 10), b = a;
        if (*s == '-') b = strtol(s + 1, &s, 10);
        for (int c = a; c <= b; c++) CPU_SET(c, &set);
        if (*s == ',') s++; else break;
    }
    return sched_setaffinity(0, sizeof set, &set) == 0;
}

int main(int argc, char **argv) {
    const char *path = argv[1];
    int fd = open(path, O_RDONLY);
    const size_t total = std::min<size_t>((size_t)lseek(fd, 0, SEEK_END), (size_t)2 << 30);
    for (int node : {-1, 0, 1}) {
        std::thread([&] {
            if (node >= 0 && !pin_node(node)) { printf("node %d: cannot pin\n", node); return; }
            for (int pinned : {0, 1}) {
                uint8_t *buf = nullptr;
                if (pinned) { if (hipHostMalloc((void **)&buf, total, hipHostMallocDefault) != hipSuccess) { printf("hipHostMalloc failed\n"); return; } }
                else { buf = (uint8_t *)mmap(0, total, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0); }
                memset(buf, 1, total);
                for (int NT : {1, 4, 8, 14, 28}) for (size_t STEP : {(size_t)16 << 20, (size_t)64 << 20, (size_t)256 << 20}) {
                    double best = 1e30;
                    for (int rep = 0; rep < 3; rep++) {
                        double t0 = now();
                        size_t pos = 0;
                        while (pos < total) {
                            size_t want = std::min(STEP, total - pos), per = (want + NT - 1) / NT;
                            std::vector<std::thread> w;
                            for (int t = 0; t < NT; t++) w.emplace_back([&, t] {
                                size_t lo = std::min(want, per * t), hi = std::min(want, lo + per), d = 0;
                                while (lo + d < hi) { ssize_t r = pread(fd, buf + pos + lo + d, hi - lo - d, pos + lo + d); if (r <= 0) break; d += r; }
                            });
                            for (auto &x : w) x.join();
                            pos += want;
                        }
                        best = std::min(best, now() - t0);
                    }
                    printf("node %2d %s NT %2d step %3zu MiB: %6.1f GB/s\n", node, pinned ? "pinned" : "anon  ", NT, STEP >> 20, total / best / 1e6);
                }
                if (pinned) (void)hipHostFree(buf); else munmap(buf, total);
            }
        }).join();
    }
    return 0;
}
