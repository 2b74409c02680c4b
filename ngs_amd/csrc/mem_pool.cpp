// mem_pool.cpp -- see mem_pool.h
#include "mem_pool.h"

#include <hip/hip_runtime.h>
#include <sys/mman.h>

#include <algorithm>
#include <cstdlib>
#include <map>
#include <mutex>
#include <vector>

namespace ngsq {
namespace {

constexpr size_t POOL_MIN = (size_t)1 << 20; // smaller blocks are not worth keeping

constexpr size_t PIN_PIECE = (size_t)32 << 20; // pinned host blocks are registered with HIP in pieces of this size

struct Block {
    void *p;
    size_t bytes;
    int device; // >= 0: memory of that device; < 0: pinned host memory filled for device -(device + 1) (its NUMA node)
    std::vector<uint8_t> reg; // pinned blocks: which pieces are registered
};

struct Pool {
    std::mutex mu;
    std::vector<Block> blocks;
    std::map<const void *, Block> pinned_live; // pinned blocks handed out, by base address
    size_t cached[2] = {0, 0}; // device | pinned
    size_t limit[2]; // device | pinned
    std::map<int, std::vector<hipStream_t>> streams[2]; // by device: normal | lowest priority
    std::map<int, std::vector<hipEvent_t>> events;
    Pool() {
        // what one ingest pipeline at its largest chunk size (1 GiB) gives back: two raw buffers + two device buffers for the
        // compressed bytes + a batch's columns and the record index ~ 4.5 GiB of device memory, two pinned buffers = 1 GiB
        // (the cache used to hold on to 12 GiB of each; at 4 GiB every other scan with 1 GiB chunks allocated one
        // of its buffers afresh: 0.26 s instead of 0.20).  NGSQ_POOL_MB sets both.
        const char *e = getenv("NGSQ_POOL_MB");
        limit[0] = (size_t)(e ? strtoull(e, nullptr, 10) : 6144ull) << 20;
        limit[1] = (size_t)(e ? strtoull(e, nullptr, 10) : 1536ull) << 20;
    }
    // smallest cached block of that kind that holds `bytes` without wasting more than it holds
    bool take(int device, size_t bytes, Block *out) {
        std::lock_guard<std::mutex> g(mu);
        size_t best = blocks.size();
        for (size_t k = 0; k < blocks.size(); k++) {
            const Block &b = blocks[k];
            if (b.device != device || b.bytes < bytes || b.bytes > 2 * bytes) continue;
            if (best == blocks.size() || b.bytes < blocks[best].bytes) best = k;
        }
        if (best == blocks.size()) return false;
        *out = blocks[best];
        cached[device < 0] -= out->bytes;
        blocks.erase(blocks.begin() + (long)best);
        return true;
    }
    bool put(const Block &b) {
        std::lock_guard<std::mutex> g(mu);
        if (b.bytes < POOL_MIN || cached[b.device < 0] + b.bytes > limit[b.device < 0]) return false;
        blocks.push_back(b);
        cached[b.device < 0] += b.bytes;
        return true;
    }
};

Pool &pool() {
    static Pool *p = new Pool(); // never destroyed: the HIP runtime may be gone by the time static destructors run
    return *p;
}

} // namespace

hipError_t pool_device_alloc(void **p, size_t bytes, size_t *got) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    Block b{};
    if (bytes >= POOL_MIN && pool().take(dev, bytes, &b)) {
        *p = b.p;
        *got = b.bytes;
        return hipSuccess;
    }
    e = hipMalloc(p, bytes);
    if (e != hipSuccess && pool_trim()) { // the cache may be what is in the way
        (void)hipGetLastError();
        e = hipMalloc(p, bytes);
    }
    *got = bytes;
    return e;
}

void pool_device_free(void *p, size_t got) {
    if (!p) return;
    int cur = 0;
    if (got >= POOL_MIN && hipGetDevice(&cur) == hipSuccess) {
        int dev = cur;
        hipPointerAttribute_t attr{};
        if (hipPointerGetAttributes(&attr, p) == hipSuccess) dev = attr.device;
        // hipFree() waits for the device; a block that goes to the cache may be handed to another stream next
        if (dev != cur) (void)hipSetDevice(dev);
        (void)hipDeviceSynchronize();
        if (dev != cur) (void)hipSetDevice(cur);
        if (pool().put(Block{p, got, dev, {}})) return;
    }
    (void)hipFree(p);
}

static void release_pinned(Block &b) {
    for (size_t k = 0; k < b.reg.size(); k++)
        if (b.reg[k]) (void)hipHostUnregister(static_cast<uint8_t *>(b.p) + k * PIN_PIECE);
    munmap(b.p, b.bytes);
}

hipError_t pool_pinned_alloc(void **p, size_t bytes, size_t *got) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    Block b{};
    if (!(bytes >= POOL_MIN && pool().take(-(dev + 1), bytes, &b))) {
        const size_t rounded = (bytes + PIN_PIECE - 1) / PIN_PIECE * PIN_PIECE;
        void *m = mmap(nullptr, rounded, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (m == MAP_FAILED) return hipErrorOutOfMemory;
        // Transparent huge pages where the kernel grants them on request: the runtime pins (and the kernel, when the process ends,
        // unpins) these buffers page by page -- 2 MiB pages make that 512 times fewer.  NGSQ_PINNED_THP=0 turns the request off.
        {
            static const bool thp = [] { const char *e = getenv("NGSQ_PINNED_THP"); return !e || atoi(e) != 0; }();
            if (thp) (void)madvise(m, rounded, MADV_HUGEPAGE);
        }
        b = Block{m, rounded, -(dev + 1), std::vector<uint8_t>(rounded / PIN_PIECE, 0)};
    }
    *p = b.p;
    *got = b.bytes;
    std::lock_guard<std::mutex> g(pool().mu);
    pool().pinned_live[b.p] = std::move(b);
    return hipSuccess;
}

void pool_pinned_free(void *p, size_t) {
    if (!p) return;
    Block b{};
    {
        std::lock_guard<std::mutex> g(pool().mu);
        auto it = pool().pinned_live.find(p);
        if (it == pool().pinned_live.end()) return; // not one of ours
        b = std::move(it->second);
        pool().pinned_live.erase(it);
    }
    if (b.bytes >= POOL_MIN && pool().put(b)) return;
    release_pinned(b);
}

hipError_t pool_pinned_h2d(void *dst, const void *block, size_t off, size_t len, hipStream_t s) {
    if (!len) return hipSuccess;
    uint8_t *base = nullptr;
    size_t bytes = 0;
    std::vector<size_t> todo; // pieces to register (this block is filled and copied from by ONE thread)
    {
        std::lock_guard<std::mutex> g(pool().mu);
        auto it = pool().pinned_live.find(block);
        if (it == pool().pinned_live.end() || off + len > it->second.bytes) return hipErrorInvalidValue;
        base = static_cast<uint8_t *>(it->second.p);
        bytes = it->second.bytes;
        for (size_t k = off / PIN_PIECE; k <= (off + len - 1) / PIN_PIECE; k++)
            if (!it->second.reg[k]) todo.push_back(k);
    }
    for (size_t k : todo) {
        hipError_t e = hipHostRegister(base + k * PIN_PIECE, std::min(PIN_PIECE, bytes - k * PIN_PIECE), hipHostRegisterDefault);
        if (e != hipSuccess) return e;
        std::lock_guard<std::mutex> g(pool().mu);
        auto it = pool().pinned_live.find(block);
        if (it != pool().pinned_live.end()) it->second.reg[k] = 1;
    }
    for (size_t o = off; o < off + len;) { // a copy may not span two registrations
        const size_t n = std::min(off + len, (o / PIN_PIECE + 1) * PIN_PIECE) - o;
        hipError_t e = hipMemcpyAsync(static_cast<uint8_t *>(dst) + (o - off), base + o, n, hipMemcpyHostToDevice, s);
        if (e != hipSuccess) return e;
        o += n;
    }
    return hipSuccess;
}

hipError_t pool_stream_get(bool low_priority, hipStream_t *s) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    {
        std::lock_guard<std::mutex> g(pool().mu);
        auto &v = pool().streams[low_priority][dev];
        if (!v.empty()) {
            *s = v.back();
            v.pop_back();
            return hipSuccess;
        }
    }
    if (!low_priority) return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
    int lo = 0, hi = 0;
    e = hipDeviceGetStreamPriorityRange(&lo, &hi);
    if (e != hipSuccess) return e;
    return hipStreamCreateWithPriority(s, hipStreamNonBlocking, lo);
}

void pool_stream_put(bool low_priority, hipStream_t s) {
    if (!s) return;
    int dev = 0;
    if (pool().limit[0] == 0 || hipGetDevice(&dev) != hipSuccess) {
        (void)hipStreamDestroy(s);
        return;
    }
    std::lock_guard<std::mutex> g(pool().mu);
    pool().streams[low_priority][dev].push_back(s);
}

hipError_t pool_event_get(hipEvent_t *ev) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    {
        std::lock_guard<std::mutex> g(pool().mu);
        auto &v = pool().events[dev];
        if (!v.empty()) {
            *ev = v.back();
            v.pop_back();
            return hipSuccess;
        }
    }
    return hipEventCreateWithFlags(ev, hipEventDisableTiming);
}

void pool_event_put(hipEvent_t ev) {
    if (!ev) return;
    int dev = 0;
    if (pool().limit[0] == 0 || hipGetDevice(&dev) != hipSuccess) {
        (void)hipEventDestroy(ev);
        return;
    }
    std::lock_guard<std::mutex> g(pool().mu);
    pool().events[dev].push_back(ev);
}

size_t pool_trim() {
    std::vector<Block> take;
    std::vector<hipStream_t> ts;
    std::vector<hipEvent_t> te;
    {
        std::lock_guard<std::mutex> g(pool().mu);
        take.swap(pool().blocks);
        pool().cached[0] = pool().cached[1] = 0;
        for (auto &m : pool().streams) {
            for (auto &kv : m) ts.insert(ts.end(), kv.second.begin(), kv.second.end());
            m.clear();
        }
        for (auto &kv : pool().events) te.insert(te.end(), kv.second.begin(), kv.second.end());
        pool().events.clear();
    }
    for (hipStream_t q : ts) (void)hipStreamDestroy(q);
    for (hipEvent_t q : te) (void)hipEventDestroy(q);
    size_t n = 0;
    for (Block &b : take) {
        n += b.bytes;
        if (b.device < 0) release_pinned(b);
        else (void)hipFree(b.p); // hipFree takes any device's pointer
    }
    return n;
}

} // namespace ngsq
