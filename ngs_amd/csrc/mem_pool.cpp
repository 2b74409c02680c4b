// mem_pool.cpp -- see mem_pool.h
#include "mem_pool.h"

#include <hip/hip_runtime.h>

#include <cstdlib>
#include <mutex>
#include <vector>

namespace ngsq {
namespace {

constexpr size_t POOL_MIN = (size_t)1 << 20; // smaller blocks are not worth keeping

struct Block {
    void *p;
    size_t bytes;
    int device; // >= 0: memory of that device; < 0: pinned host memory filled for device -(device + 1) (its NUMA node)
};

struct Pool {
    std::mutex mu;
    std::vector<Block> blocks;
    size_t cached[2] = {0, 0}; // device | pinned
    size_t limit;
    Pool() {
        const char *e = getenv("NGSQ_POOL_MB");
        limit = (size_t)(e ? strtoull(e, nullptr, 10) : 12288ull) << 20;
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
        if (b.bytes < POOL_MIN || cached[b.device < 0] + b.bytes > limit) return false;
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
        if (pool().put(Block{p, got, dev})) return;
    }
    (void)hipFree(p);
}

hipError_t pool_pinned_alloc(void **p, size_t bytes, size_t *got) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    Block b{};
    if (bytes >= POOL_MIN && pool().take(-(dev + 1), bytes, &b)) {
        *p = b.p;
        *got = b.bytes;
        return hipSuccess;
    }
    hipError_t e = hipHostMalloc(p, bytes, hipHostMallocDefault);
    if (e != hipSuccess && pool_trim()) {
        (void)hipGetLastError();
        e = hipHostMalloc(p, bytes, hipHostMallocDefault);
    }
    *got = bytes;
    return e;
}

void pool_pinned_free(void *p, size_t got) {
    if (!p) return;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (got >= POOL_MIN && pool().put(Block{p, got, -(dev + 1)})) return;
    (void)hipHostFree(p);
}

size_t pool_trim() {
    std::vector<Block> take;
    {
        std::lock_guard<std::mutex> g(pool().mu);
        take.swap(pool().blocks);
        pool().cached[0] = pool().cached[1] = 0;
    }
    size_t n = 0;
    for (const Block &b : take) {
        n += b.bytes;
        if (b.device < 0) (void)hipHostFree(b.p);
        else (void)hipFree(b.p); // hipFree takes any device's pointer
    }
    return n;
}

} // namespace ngsq
