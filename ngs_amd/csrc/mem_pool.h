// mem_pool.h -- big blocks of device and pinned host memory that a finished device ingest gives back are kept for the
// next file scanned by the process instead of going back to the driver: allocating and pinning the pipeline's buffers
// costs ~130 ms per file and freeing them ~100 ms (GiB-sized hipMalloc / hipHostMalloc / hipFree are page-table work),
// against ~300 ms of actual scanning for a 6 GB BAM.  Released by ngsq_release_cached_memory() and whenever a context
// is destroyed; NGSQ_POOL_MB=0 turns the cache off, NGSQ_POOL_MB=<n> caps what it keeps per kind (default 12288).
#pragma once

#include <hip/hip_runtime_api.h>
#include <stddef.h>

namespace ngsq {

// *got = bytes of the block (>= bytes); the same value goes back to pool_*_free.  Device blocks belong to the
// current device (hipSetDevice) of the calling thread; pinned blocks are kept per device too (they were allocated by a
// thread on that device's NUMA node).
hipError_t pool_device_alloc(void **p, size_t bytes, size_t *got);
void pool_device_free(void *p, size_t got);
hipError_t pool_pinned_alloc(void **p, size_t bytes, size_t *got);
void pool_pinned_free(void *p, size_t got);
// give everything that is cached back to the driver; returns the bytes released
size_t pool_trim();

} // namespace ngsq
