// mem_pool.h -- big blocks of device and pinned host memory that a finished device ingest gives back are kept for the
// next file scanned by the process instead of going back to the driver: allocating and pinning the pipeline's buffers
// costs ~130 ms per file and freeing them ~100 ms (GiB-sized hipMalloc / hipHostMalloc / hipFree are page-table work),
// against ~300 ms of actual scanning for a 6 GB BAM.  Released by ngsq_release_cached_memory() and whenever a context
// is destroyed; NGSQ_POOL_MB=0 turns the cache off, NGSQ_POOL_MB=<n> caps what it keeps per kind (default: 6144 of device memory, 1536 pinned).
#pragma once

#include <hip/hip_runtime_api.h>
#include <stddef.h>

namespace ngsq {

// *got = bytes of the block (>= bytes); the same value goes back to pool_*_free.  Device blocks belong to the
// current device (hipSetDevice) of the calling thread; pinned blocks are kept per device too (they were allocated by a
// thread on that device's NUMA node).
hipError_t pool_device_alloc(void **p, size_t bytes, size_t *got);
void pool_device_free(void *p, size_t got);
// Pinned host blocks are anonymous mappings that are registered with HIP (hipHostRegister) 32 MiB at a time, when a
// piece is first copied FROM: hipHostMalloc pins at 0.19 ms per MiB up front -- 90 ms for the two 256 MiB buffers of a
// 6 GB file's pipeline before its first byte is read -- while registering a piece that has just been written (its pages
// are there) costs 0.06 ms per MiB, and only for the pieces in use.  A cached block keeps its registrations.
hipError_t pool_pinned_alloc(void **p, size_t bytes, size_t *got);
void pool_pinned_free(void *p, size_t got);
// hipMemcpyAsync(dst, block + off, len, host to device, s), registering the pieces of the block it reads that are not
// registered yet; `block` is what pool_pinned_alloc returned.  The copy is issued piece by piece (a copy may not span two
// registrations).
hipError_t pool_pinned_h2d(void *dst, const void *block, size_t off, size_t len, hipStream_t s);
// The streams and events of a finished device ingest are kept for the next one as well: creating three streams and fifteen
// events costs a scan 6-7 ms before its first byte is read (measured, round 4: 3 % of a 6 GB file's scan).  Per device (the
// calling thread's current one); low_priority: created with the device's lowest stream priority.  A stream goes back
// idle (the caller has synchronised it).  With the cache off (NGSQ_POOL_MB=0) put destroys.
hipError_t pool_stream_get(bool low_priority, hipStream_t *s);
void pool_stream_put(bool low_priority, hipStream_t s);
hipError_t pool_event_get(hipEvent_t *e); // hipEventDisableTiming
void pool_event_put(hipEvent_t e);
// give everything that is cached back to the driver; returns the bytes released
size_t pool_trim();

} // namespace ngsq
