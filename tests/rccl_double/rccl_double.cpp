// rccl_double.cpp -- a TEST DOUBLE for librccl (tests only; never shipped, never measured).
//
// RCCL refuses two ranks on one device and no box of this pool has two GPUs, so the library's RCCL transport
// (ngs_amd/csrc/comm.cpp RcclComm: collectives on device buffers on the context's stream, point-to-point halos as one
// ncclGroup of ncclSend/ncclRecv) would otherwise only ever run with ONE rank, where nothing is exchanged.  This double
// implements the ten entry points the library binds, with RCCL's calling conventions, over POSIX shared memory and
// hipMemcpy, so that tests/test_parity_gpu.py can run the exchange with three ranks sharing the box's GPU through exactly
// the code path a multi-GPU node takes (NGSQ_RCCL_LIB points the library at it).  What it checks is OUR use of the API --
// buffers, counts, data types, peers, group pairing, stream ordering -- not RCCL.
//
// Semantics kept: collectives are ordered after the work queued on `stream` (the double synchronises it); all ranks of
// a communicator call collectives in the same order; sends and receives between a pair match in order and size; the
// operations of a group progress concurrently (a rank may both send to and receive from a peer in one group).
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sched.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <cstdio>
#include <cstring>
#include <random>
#include <string>
#include <algorithm>
#include <vector>

namespace {

constexpr uint64_t SLOT = 1 << 20;   // bytes per rank of a collective round
constexpr uint64_t MBOX = 1 << 18;   // bytes per point-to-point chunk
constexpr int MAX_WORLD = 8;

struct Mailbox {
    std::atomic<uint64_t> written, read; // chunks
    uint64_t len;
    uint8_t data[MBOX];
};
struct Shared {
    std::atomic<uint32_t> arrived, generation, attached;
    uint32_t world;
    uint8_t slot[MAX_WORLD][SLOT];
    Mailbox box[MAX_WORLD][MAX_WORLD]; // [from][to]
};

double now_s() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + ts.tv_nsec * 1e-9;
}

struct P2p {
    bool send;
    uint8_t *buf;
    uint64_t bytes, done;
    int peer;
    hipStream_t stream;
};

thread_local int g_group_depth = 0;
thread_local std::vector<std::pair<struct ncclComm *, P2p>> g_group;

} // namespace

struct ncclComm {
    int rank = 0, world = 0;
    Shared *sh = nullptr;
    std::vector<uint8_t> host;

    bool barrier() {
        const uint32_t gen = sh->generation.load(std::memory_order_acquire);
        if (sh->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)world) {
            sh->arrived.store(0, std::memory_order_relaxed);
            sh->generation.fetch_add(1, std::memory_order_acq_rel);
            return true;
        }
        const double t0 = now_s();
        for (uint64_t spins = 0; sh->generation.load(std::memory_order_acquire) == gen; spins++) {
            if (spins > 1000) sched_yield();
            if ((spins & 0xFFFF) == 0 && now_s() - t0 > 300.0) return false;
        }
        return true;
    }
};

static size_t type_size(ncclDataType_t t) {
    switch (t) {
    case ncclUint8: case ncclInt8: return 1;
    case ncclUint32: case ncclInt32: return 4;
    case ncclUint64: case ncclInt64: return 8;
    default: return 0;
    }
}

// progress every queued send / receive until all are complete
static ncclResult_t run_p2p(std::vector<std::pair<ncclComm *, P2p>> &ops) {
    for (auto &o : ops)
        if (hipStreamSynchronize(o.second.stream) != hipSuccess) return ncclUnhandledCudaError;
    const double t0 = now_s();
    for (uint64_t spins = 0;; spins++) {
        bool all = true, moved = false;
        std::vector<Mailbox *> busy; // messages between one pair of ranks match in list order: one unfinished operation per mailbox at a time
        for (auto &e : ops) {
            ncclComm *c = e.first;
            P2p &o = e.second;
            if (o.done == o.bytes) continue;
            all = false;
            Mailbox &m = o.send ? c->sh->box[c->rank][o.peer] : c->sh->box[o.peer][c->rank];
            if (std::find(busy.begin(), busy.end(), &m) != busy.end()) continue;
            busy.push_back(&m);
            const uint64_t w = m.written.load(std::memory_order_acquire), r = m.read.load(std::memory_order_acquire);
            if (o.send && w == r) {
                const uint64_t n = std::min<uint64_t>(MBOX, o.bytes - o.done);
                if (hipMemcpy(m.data, o.buf + o.done, n, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
                m.len = n;
                m.written.store(w + 1, std::memory_order_release);
                o.done += n;
                moved = true;
            } else if (!o.send && w > r) {
                const uint64_t n = m.len;
                if (n > o.bytes - o.done) return ncclInvalidUsage; // a send larger than the matching receive
                if (hipMemcpy(o.buf + o.done, m.data, n, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
                m.read.store(r + 1, std::memory_order_release);
                o.done += n;
                moved = true;
            }
        }
        if (all) return ncclSuccess;
        if (!moved) {
            if (spins > 1000) sched_yield();
            if ((spins & 0xFFFF) == 0 && now_s() - t0 > 300.0) return ncclSystemError;
        }
    }
}

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
    memset(id, 0, sizeof *id);
    std::random_device rd;
    snprintf(id->internal, sizeof id->internal, "/rccl-double-%08x%08x", (unsigned)rd(), (unsigned)getpid());
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *out, int nranks, ncclUniqueId id, int rank) {
    if (!out || nranks < 1 || nranks > MAX_WORLD || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    if (const char *st = getenv("RCCL_DOUBLE_STALL_S")) // a bootstrap that never completes (the library's time limit is what is tested)
        usleep((useconds_t)(atof(st) * 1e6));
    char name[NCCL_UNIQUE_ID_BYTES + 1];
    memcpy(name, id.internal, NCCL_UNIQUE_ID_BYTES);
    name[NCCL_UNIQUE_ID_BYTES] = 0;
    int fd = -1;
    const double t0 = now_s();
    if (rank == 0) {
        fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, sizeof(Shared)) != 0) return ncclSystemError;
    } else {
        while ((fd = shm_open(name, O_RDWR, 0600)) < 0) {
            if (now_s() - t0 > 300.0) return ncclSystemError;
            usleep(1000);
        }
        for (;;) { // wait for rank 0's ftruncate
            const off_t sz = lseek(fd, 0, SEEK_END);
            if (sz >= (off_t)sizeof(Shared)) break;
            if (now_s() - t0 > 300.0) return ncclSystemError;
            usleep(1000);
        }
    }
    void *p = mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return ncclSystemError;
    ncclComm *c = new ncclComm();
    c->rank = rank;
    c->world = nranks;
    c->sh = static_cast<Shared *>(p);
    if (rank == 0) c->sh->world = (uint32_t)nranks;
    c->sh->attached.fetch_add(1, std::memory_order_acq_rel);
    while (c->sh->attached.load(std::memory_order_acquire) < (uint32_t)nranks) {
        if (now_s() - t0 > 300.0) return ncclSystemError;
        usleep(200);
    }
    if (rank == 0) shm_unlink(name); // everyone has it mapped
    *out = c;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) {
    if (!c) return ncclSuccess;
    munmap(c->sh, sizeof(Shared));
    delete c;
    return ncclSuccess;
}

const char *ncclGetErrorString(ncclResult_t r) {
    switch (r) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "unhandled cuda error (rccl double)";
    case ncclSystemError: return "unhandled system error (rccl double: timeout or shared memory)";
    case ncclInvalidArgument: return "invalid argument (rccl double)";
    case ncclInvalidUsage: return "invalid usage (rccl double)";
    default: return "error (rccl double)";
    }
}

ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t c, hipStream_t s) {
    const size_t eb = type_size(dt);
    if (!c || op != ncclSum || (eb != 4 && eb != 8)) return ncclInvalidArgument;
    if (hipStreamSynchronize(s) != hipSuccess) return ncclUnhandledCudaError;
    const uint64_t bytes = count * eb;
    std::vector<uint8_t> acc;
    for (uint64_t o = 0; o < bytes || (o == 0 && bytes == 0); o += SLOT) {
        const uint64_t n = std::min<uint64_t>(SLOT, bytes - o);
        if (n && hipMemcpy(c->sh->slot[c->rank], static_cast<const uint8_t *>(send) + o, n, hipMemcpyDeviceToHost) != hipSuccess)
            return ncclUnhandledCudaError;
        if (!c->barrier()) return ncclSystemError;
        acc.assign(n, 0);
        for (int r = 0; r < c->world; r++) {
            if (eb == 8) {
                uint64_t *d = reinterpret_cast<uint64_t *>(acc.data());
                const uint64_t *x = reinterpret_cast<const uint64_t *>(c->sh->slot[r]);
                for (uint64_t i = 0; i < n / 8; i++) d[i] += x[i];
            } else {
                uint32_t *d = reinterpret_cast<uint32_t *>(acc.data());
                const uint32_t *x = reinterpret_cast<const uint32_t *>(c->sh->slot[r]);
                for (uint64_t i = 0; i < n / 4; i++) d[i] += x[i];
            }
        }
        if (!c->barrier()) return ncclSystemError;
        if (n && hipMemcpy(static_cast<uint8_t *>(recv) + o, acc.data(), n, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
        if (bytes == 0) break;
    }
    return ncclSuccess;
}

ncclResult_t ncclAllGather(const void *send, void *recv, size_t sendcount, ncclDataType_t dt, ncclComm_t c, hipStream_t s) {
    const size_t eb = type_size(dt);
    if (!c || !eb) return ncclInvalidArgument;
    if (hipStreamSynchronize(s) != hipSuccess) return ncclUnhandledCudaError;
    const uint64_t bytes = sendcount * eb;
    for (uint64_t o = 0; o < bytes; o += SLOT) {
        const uint64_t n = std::min<uint64_t>(SLOT, bytes - o);
        if (hipMemcpy(c->sh->slot[c->rank], static_cast<const uint8_t *>(send) + o, n, hipMemcpyDeviceToHost) != hipSuccess)
            return ncclUnhandledCudaError;
        if (!c->barrier()) return ncclSystemError;
        for (int r = 0; r < c->world; r++)
            if (hipMemcpy(static_cast<uint8_t *>(recv) + (uint64_t)r * bytes + o, c->sh->slot[r], n, hipMemcpyHostToDevice) != hipSuccess)
                return ncclUnhandledCudaError;
        if (!c->barrier()) return ncclSystemError;
    }
    return ncclSuccess;
}

ncclResult_t ncclGroupStart() {
    g_group_depth++;
    return ncclSuccess;
}

ncclResult_t ncclGroupEnd() {
    if (g_group_depth <= 0) return ncclInvalidUsage;
    if (--g_group_depth) return ncclSuccess;
    std::vector<std::pair<ncclComm *, P2p>> ops;
    ops.swap(g_group);
    return run_p2p(ops);
}

static ncclResult_t p2p(bool send, void *buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t c, hipStream_t s) {
    const size_t eb = type_size(dt);
    if (!c || !eb || peer < 0 || peer >= c->world) return ncclInvalidArgument;
    if (peer == c->rank) return ncclInvalidUsage; // (RCCL allows it; the library never does it)
    g_group.push_back({c, P2p{send, static_cast<uint8_t *>(buf), count * eb, 0, peer, s}});
    if (g_group_depth) return ncclSuccess;
    std::vector<std::pair<ncclComm *, P2p>> ops;
    ops.swap(g_group);
    return run_p2p(ops);
}
ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t c, hipStream_t s) {
    return p2p(true, const_cast<void *>(buf), count, dt, peer, c, s);
}
ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t c, hipStream_t s) {
    return p2p(false, buf, count, dt, peer, c, s);
}

} // extern "C"
