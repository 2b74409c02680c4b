// comm.h -- private definition of ngsq_comm (include/ngsq_comm.h): the transports of the
// multi-GPU exchange.  A transport moves bytes between the ranks of one job; what is moved and
// when is decided by exchange.cpp.
#pragma once

#include <hip/hip_runtime_api.h>

#include <string>

#include "../../include/ngsq_comm.h"

struct ngsq_comm {
    int rank = 0, world = 1;
    const char *kind = "";
    bool device = false; // true: buffers are device memory and operations are enqueued on a stream (RCCL)
    std::string err;
    virtual ~ngsq_comm() {}
    // in-place wrap-around sum; elem_bytes 4 or 8
    virtual int allreduce(void *buf, uint64_t count, uint32_t elem_bytes, hipStream_t s) = 0;
    virtual int allgather(const void *send, void *recv, uint64_t bytes, hipStream_t s) = 0;
    virtual int sendrecv(const ngsq_p2p *sends, uint32_t n_sends, const ngsq_p2p *recvs, uint32_t n_recvs,
                         hipStream_t s) = 0;
    // host-buffer collectives (device transports stage through their own scratch and stream)
    virtual int allgather_host(const void *send, void *recv, uint64_t bytes) { return allgather(send, recv, bytes, nullptr); }
    virtual int allreduce_host(void *buf, uint64_t count, uint32_t eb) { return allreduce(buf, count, eb, nullptr); }
    virtual int sendrecv_host(const ngsq_p2p *s, uint32_t ns, const ngsq_p2p *r, uint32_t nr) {
        return sendrecv(s, ns, r, nr, nullptr);
    }
};

namespace ngsq {
int comm_fail(ngsq_comm *c, int code, const char *fmt, ...);
}
