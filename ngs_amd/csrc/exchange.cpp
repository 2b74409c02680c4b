// exchange.cpp -- the one exchange of a sharded `ngs qc` scan (include/ngsq_comm.h):
// counters all-reduce, owner-computes Coverage teardown with point-to-point halos, all-reduce of the
// partial teardown results.  The protocol is written once, against
//   * a shard state (ngsq_shard_state): the context's device blocks + four small operations, or
//     host arrays supplied by the caller, and
//   * a transport (comm.h): RCCL on the state's stream, or a host transport (device state is then
//     staged through host memory).
// Every rank issues the same collectives in the same order; everything that decides the order
// (the plan) is a pure function of all-gathered values.
//
// Reference counterpart: none (the reference is single-threaded); what is split is pass 2's
// per-sequence teardown, src/qc/sequence_based/coverage.rs:182-262, and what is summed are the
// facets' integer states (general.rs:31-124, template_length.rs:79-87, gc_content.rs:38-100,
// quality_scores.rs:37-49, coverage.rs:148-180, edits.rs:217-303).
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "bam_reader.h"
#include "comm.h"
#include "context.h"

using ngsq::comm_fail;

namespace {

constexpr uint64_t CH = ngsq::COV_CHUNK;
constexpr uint64_t STAGE_PIECE = 64ull << 20;

struct Plan {
    std::vector<uint64_t> own;    // [world][2]
    std::vector<uint32_t> order;  // owners by position
    struct X {
        uint32_t src, dst;
        uint64_t c0, c1;
    };
    std::vector<X> xfer;
};

Plan make_plan(const uint64_t *ranges, uint32_t world, uint64_t n_chunks) {
    Plan p;
    p.own.assign(2 * (size_t)world, 0);
    for (uint32_t r = 0; r < world; r++)
        if (ranges[2 * r + 1] > ranges[2 * r]) p.order.push_back(r);
    std::stable_sort(p.order.begin(), p.order.end(), [&](uint32_t a, uint32_t b) { return ranges[2 * a] < ranges[2 * b]; });
    for (size_t k = 0; k < p.order.size(); k++) {
        const uint32_t r = p.order[k];
        const uint64_t b0 = k == 0 ? 0 : ranges[2 * r];
        const uint64_t b1 = k + 1 == p.order.size() ? n_chunks : ranges[2 * p.order[k + 1]];
        p.own[2 * r] = b0;
        p.own[2 * r + 1] = std::max(b0, b1);
    }
    for (uint32_t s = 0; s < world; s++) {
        const uint64_t lo = ranges[2 * s], hi = ranges[2 * s + 1];
        if (hi <= lo) continue;
        for (uint32_t d : p.order) {
            if (d == s) continue;
            const uint64_t c0 = std::max(lo, p.own[2 * d]), c1 = std::min(hi, p.own[2 * d + 1]);
            if (c1 > c0) p.xfer.push_back({s, d, c0, c1});
        }
    }
    return p;
}

// scratch of one exchange: a block in state memory + host staging
struct Scratch {
    bool device = false;
    uint8_t *state = nullptr;
    uint64_t state_cap = 0;
    std::vector<uint8_t> host;
    ~Scratch() { release(); }
    void release() {
        if (state) {
            if (device) (void)hipFree(state);
            else free(state);
        }
        state = nullptr;
        state_cap = 0;
    }
};

struct Xchg {
    const ngsq_shard_state &S;
    ngsq_comm &T;
    Scratch &scr;
    ngsq_exchange_report rep{};
    const bool dev, staged;
    hipStream_t st;

    Xchg(const ngsq_shard_state &s, ngsq_comm &t, Scratch &sc)
        : S(s), T(t), scr(sc), dev(s.memory == NGSQ_MEM_DEVICE), staged(s.memory == NGSQ_MEM_DEVICE && !t.device),
          st((hipStream_t)s.stream) {}

    int fail(int code, const char *msg) { return comm_fail(&T, code, "%s", msg); }
    int hip(hipError_t e, const char *what) {
        return e == hipSuccess ? NGSQ_OK : comm_fail(&T, NGSQ_ERR_DEVICE, "%s failed: %s", what, hipGetErrorString(e));
    }
    int sync() {
        rep.host_syncs += dev ? 1 : 0;
        const int rc = S.synchronize(S.user);
        return rc ? comm_fail(&T, rc, "shard state: synchronize failed (%d)", rc) : NGSQ_OK;
    }
    int reserve(uint64_t bytes) {
        if (scr.state && scr.device != dev) scr.release();
        if (scr.state_cap >= bytes) return NGSQ_OK;
        if (scr.state) { // buffers of the previous exchange may still be read by queued work
            int rc = sync();
            if (rc) return rc;
        }
        scr.release();
        const uint64_t cap = std::max<uint64_t>(bytes + bytes / 4, 1 << 16);
        scr.device = dev;
        if (dev) {
            int rc = hip(hipMalloc((void **)&scr.state, cap), "hipMalloc (exchange scratch)");
            if (rc) return rc;
        } else {
            scr.state = static_cast<uint8_t *>(malloc(cap));
            if (!scr.state) return fail(NGSQ_ERR_DEVICE, "out of memory (exchange scratch)");
        }
        scr.state_cap = cap;
        return NGSQ_OK;
    }
    int to_host(void *dst, const void *src, uint64_t bytes) { // no sync
        if (!bytes) return NGSQ_OK;
        if (!dev) {
            memcpy(dst, src, bytes);
            return NGSQ_OK;
        }
        return hip(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, st), "hipMemcpyAsync (D2H)");
    }
    int to_state(void *dst, const void *src, uint64_t bytes) {
        if (!bytes) return NGSQ_OK;
        if (!dev) {
            memcpy(dst, src, bytes);
            return NGSQ_OK;
        }
        return hip(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st), "hipMemcpyAsync (H2D)");
    }
    // in-place sum of a block in state memory
    int allreduce(void *buf, uint64_t count, uint32_t eb) {
        if (!count) return NGSQ_OK;
        if (!staged) return T.allreduce(buf, count, eb, st);
        uint8_t *p = static_cast<uint8_t *>(buf);
        const uint64_t bytes = count * eb;
        for (uint64_t o = 0; o < bytes; o += STAGE_PIECE) {
            const uint64_t n = std::min(STAGE_PIECE, bytes - o);
            if (scr.host.size() < n) scr.host.resize(n);
            int rc = to_host(scr.host.data(), p + o, n);
            if (!rc) rc = sync();
            if (!rc) rc = T.allreduce(scr.host.data(), n / eb, eb, nullptr);
            if (!rc) rc = to_state(p + o, scr.host.data(), n);
            if (!rc) rc = sync(); // the staging buffer is reused
            if (rc) return rc;
        }
        return NGSQ_OK;
    }
    // every rank's `bytes` at send (state memory) -> recv_state (state memory, may be null) and recv_host (may be null;
    // valid after the next sync())
    int allgather(const void *send, uint64_t bytes, void *recv_state, void *recv_host) {
        const uint64_t all = bytes * T.world;
        if (!staged && dev) {
            int rc = T.allgather(send, recv_state, bytes, st);
            if (!rc && recv_host) rc = to_host(recv_host, recv_state, all);
            return rc;
        }
        std::vector<uint8_t> mine(bytes), got(all);
        int rc = to_host(mine.data(), send, bytes);
        if (!rc && dev) rc = sync();
        if (!rc) rc = T.allgather(mine.data(), got.data(), bytes, nullptr);
        if (rc) return rc;
        if (recv_host) memcpy(recv_host, got.data(), all);
        if (recv_state) {
            if (!dev) memcpy(recv_state, got.data(), all);
            else {
                if (scr.host.size() < all) scr.host.resize(all);
                memcpy(scr.host.data(), got.data(), all);
                rc = to_state(recv_state, scr.host.data(), all);
                if (!rc) rc = sync();
            }
        }
        return rc;
    }
    int sendrecv(std::vector<ngsq_p2p> &sends, std::vector<ngsq_p2p> &recvs) {
        if (!staged) return T.sendrecv(sends.data(), (uint32_t)sends.size(), recvs.data(), (uint32_t)recvs.size(), st);
        uint64_t total = 0;
        for (auto &m : sends) total += m.bytes;
        for (auto &m : recvs) total += m.bytes;
        if (scr.host.size() < total) scr.host.resize(total);
        std::vector<ngsq_p2p> hs(sends), hr(recvs);
        uint64_t off = 0;
        int rc = NGSQ_OK;
        for (size_t i = 0; i < sends.size() && !rc; i++) {
            hs[i].buf = scr.host.data() + off;
            rc = to_host(hs[i].buf, sends[i].buf, sends[i].bytes);
            off += sends[i].bytes;
        }
        for (size_t i = 0; i < recvs.size(); i++) {
            hr[i].buf = scr.host.data() + off;
            off += recvs[i].bytes;
        }
        if (!rc) rc = sync();
        if (!rc) rc = T.sendrecv(hs.data(), (uint32_t)hs.size(), hr.data(), (uint32_t)hr.size(), nullptr);
        for (size_t i = 0; i < recvs.size() && !rc; i++) rc = to_state(recvs[i].buf, hr[i].buf, recvs[i].bytes);
        if (!rc) rc = sync();
        return rc;
    }

    int run() {
        const uint32_t world = (uint32_t)T.world, rank = (uint32_t)T.rank;
        if (!dev && T.device) return fail(NGSQ_ERR_INVALID_ARGUMENT, "a device transport (rccl) cannot move host state");
        if (!S.counters || !S.synchronize || !S.teardown_range)
            return fail(NGSQ_ERR_INVALID_ARGUMENT, "ngsq_shard_state: counters / synchronize / teardown_range missing");

        // ---- step 0: every rank must hold the SAME state layout -- the blocks are summed element by element, and a rank built
        // against another ABI (or configured with other sequences / facets / quality rows) would be summed misaligned, silently
        {
            const uint64_t mine[6] = {NGSQ_ABI_VERSION, S.n_counters, S.n_depth, S.n_edits, S.n_teardown, S.n_chunks};
            std::vector<uint64_t> all(6 * (size_t)world);
            int rc0 = ngsq_comm_allgather_host(&T, mine, all.data(), sizeof mine);
            if (rc0) return rc0;
            for (uint32_t r = 0; r < world; r++)
                if (memcmp(&all[6 * (size_t)r], mine, sizeof mine) != 0) {
                    static const char *const what[6] = {"ABI version", "counters", "depth", "edits", "teardown", "chunks"};
                    int f = 0;
                    while (f < 5 && all[6 * (size_t)r + f] == mine[f]) f++;
                    // (the same verdict on every rank: each compares with all the others)
                    return comm_fail(&T, NGSQ_ERR_STATE, "shard state layouts differ: %s is %llu on rank %u and %llu on rank %u", what[f],
                                     (unsigned long long)all[6 * (size_t)r + f], r, (unsigned long long)mine[f], rank);
                }
        }
        // ---- step 1: the record-facet state (and the Edits refs/alts) of all shards
        int rc = allreduce(S.counters, S.n_counters, 8);
        if (rc) return rc;
        if (S.edits && S.n_edits && (rc = allreduce(S.edits, S.n_edits, 4))) return rc;
        const bool coverage = S.depth && S.n_chunks;
        if (!coverage) {
            rep.mode = NGSQ_EXCHANGE_NONE;
            rc = S.teardown_range(S.user, 0, 0, nullptr, 0, rank, world);
            if (rc) return comm_fail(&T, rc, "shard state: teardown failed (%d)", rc);
            if (S.edits && S.n_edits) rc = allreduce(S.teardown, S.n_teardown, 8);
            return rc;
        }
        if (!S.touched || !S.halo_add || !S.summary)
            return fail(NGSQ_ERR_INVALID_ARGUMENT, "ngsq_shard_state: touched / halo_add / summary missing");

        // scratch in state memory: [ranges world*16 | my words 16 | all words world*8 (+pad) | halos]
        const uint64_t off_words = (uint64_t)world * 16, off_all = off_words + 16, off_halo = (off_all + (uint64_t)world * 8 + 255) & ~255ull;
        rc = reserve(off_halo);
        if (rc) return rc;

        // ---- step 2: who wrote where
        std::vector<uint64_t> touched(2 * (size_t)world), ranges(2 * (size_t)world);
        rc = allgather(S.touched, 16, scr.state, touched.data());
        if (!rc) rc = sync();
        if (rc) return rc;
        for (uint32_t r = 0; r < world; r++) {
            const uint64_t lo = touched[2 * r], hi = touched[2 * r + 1];
            if (lo == ~0ull || hi <= lo) {
                ranges[2 * r] = ranges[2 * r + 1] = 0;
            } else {
                ranges[2 * r] = std::min(lo / CH, S.n_chunks);
                ranges[2 * r + 1] = std::min((hi + CH - 1) / CH, S.n_chunks);
            }
        }
        const Plan plan = make_plan(ranges.data(), world, S.n_chunks);
        std::vector<uint64_t> out_bytes(world, 0);
        for (auto &x : plan.xfer) out_bytes[x.src] += (x.c1 - x.c0) * (CH + 1) * 4;
        const uint64_t max_out = *std::max_element(out_bytes.begin(), out_bytes.end());
        uint32_t *my_words = reinterpret_cast<uint32_t *>(scr.state + off_words);
        uint32_t *all_words = reinterpret_cast<uint32_t *>(scr.state + off_all);
        std::vector<uint32_t> h_words(2 * (size_t)world, 0);
        auto any_bad = [&]() {
            for (uint32_t r = 0; r < world; r++)
                if (h_words[2 * r + 1]) return true;
            return false;
        };
        const char *overlap = "sorted_input shards overlap: records of another shard reach into positions this shard already "
                              "finished (cov_head_guard too small, or the shards are not in coordinate order); re-run without "
                              "sorted_input";

        uint64_t halo_limit = NGSQ_HALO_LIMIT_BYTES;
        if (const char *e = getenv("NGSQ_HALO_LIMIT_BYTES")) halo_limit = strtoull(e, nullptr, 0); // tests force the fallback
        if (max_out > halo_limit) {
            // ---- unsorted shards: the written ranges overlap -- sum the whole block, every rank scans all of it
            if (S.chunk_flags) { // nothing may have been finished while streaming, on any rank
                const uint64_t whole[2] = {0, S.n_chunks};
                rc = S.summary(S.user, 0, 0, whole, 1, my_words);
                if (rc) return comm_fail(&T, rc, "shard state: summary failed (%d)", rc);
                rc = allgather(my_words, 8, all_words, h_words.data());
                if (!rc) rc = sync();
                if (rc) return rc;
                if (any_bad()) return fail(NGSQ_ERR_UNSORTED, overlap);
            }
            rc = allreduce(S.depth, S.n_depth, 4);
            if (rc) return rc;
            rep.mode = NGSQ_EXCHANGE_ALLREDUCE;
            rep.owned_chunk_lo = 0;
            rep.owned_chunk_hi = S.n_chunks;
            rc = S.teardown_range(S.user, 0, S.n_chunks, nullptr, 0, 0, 1);
            return rc ? comm_fail(&T, rc, "shard state: teardown failed (%d)", rc) : NGSQ_OK;
        }

        // ---- step 3: halos to their owners, point to point
        uint64_t in_bytes = 0;
        for (auto &x : plan.xfer)
            if (x.dst == rank) in_bytes += (x.c1 - x.c0) * (CH + 1) * 4;
        rc = reserve(off_halo + in_bytes);
        if (rc) return rc;
        my_words = reinterpret_cast<uint32_t *>(scr.state + off_words); // reserve() may have moved the block
        all_words = reinterpret_cast<uint32_t *>(scr.state + off_all);
        std::vector<ngsq_p2p> sends, recvs;
        std::vector<uint64_t> in_ranges, out_ranges;
        {
            uint64_t off = off_halo;
            for (auto &x : plan.xfer) {
                const uint64_t nc = x.c1 - x.c0;
                if (x.src == rank) {
                    sends.push_back({(int32_t)x.dst, 0, S.depth + x.c0 * CH, nc * CH * 4});
                    sends.push_back({(int32_t)x.dst, 0, S.depth + S.n_diff + x.c0, nc * 4});
                    rep.halo_bytes_sent += nc * (CH + 1) * 4;
                    out_ranges.push_back(x.c0);
                    out_ranges.push_back(x.c1);
                }
                if (x.dst == rank) {
                    recvs.push_back({(int32_t)x.src, 0, scr.state + off, nc * CH * 4});
                    recvs.push_back({(int32_t)x.src, 0, scr.state + off + nc * CH * 4, nc * 4});
                    off += nc * (CH + 1) * 4;
                    in_ranges.push_back(x.c0);
                    in_ranges.push_back(x.c1);
                    rep.halo_bytes_received += nc * (CH + 1) * 4;
                }
            }
        }
        if (!plan.xfer.empty()) { // the same decision on every rank
            rc = sendrecv(sends, recvs);
            if (rc) return rc;
        }
        for (size_t i = 0; i + 1 < recvs.size(); i += 2) {
            const uint64_t c0 = in_ranges[i], c1 = in_ranges[i + 1];
            rc = S.halo_add(S.user, c0, c1, static_cast<const uint32_t *>(recvs[i].buf), static_cast<const uint32_t *>(recvs[i + 1].buf));
            if (rc) return comm_fail(&T, rc, "shard state: halo_add failed (%d)", rc);
        }

        // ---- step 4: one word per rank: what its owned range sums to now (+ the verdict on the incoming ranges)
        const uint64_t b0 = plan.own[2 * rank], b1 = plan.own[2 * rank + 1];
        // The verdict: no chunk this shard finished while streaming may lie in a range it RECEIVES entries for (the halo of
        // the shard in front would come too late) -- nor in a range it SENDS (the chunk belongs to a shard that starts in
        // front of positions this one has already tallied: shards out of coordinate order, each position counted twice)
        std::vector<uint64_t> check(in_ranges);
        check.insert(check.end(), out_ranges.begin(), out_ranges.end());
        rc = S.summary(S.user, b0, b1, check.data(), (uint32_t)(check.size() / 2), my_words);
        if (rc) return comm_fail(&T, rc, "shard state: summary failed (%d)", rc);
        rc = allgather(my_words, 8, all_words, h_words.data());
        if (rc) return rc;
        uint64_t front = 0;
        for (uint32_t r : plan.order) {
            if (r == rank) break;
            front |= 1ull << r;
        }

        // ---- step 5: tear down the owned chunks, sum the partial results
        rep.mode = NGSQ_EXCHANGE_OWNER;
        rep.owned_chunk_lo = b0;
        rep.owned_chunk_hi = b1;
        rc = S.teardown_range(S.user, b0, b1, all_words, front, rank, world);
        if (rc) return comm_fail(&T, rc, "shard state: teardown failed (%d)", rc);
        rc = allreduce(S.teardown, S.n_teardown, 8);
        if (!rc) rc = sync(); // h_words has landed
        if (rc) return rc;
        if (any_bad()) return fail(NGSQ_ERR_UNSORTED, overlap);
        return NGSQ_OK;
    }
};

int run_exchange(const ngsq_shard_state *s, ngsq_comm *comm, Scratch &scr, ngsq_exchange_report *report) {
    if (!s || !comm) return comm_fail(comm, NGSQ_ERR_INVALID_ARGUMENT, "null argument");
    if (s->struct_size != sizeof(ngsq_shard_state))
        return comm_fail(comm, NGSQ_ERR_INVALID_ARGUMENT, "ngsq_shard_state.struct_size %u != %zu", s->struct_size, sizeof(ngsq_shard_state));
    if (report && report->struct_size != sizeof(ngsq_exchange_report))
        return comm_fail(comm, NGSQ_ERR_INVALID_ARGUMENT, "ngsq_exchange_report.struct_size %u != %zu", report->struct_size,
                         sizeof(ngsq_exchange_report));
    if (s->memory != NGSQ_MEM_HOST && s->memory != NGSQ_MEM_DEVICE)
        return comm_fail(comm, NGSQ_ERR_INVALID_ARGUMENT, "ngsq_shard_state.memory %u", s->memory);
    Xchg x(*s, *comm, scr);
    const int rc = x.run();
    if (report) {
        x.rep.struct_size = sizeof(ngsq_exchange_report);
        *report = x.rep;
    }
    return rc;
}

} // namespace

extern "C" {

int64_t ngsq_exchange_plan(const uint64_t *ranges, uint32_t world, uint64_t n_chunks, uint64_t *own, uint32_t *order,
                           uint32_t *n_owners, uint64_t *xfer, uint64_t xfer_cap) {
    if (!ranges || !own || !order || !n_owners || (xfer_cap && !xfer) || !world)
        return comm_fail(nullptr, NGSQ_ERR_INVALID_ARGUMENT, "null argument");
    for (uint32_t r = 0; r < world; r++)
        if (ranges[2 * r] < ranges[2 * r + 1] && ranges[2 * r + 1] > n_chunks)
            return comm_fail(nullptr, NGSQ_ERR_INVALID_ARGUMENT, "range of rank %u outside [0, n_chunks]", r);
    const Plan p = make_plan(ranges, world, n_chunks);
    memcpy(own, p.own.data(), p.own.size() * 8);
    *n_owners = (uint32_t)p.order.size();
    for (size_t k = 0; k < p.order.size(); k++) order[k] = p.order[k];
    if (p.xfer.size() > xfer_cap) return comm_fail(nullptr, NGSQ_ERR_BUFFER_TOO_SMALL, "%zu transfers", p.xfer.size());
    for (size_t k = 0; k < p.xfer.size(); k++) {
        xfer[4 * k] = p.xfer[k].src;
        xfer[4 * k + 1] = p.xfer[k].dst;
        xfer[4 * k + 2] = p.xfer[k].c0;
        xfer[4 * k + 3] = p.xfer[k].c1;
    }
    return (int64_t)p.xfer.size();
}

int ngsq_exchange_state(const ngsq_shard_state *state, ngsq_comm *comm, ngsq_exchange_report *report) {
    Scratch scr;
    const int rc = run_exchange(state, comm, scr, report);
    if (state && state->memory == NGSQ_MEM_DEVICE && state->synchronize) (void)state->synchronize(state->user); // scratch is freed
    return rc;
}

} // extern "C"

// ---------------------------------------------------------------------------------------------
// the context's device state behind ngsq_shard_state
// ---------------------------------------------------------------------------------------------
namespace ngsq {
hipError_t launch_halo_add(uint32_t *dst, const uint32_t *src, uint64_t n, hipStream_t s);
hipError_t launch_range_summary(const uint32_t *chunk_sums, uint64_t b0, uint64_t b1, const uint8_t *flags, const uint64_t *in_ranges,
                                uint32_t n_in, uint32_t *out2, hipStream_t s);
} // namespace ngsq

namespace {

int ctx_sync(void *u) { return ngsq_synchronize(static_cast<ngsq_ctx *>(u)); }

int ctx_halo_add(void *u, uint64_t c0, uint64_t c1, const uint32_t *diff, const uint32_t *sums) {
    ngsq_ctx *c = static_cast<ngsq_ctx *>(u);
    if (c1 <= c0) return NGSQ_OK;
    if (c1 > c->n_chunks) return NGSQ_ERR_INVALID_ARGUMENT;
    if (ngsq::launch_halo_add(c->st.depth + c0 * CH, diff, (c1 - c0) * CH, c->stream) != hipSuccess) return NGSQ_ERR_DEVICE;
    if (ngsq::launch_halo_add(c->st.chunk_sums + c0, sums, c1 - c0, c->stream) != hipSuccess) return NGSQ_ERR_DEVICE;
    return NGSQ_OK;
}

int ctx_summary(void *u, uint64_t b0, uint64_t b1, const uint64_t *in_ranges, uint32_t n_in, uint32_t *out2) {
    ngsq_ctx *c = static_cast<ngsq_ctx *>(u);
    if (b1 > c->n_chunks || b0 > b1) return NGSQ_ERR_INVALID_ARGUMENT;
    const hipError_t e = ngsq::launch_range_summary(c->st.chunk_sums, b0, b1, c->stream_cov ? c->d_chunk_flags : nullptr, in_ranges, n_in, out2,
                                                    c->stream);
    return e == hipSuccess ? NGSQ_OK : NGSQ_ERR_DEVICE;
}

int ctx_teardown_range(void *u, uint64_t b0, uint64_t b1, const uint32_t *words, uint64_t front_mask, uint32_t part, uint32_t parts) {
    ngsq_ctx *c = static_cast<ngsq_ctx *>(u);
    if (c->n_chunks) {
        int rc = ngsq_set_scan_range(c, b0, b1, 0);
        if (rc) return rc;
        c->scan_words = front_mask ? words : nullptr;
        c->scan_front = front_mask;
        if (front_mask) c->scan_partial = true;
    }
    c->vaf_part = part;
    c->vaf_parts = parts ? parts : 1;
    return ngsq_teardown(c);
}

} // namespace

extern "C" int ngsq_exchange(ngsq_ctx *c, ngsq_comm *comm, ngsq_exchange_report *report) {
    if (!c || !comm) return comm_fail(comm, NGSQ_ERR_INVALID_ARGUMENT, "null argument");
    if (c->finalized || c->torn_down) return comm_fail(comm, NGSQ_ERR_STATE, "ngsq_exchange after the teardown; call ngsq_reset");
    if (!c->ft_deferred.empty())
        return comm_fail(comm, NGSQ_ERR_STATE, "NGSQ_FACET_FEATURES: batches were scanned but ngsq_set_features was never called");
    if (hipSetDevice(c->device) != hipSuccess) return comm_fail(comm, NGSQ_ERR_DEVICE, "hipSetDevice(%d) failed", c->device);
    {   // the shards' quality tables have grown to the longest read each of them met: the counter blocks are summed element
        // by element, so they take the size of the largest first
        std::vector<uint32_t> rows((size_t)comm->world, 0);
        const uint32_t mine = c->st.max_read_len;
        int rc = ngsq_comm_allgather_host(comm, &mine, rows.data(), sizeof mine);
        if (rc) return rc;
        rc = ngsq::grow_quality_table(c, *std::max_element(rows.begin(), rows.end()));
        if (rc) return comm_fail(comm, rc, "%s", c->err.c_str());
    }
    ngsq_shard_state s{};
    s.struct_size = sizeof s;
    s.memory = NGSQ_MEM_DEVICE;
    s.user = c;
    s.stream = c->stream;
    s.counters = reinterpret_cast<uint64_t *>(c->st.counters);
    s.n_counters = c->n_counters;
    s.depth = c->st.depth;
    s.n_depth = c->n_depth;
    s.n_diff = c->n_diff;
    s.n_chunks = c->n_chunks;
    s.teardown = reinterpret_cast<uint64_t *>(c->d_td);
    s.n_teardown = c->n_td;
    s.edits = c->st.edits;
    s.n_edits = c->n_edits;
    s.chunk_flags = c->stream_cov ? c->d_chunk_flags : nullptr;
    s.touched = reinterpret_cast<const uint64_t *>(c->d_touched);
    s.synchronize = ctx_sync;
    s.halo_add = ctx_halo_add;
    s.summary = ctx_summary;
    s.teardown_range = ctx_teardown_range;
    if (!c->xchg_scratch) c->xchg_scratch = new Scratch();
    const int rc = run_exchange(&s, comm, *static_cast<Scratch *>(c->xchg_scratch), report);
    if (rc) c->err = comm->err;
    return rc;
}

namespace ngsq {
void free_exchange_scratch(void *p) { delete static_cast<Scratch *>(p); }
} // namespace ngsq

// ---------------------------------------------------------------------------------------------
// one BAM file, several GPUs: the shards scan on an assumption about their first record and compare notes afterwards
// ---------------------------------------------------------------------------------------------
extern "C" int ngsq_bam_shard_open(ngsq_bam *bam, ngsq_ctx *ctx, ngsq_comm *comm) {
    if (!bam || !ctx || !comm) return comm_fail(comm, NGSQ_ERR_INVALID_ARGUMENT, "null argument");
    const int rc = ngsq_bam_shard_begin(bam, ctx, (uint32_t)comm->rank, (uint32_t)comm->world, 0);
    return rc ? comm_fail(comm, rc, "%s", ngsq_bam_last_error()) : NGSQ_OK;
}

extern "C" int ngsq_bam_shard_verify(ngsq_bam *bam, ngsq_ctx *ctx, ngsq_comm *comm, ngsq_bam_shard_info *out, int *again) {
    if (!bam || !ctx || !comm || !out || !again) return comm_fail(comm, NGSQ_ERR_INVALID_ARGUMENT, "null argument");
    const uint32_t world = (uint32_t)comm->world, rank = (uint32_t)comm->rank;
    *again = 0;
    ngsq_bam_shard_info info{};
    // a rank whose scan failed still takes part in the collective so that nobody hangs
    int assumed = 0;
    const int rc_end = ngsq_bam_shard_peek(bam, &info, &assumed);
    const std::string why = rc_end ? ngsq_bam_last_error() : "";
    constexpr size_t W = 8;
    enum { R_N, R_BEGIN, R_END, R_FAILED, R_FIRST_KEY, R_LAST_KEY, R_ASSUMED };
    std::vector<uint64_t> rows(W * (size_t)world);
    const uint64_t mine[W] = {info.n_records, info.begin_voffset, info.end_voffset, (uint64_t)(rc_end != NGSQ_OK), info.first_key, info.last_key,
                              (uint64_t)assumed, 0};
    const int rc2 = ngsq_comm_allgather_host(comm, mine, rows.data(), sizeof mine);
    if (rc2) return rc2;
    auto row = [&](uint32_t k, int f) { return rows[W * k + (size_t)f]; };
    // Shard k+1 must begin where shard k's record chain ends (shards without a record start pass it on: begin == end).  By
    // induction from the header shard 0 is right, hence its end, hence shard 1's begin once corrected, ...: a shard whose begin
    // differs from its predecessor's end scans again from there (its end may change with it, so everybody compares again
    // afterwards; at most `world` rounds).  A shard whose scan FAILED while it ran on an assumed first record is treated the
    // same way -- a chain of plausible records inside somebody's auxiliary data can die later (round 4) -- but only
    // that once: a failure of shard 0, or of a scan from a confirmed offset, is the file's and every rank returns it.
    for (uint32_t k = 0; k < world; k++)
        if (row(k, R_FAILED) && (k == 0 || !row(k, R_ASSUMED))) {
            if (k == rank) return comm_fail(comm, rc_end, "%s", why.c_str());
            return comm_fail(comm, NGSQ_ERR_STATE, "shard %u of the file failed", k);
        }
    bool stable = true;
    uint64_t first = 0;
    for (uint32_t k = 0; k < world; k++) {
        if (row(k, R_FAILED)) stable = false;
        else if (k && !row(k - 1, R_FAILED)) stable = stable && row(k - 1, R_END) == row(k, R_BEGIN);
        if (k < rank) first += row(k, R_N);
    }
    if (!stable) {
        *again = 1;
        *out = info;
        // (behind a shard that failed nothing is known yet: its successor waits for the next round)
        if (rank && !row(rank - 1, R_FAILED) && (rc_end != NGSQ_OK || row(rank - 1, R_END) != info.begin_voffset)) {
            out->rescan = 1;
            // a failure to re-arm shows in the next round (the handle then holds a failed scan from a confirmed offset:
            // every rank gets the error); returning it here would leave the others waiting in that round's all-gather
            (void)ngsq_bam_shard_begin(bam, ctx, rank, world, row(rank - 1, R_END));
        }
        return NGSQ_OK;
    }
    info.first_record_index = first;
    *out = info;
    // a coordinate-sorted file stays sorted across the cuts (sorted_input contexts finish positions while they stream:
    // what the shard in front still covers is exchanged as a halo, a shard that starts EARLIER than its predecessor
    // ended would be tallied twice)
    if (ctx->stream_cov) {
        uint64_t prev_last = 0;
        bool have = false;
        for (uint32_t k = 0; k < world; k++) {
            if (!row(k, R_N)) continue;
            if (have && row(k, R_FIRST_KEY) < prev_last)
                return comm_fail(comm, NGSQ_ERR_UNSORTED, "sorted_input: shard %u begins in front of the last record of the shard before it", k);
            prev_last = row(k, R_LAST_KEY);
            have = true;
        }
    }
    return NGSQ_OK;
}
