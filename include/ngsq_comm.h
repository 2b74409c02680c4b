/*
 * ngsq_comm.h -- the multi-GPU step of the `ngs qc` hot path behind the C ABI: one process per GPU,
 * each scanning a contiguous BGZF block range of the file, and ONE exchange of integer state before
 * the sequence-facet teardown.
 *
 * Reference counterpart: none -- the reference is one thread (src/qc/command.rs:226-421).  What is
 * sharded is its two loops: pass 1 (command.rs:305-316) and pass 2 (command.rs:356-397); every facet's
 * state after `process` is a sum of per-record integer contributions, so the shards' states add.
 *
 * Transports (all collectives are issued in the same order by every rank):
 *   rccl    RCCL over xGMI, called directly from C++ on the context's stream (device buffers);
 *           the 128-byte unique id of rank 0 reaches the other ranks through the host program
 *           (a store, a pipe, MPI ...), as with ncclCommInitRank
 *   shm     POSIX shared memory between the processes of one node (host buffers; ranks that share
 *           one GPU -- test boxes -- and the bootstrap of `ngs qc --gpus N`)
 *   custom  three callbacks supplied by the host program (MPI, gloo, ...; host buffers)
 * With a host transport the exchanged device buffers are staged through host memory.
 */
#ifndef NGSQ_COMM_H
#define NGSQ_COMM_H

#include "ngsq.h"
#include "ngsq_bam.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ngsq_comm ngsq_comm;

#define NGSQ_COMM_ID_BYTES 128 /* = NCCL_UNIQUE_ID_BYTES */
#define NGSQ_COMM_MAX_WORLD 64

/* one message of a grouped point-to-point exchange */
typedef struct ngsq_p2p {
    int32_t peer;
    uint32_t reserved;
    void *buf;
    uint64_t bytes;
} ngsq_p2p;

/* a host transport supplied by the caller; every function returns 0 on success */
typedef struct ngsq_comm_ops {
    uint32_t struct_size; /* = sizeof(ngsq_comm_ops) */
    uint32_t reserved;
    void *user;
    /* element-wise wrap-around sum over all ranks, in place; elem_bytes is 4 (uint32) or 8 (uint64) */
    int (*allreduce_sum)(void *user, void *buf, uint64_t count, uint32_t elem_bytes);
    /* recv[r * bytes .. (r+1) * bytes) = rank r's send */
    int (*allgather)(void *user, const void *send, void *recv, uint64_t bytes);
    /* all sends and receives of one step, progressed together (messages between one pair of ranks
     * match in list order) */
    int (*sendrecv)(void *user, const ngsq_p2p *sends, uint32_t n_sends, const ngsq_p2p *recvs, uint32_t n_recvs);
} ngsq_comm_ops;

/* message of the last failing ngsq_comm_* / ngsq_exchange* call of this thread that had no object to keep it */
const char *ngsq_comm_last_error(const ngsq_comm *comm);

/* rank 0: ncclGetUniqueId */
int ngsq_comm_unique_id(uint8_t id[NGSQ_COMM_ID_BYTES]);
/* ncclCommInitRank on HIP device `device`; collective over all ranks */
int ngsq_comm_create_rccl(int rank, int world, const uint8_t id[NGSQ_COMM_ID_BYTES], int device, ngsq_comm **out);
/* shared-memory transport: rank 0 creates the segment `name` ("/ngsq-...": shm_open), the others wait for it;
 * slot_bytes = 0 -> 4 MiB per rank.  Unlinked when rank 0 destroys its communicator. */
int ngsq_comm_create_shm(const char *name, int rank, int world, uint64_t slot_bytes, ngsq_comm **out);
int ngsq_comm_create_custom(int rank, int world, const ngsq_comm_ops *ops, ngsq_comm **out);
void ngsq_comm_destroy(ngsq_comm *comm);
int ngsq_comm_rank(const ngsq_comm *comm);
int ngsq_comm_world(const ngsq_comm *comm);
/* "rccl", "shm" or "custom" */
const char *ngsq_comm_kind(const ngsq_comm *comm);
/* ncclGetVersion of the library the rccl transport bound (e.g. 22605), 0 when none could be loaded */
int ngsq_comm_rccl_version(void);
/* 1: an ngsq_comm_create_rccl of this process gave up on ncclCommInitRank (NGSQ_RCCL_INIT_TIMEOUT_S, default 60 s) and the
 * thread it ran on is still inside RCCL.  It cannot be cancelled, and the runtime's exit handlers may wait for it: such a
 * process leaves with _exit() once its work is done (the launchers of this repository do). */
int ngsq_comm_rccl_stuck(void);

/* Collectives on HOST buffers over any transport (rccl: staged through device memory): what the host side
 * of a sharded run needs -- agreeing on record boundaries, record counts, timings. */
int ngsq_comm_allgather_host(ngsq_comm *comm, const void *send, void *recv, uint64_t bytes);
int ngsq_comm_allreduce_host(ngsq_comm *comm, void *buf, uint64_t count, uint32_t elem_bytes);
int ngsq_comm_sendrecv_host(ngsq_comm *comm, const ngsq_p2p *sends, uint32_t n_sends, const ngsq_p2p *recvs,
                            uint32_t n_recvs);
int ngsq_comm_barrier(ngsq_comm *comm);

/* ---- the exchange ------------------------------------------------------------------------------------
 *
 * Call on every rank between the last ngsq_process_batch and ngsq_finalize.  Steps:
 *   1. all-reduce of the packed counter block (record facets, `seen`, error counts; ~130 KB) and, with Edits,
 *      of the refs/alts block;
 *   2. all-gather of the range of the coverage difference arrays each shard wrote; the axis is cut at the
 *      sorted range starts: every 4096-entry chunk gets exactly one OWNER;
 *   3. entries a shard wrote inside another shard's range (the read-length halo at a shard boundary: a few KB
 *      for coordinate-sorted shards) go to the owner point to point and are added there;
 *   4. all-gather of one word per rank (the sum of its owned range): the running depth in front of an owner
 *      is the sum of the words of the owners in front of it;
 *   5. every rank tears down its own chunks only (the scan of coverage.rs:182-246 is split N ways) and the
 *      partial results (depth histograms, bin totals, VAF histogram; ~100 KB) are all-reduced.
 * Shards whose written ranges overlap by more than NGSQ_HALO_LIMIT_BYTES (unsorted input) all-reduce the
 * whole depth block instead and split the scan evenly.  sorted_input contexts (streamed Coverage) take part
 * with the seams they left on the arrays; an exchanged entry that falls into a chunk its owner has already
 * finished fails on EVERY rank with NGSQ_ERR_UNSORTED (cov_head_guard too small, or shards out of order).
 * Afterwards every rank's ngsq_finalize / ngsq_get_* / ngsq_results_json give the whole-file result.
 */
#define NGSQ_HALO_LIMIT_BYTES (64ull << 20)

#define NGSQ_EXCHANGE_NONE 0u      /* no coverage state: counters only */
#define NGSQ_EXCHANGE_OWNER 1u     /* steps 2-5 */
#define NGSQ_EXCHANGE_ALLREDUCE 2u /* whole depth block summed, scan split evenly */

typedef struct ngsq_exchange_report {
    uint32_t struct_size; /* = sizeof(ngsq_exchange_report), set by the caller */
    uint32_t mode;        /* NGSQ_EXCHANGE_* */
    uint64_t halo_bytes_sent;
    uint64_t halo_bytes_received;
    uint64_t owned_chunk_lo, owned_chunk_hi; /* chunks this rank tore down */
    uint32_t host_syncs;  /* times the host waited for the device inside the call */
    uint32_t reserved;
} ngsq_exchange_report;

int ngsq_exchange(ngsq_ctx *ctx, ngsq_comm *comm, ngsq_exchange_report *report /* may be NULL */);

/* The ownership plan of step 2 as a pure function (every rank computes the same plan from the gathered
 * ranges).  ranges[2r], ranges[2r+1] = chunk range [lo, hi) rank r wrote (lo == hi: nothing).
 * own[2r], own[2r+1] = the range rank r owns ((0,0): none); order[0..*n_owners) = owners by position;
 * xfer = up to xfer_cap rows of (src, dst, chunk_lo, chunk_hi); returns the number of transfers, or a
 * negative status. */
int64_t ngsq_exchange_plan(const uint64_t *ranges, uint32_t world, uint64_t n_chunks, uint64_t *own, uint32_t *order,
                           uint32_t *n_owners, uint64_t *xfer, uint64_t xfer_cap);

/* The same protocol over shard state that lives elsewhere -- another device runtime, or plain host arrays (this
 * is how the CPU tests drive the protocol without a GPU).  `memory` says where every pointer of the struct
 * lives; the operations are ordered with respect to each other and to the transport (device: one stream). */
typedef struct ngsq_shard_state {
    uint32_t struct_size; /* = sizeof(ngsq_shard_state) */
    uint32_t memory;      /* NGSQ_MEM_HOST or NGSQ_MEM_DEVICE */
    void *user;
    void *stream;         /* hipStream_t of the operations (device memory), else NULL */
    uint64_t *counters;
    uint64_t n_counters;
    uint32_t *depth;      /* n_diff difference entries | n_chunks chunk sums | scratch of the scan */
    uint64_t n_depth;
    uint64_t n_diff, n_chunks;
    uint64_t *teardown;   /* partial teardown results */
    uint64_t n_teardown;
    uint32_t *edits;      /* refs/alts block or NULL */
    uint64_t n_edits;
    const uint8_t *chunk_flags; /* [n_chunks] 1 = finished while streaming; NULL for array contexts */
    const uint64_t *touched;    /* [2] first / one-past-last depth entry written, touched[0] = ~0: nothing */
    /* wait until every operation issued so far has completed */
    int (*synchronize)(void *user);
    /* diff entries of chunks [c0, c1) += diff[], chunk sums [c0, c1) += sums[] (wrap-around) */
    int (*halo_add)(void *user, uint64_t c0, uint64_t c1, const uint32_t *diff, const uint32_t *sums);
    /* out2[0] = sum of the chunk sums [b0, b1) mod 2^32; out2[1] = 1 iff a chunk flag is set inside any of the
     * n_in chunk ranges in_ranges[2k], in_ranges[2k+1] (host array) */
    int (*summary)(void *user, uint64_t b0, uint64_t b1, const uint64_t *in_ranges, uint32_t n_in, uint32_t *out2);
    /* sequence-facet teardown of chunks [b0, b1) with the running depth in front = sum over the set bits r of
     * front_mask of words[2r] (words: [world][2] in state memory, may be NULL when front_mask == 0);
     * Edits: VAF histogram of the part-th of `parts` equal slices of every sequence's positions */
    int (*teardown_range)(void *user, uint64_t b0, uint64_t b1, const uint32_t *words, uint64_t front_mask,
                          uint32_t part, uint32_t parts);
} ngsq_shard_state;

int ngsq_exchange_state(const ngsq_shard_state *state, ngsq_comm *comm, ngsq_exchange_report *report);

/* ---- one BAM file, several GPUs (include/ngsq_bam.h "sharded device ingest"); shard = ngsq_comm_rank of ngsq_comm_world.
 *   ngsq_bam_shard_open    ngsq_bam_shard_begin for this rank (no communication): ngsq_bam_next_batch_device then streams
 *                          this shard's records
 *   ngsq_bam_shard_verify  collective, after the last batch and BEFORE ngsq_exchange: one all-gather of (records, assumed
 *                          begin, found end, failure flag, first / last sort key).  *again = 0: the boundaries are exact,
 *                          out->first_record_index = records of the shards in front; a sorted_input context whose
 *                          neighbours are out of coordinate order gets NGSQ_ERR_UNSORTED (on every rank).  *again = 1 (on
 *                          every rank): some shard's assumption was wrong; the ranks with out->rescan = 1 have been re-armed
 *                          from the confirmed offset -- ngsq_reset the context, scan again -- the others keep their state
 *                          (their reader is at its end); then everybody calls ngsq_bam_shard_verify again.
 * A rank whose scan failed calls ngsq_bam_shard_verify all the same: every rank then returns an error instead of
 * waiting for it. */
int ngsq_bam_shard_open(ngsq_bam *bam, ngsq_ctx *ctx, ngsq_comm *comm);
int ngsq_bam_shard_verify(ngsq_bam *bam, ngsq_ctx *ctx, ngsq_comm *comm, ngsq_bam_shard_info *out, int *again);

#ifdef __cplusplus
}
#endif
#endif /* NGSQ_COMM_H */
