// bam_device_reader.cpp -- device ingest (include/ngsq_bam.h): compressed BGZF bytes cross PCIe,
// the GPU inflates them and parses the BAM records into structure-of-arrays columns.
#include <hip/hip_runtime_api.h>
#include <pthread.h>
#include <sched.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>

#include <cctype>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/ngsq_bam.h"
#include "bam_reader.h"
#include "bgzf.h"
#include "context.h"
#include "ingest_kernels.h"
#include "mem_pool.h"

using namespace ngsq;

namespace {

int dfail(ngsq_ctx *c, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf;
    return code;
}

#define DHIP(c, expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return dfail(c, NGSQ_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

const char *inflate_status_text(uint32_t s) {
    switch (s) {
    case INF_BAD_BLOCK_TYPE: return "invalid DEFLATE block type";
    case INF_BAD_STORED_LEN: return "stored block length check failed";
    case INF_BAD_CODE_LENGTHS: return "invalid Huffman code lengths";
    case INF_BAD_SYMBOL: return "invalid literal/length or distance code";
    case INF_BAD_DISTANCE: return "match distance reaches before the block";
    case INF_OUTPUT_OVERRUN: return "more data than ISIZE";
    case INF_INPUT_OVERRUN: return "compressed data ends inside a symbol";
    case INF_SIZE_MISMATCH: return "decompressed size differs from ISIZE";
    case INF_CRC_MISMATCH: return "CRC mismatch";
    default: return "ok";
    }
}


// growable device buffer
// a device array that only grows; its memory comes from (and goes back to) the process's block cache (mem_pool.h)
template <typename T> struct DevBuf {
    T *p = nullptr;
    size_t cap = 0;   // elements
    size_t bytes = 0; // of the block behind p
    hipError_t reserve(size_t n, bool keep = false) {
        if (n <= cap) return hipSuccess;
        const size_t want = n + n / 8 + 64;
        void *q = nullptr;
        size_t got = 0;
        hipError_t e = ngsq::pool_device_alloc(&q, want * sizeof(T), &got);
        if (e != hipSuccess) return e;
        if (keep && p && cap) (void)hipMemcpy(q, p, cap * sizeof(T), hipMemcpyDeviceToDevice);
        ngsq::pool_device_free(p, bytes);
        p = static_cast<T *>(q);
        cap = got / sizeof(T);
        bytes = got;
        return hipSuccess;
    }
    ~DevBuf() { ngsq::pool_device_free(p, bytes); }
};

// pinned host memory the device addresses directly (hipHostMalloc: mapped and coherent), grown on demand
struct PinBuf {
    void *h = nullptr, *dev = nullptr; // the same memory as the host and as the device see it
    size_t cap = 0;
    hipError_t reserve(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (h) (void)hipHostFree(h);
        h = dev = nullptr;
        cap = 0;
        const size_t want = bytes + bytes / 4 + 4096;
        hipError_t e = hipHostMalloc(&h, want, hipHostMallocMapped);
        if (e != hipSuccess) return e;
        e = hipHostGetDevicePointer(&dev, h, 0);
        if (e != hipSuccess) return e;
        cap = want;
        return hipSuccess;
    }
    ~PinBuf() {
        if (h) (void)hipHostFree(h);
    }
};

// NGSQ_INGEST_TRACE=1: wall-clock of the ingest stages on stderr (measurement aid)
bool trace_on() {
    static const bool on = getenv("NGSQ_INGEST_TRACE") && atoi(getenv("NGSQ_INGEST_TRACE"));
    return on;
}
double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// Keep the calling thread (and the threads it starts) on the CPUs of the NUMA node the device hangs off.  The reader
// copies the file out of the page cache into pinned memory with a dozen threads: left to the scheduler they end up spread
// over both sockets and the same copy takes 31 ms per chunk instead of 8-14 (measured on a two-socket MI355X host; which
// node holds the page cache matters less than not straddling them).  NGSQ_READER_NODE=-1 turns it off, =k picks node k.
void pin_to_device_node(int device) {
    int node = -1;
    if (const char *e = getenv("NGSQ_READER_NODE")) {
        node = atoi(e);
        if (node < 0) return;
    } else {
        char bdf[64] = {0};
        if (hipDeviceGetPCIBusId(bdf, (int)sizeof bdf, device) != hipSuccess) return;
        for (char *p = bdf; *p; p++) *p = (char)tolower((unsigned char)*p);
        const std::string path = std::string("/sys/bus/pci/devices/") + bdf + "/numa_node";
        if (FILE *f = fopen(path.c_str(), "r")) {
            if (fscanf(f, "%d", &node) != 1) node = -1;
            fclose(f);
        }
        if (node < 0) return;
    }
    char list[4096] = {0};
    const std::string path = "/sys/devices/system/node/node" + std::to_string(node) + "/cpulist";
    FILE *f = fopen(path.c_str(), "r");
    if (!f) return;
    const bool got = fgets(list, sizeof list, f) != nullptr;
    fclose(f);
    if (!got) return;
    cpu_set_t set;
    CPU_ZERO(&set);
    int n_set = 0;
    for (char *tok = strtok(list, ",\n"); tok; tok = strtok(nullptr, ",\n")) {
        int a = 0, b = 0;
        const int k = sscanf(tok, "%d-%d", &a, &b);
        if (k < 1) continue;
        if (k == 1) b = a;
        for (int c = a; c <= b && c < CPU_SETSIZE; c++) {
            CPU_SET(c, &set);
            n_set++;
        }
    }
    if (n_set) (void)pthread_setaffinity_np(pthread_self(), sizeof set, &set); // a cpuset that excludes them: stay as we are
    if (trace_on()) fprintf(stderr, "[ingest] reader threads on NUMA node %d (%d CPUs)\n", node, n_set);
}

size_t env_mb(const char *name, size_t dflt_mb) {
    const char *e = getenv(name);
    const long v = e ? atol(e) : 0;
    return (size_t)(v > 0 ? v : (long)dflt_mb) << 20;
}

} // namespace

namespace ngsq {

// State of the device ingest of one BAM file: compressed chunk -> inflate -> record index -> batches.
struct DeviceIngest {
    FILE *f = nullptr;
    ngsq_ctx *ctx = nullptr;
    int device = -1; // ctx->device, kept for the destructor (the context may be gone by then)
    size_t raw_cap = 0, comp_chunk = 0;
    // Compressed chunks come from a reader thread: while the GPU inflates and parses chunk k the
    // thread reads and frames chunk k+1 into the other pinned buffer.
    struct HostChunk {
        uint8_t *h = nullptr; // the chunk's compressed bytes: h[j] = byte file_off + j of the file -- in `pin` (behind a pad that makes
                              // the address congruent to the file offset mod 4096: O_DIRECT reads land where they belong), or, for a
                              // chunk whose bytes are in the page cache, in the MAPPING of the file itself (nothing is copied by the host)
        uint8_t *pin = nullptr; // pinned block, 2 x comp_chunk + 2 pages (mem_pool.h)
        size_t h_bytes = 0;     // of the block behind pin
        bool mapped = false;    // h points into the file's mapping
        size_t fill = 0, consumed = 0;
        uint64_t total = 0; // decompressed bytes of `blocks`
        std::vector<BgzfBlock> blocks;
        uint64_t file_off = 0;      // file offset of h[0] (a block start)
        std::vector<uint64_t> coff; // file offset of every block of `blocks`
        bool ready = false, last = false; // last: the byte range ends with this chunk
        std::string err;
        PinBuf tab; // [block table | file offsets], pinned: the reader thread sends them behind the chunk's bytes, on the copy stream
    } hc[4];
    // Round 4: a chunk's context -- pinned buffer, device copy of the compressed bytes, block tables, Pending -- is one of NC
    // = 4 used in turn (chunk j: context j % 4), its inflated bytes go to one of NR = 3 raw buffers (j % 3), and the inflates
    // alternate between two streams.  Until then there were two of everything, on one inflate stream: a context was refilled
    // -- read, framed, sent across PCIe: ~5 ms -- only when the inflate of its chunk had finished and had to be there again
    // one inflate later; and the decoders of chunk j + 1 started when the LAST decoder of chunk j had finished -- a launch
    // of 9 k blocks on 6 k resident decoders ends with half the device waiting for the blocks still in work (the same
    // bytes inflate 22 % faster per byte in launches twice the size).  Now the inflates of chunks j + 1 and j + 2 are queued
    // while chunk j is parsed: j + 1 on the other stream, so that its decoders take the wave slots chunk j's leave.
    static constexpr int NC = 4, NR = 3;
    // the compressed bytes of a chunk cross PCIe on their own stream as soon as the reader thread has
    // framed them, i.e. while the GPU works on the previous chunk
    DevBuf<uint8_t> d_comp_slot[NC];
    hipStream_t copy_stream = nullptr;
    hipEvent_t h2d_done[NC] = {nullptr, nullptr, nullptr, nullptr};
    bool h2d_issued[NC] = {false, false, false, false};
    std::thread reader;
    std::mutex mu;
    std::condition_variable cv;
    bool stop = false;
    bool file_done = false; // the consumer has taken the last chunk
    bool first_chunk = true;
    // The inflate of chunk k+1 runs on its own stream while chunk k is indexed, cut into columns and scanned on the
    // context's stream: two raw buffers, each with CARRY_MAX bytes of headroom in front of the inflated data for the
    // record the previous chunk ended in (so where a chunk is inflated to does not depend on the chunk before it).
    struct Pending {
        bool issued = false, last = false;
        size_t n_blk = 0, consumed = 0;
        uint64_t total = 0, next_coff = 0; // next_coff: file offset behind the last block
        std::vector<BgzfBlock> blocks;
        std::vector<uint64_t> coff;
        // the decoders' verdicts come back into PINNED memory: an "asynchronous" copy into pageable memory makes the
        // calling thread wait for everything queued on the stream in front of it -- here the whole inflate of the next
        // chunk, so that the parse kernels of this chunk were launched only when it had finished (the two streams never
        // overlapped, rounds 1 and 2)
        // (the block table and the file offsets travel the other way through the same pinned block: a copy FROM pageable
        // memory is staged, and waits for the stream too when the staging buffers are in use)
        PinBuf pin; // [status | block table | file offsets]
        uint32_t *status = nullptr;
        BgzfBlock *pin_blocks = nullptr;
        uint64_t *pin_coff = nullptr;
        size_t status_cap = 0;
        int rslot = 0; // the raw buffer its bytes are inflated to
        std::string err;
    } pend[NC];
    PinBuf h_cand, h_seg, h_small; // the candidate table, the segments' verdicts, a few result words: host <-> device without DMA
    // The block table and the blocks' file offsets of a chunk are read by its inflate AND, later, by the column kernels of its
    // batches (the records' ids): they belong to the chunk's context, which goes back to the reader thread when the chunk is
    // RETIRED -- the next chunk has been taken and everything that reads this one's buffers has been queued on the context's
    // stream; retired_ev[k] marks that point of the stream, the reader's copies into context k wait for it.
    DevBuf<uint64_t> d_coff_s[NC], d_record_id;
    DevBuf<BgzfBlock> d_blocks_s[NC];
    DevBuf<uint32_t> d_status_s[NC];
    hipEvent_t retired_ev[NC] = {nullptr, nullptr, nullptr, nullptr}, inf_done[NC] = {nullptr, nullptr, nullptr, nullptr};
    bool retired_set[NC] = {false, false, false, false}; // (under mu)
    uint64_t chunks_issued = 0, chunks_loaded = 0; // inflates queued / chunks taken by load_chunk
    int inflate_ahead = 2, inflate_streams = 2;    // inflates queued beyond the chunk being parsed; streams they alternate between (NGSQ_INFLATE_AHEAD, NGSQ_INFLATE_STREAMS: A/B measurements)
    DevBuf<uint8_t> d_rawb[NR];
    hipStream_t inf_stream[2] = {nullptr, nullptr};
    bool inf_low_priority = true;
    hipEvent_t raw_free[NR] = {nullptr, nullptr, nullptr};
    bool raw_free_set[NR] = {false, false, false};
    uint8_t *raw = nullptr; // the inflated bytes being indexed / cut into batches: d_raw (sharded mode) or a view into a raw buffer
    // device
    DevBuf<uint8_t> d_comp, d_seq, d_qual, d_scan_tmp;
    DevBuf<BgzfBlock> d_blocks;
    DevBuf<uint32_t> d_status, d_l_seq, d_cigar;
    DevBuf<RecPieces> d_pieces;
    DevBuf<uint64_t> d_var_base;              // per record of the batch: offset of its CIGAR in raw
    DevBuf<uint64_t> d_rec_off, d_len; // d_len: seq | qual | cigar lengths -> offsets
    DevBuf<unsigned long long> d_small;       // REC_WORK_WORDS words shared by k_rec_offsets / k_rec_fixed (ingest_kernels.h RecWork)
    uint32_t inf_ctr_base[2] = {0, 0};        // what the decoders' counters (d_small[W_INF0 / W_INF1]) hold before the next launch
    DevBuf<uint16_t> d_flag, d_n_cigar;
    DevBuf<uint8_t> d_mapq;
    DevBuf<int32_t> d_ref_id, d_pos, d_mate, d_tlen;
    uint64_t raw_len = 0, tail_off = 0;
    uint64_t n_rec = 0, cursor = 0; // records indexed in the current chunk / handed out
    uint64_t blocks_done = 0;
    // The byte range of the file this ingest scans.  The whole file: [0, size).  A shard (ngsq_bam_shard_begin): the
    // blocks that start in [pos_lo, pos_hi) are its own -- a record belongs to the shard its first byte lies in --
    // and the reader goes on to pos_end for the blocks that complete its last record.
    uint64_t file_size = 0, pos_lo = 0, pos_hi = 0, pos_end = 0;
    bool sharded = false;
    int reader_threads = 0; // 0: the cgroup's quota less two
    // where the first record starts: behind the header (shard 0), at a given offset of the first block's data (a
    // confirmed virtual offset), or wherever the first plausible record chain of the first chunk starts (an
    // assumption the neighbour shard confirms after the scan: ngsq_bam_shard_verify)
    enum EntryMode { ENTRY_HEADER, ENTRY_KNOWN, ENTRY_FIND } entry_mode = ENTRY_HEADER;
    uint64_t entry_uoff = 0;
    bool boundary_passed = false; // a block at or behind pos_hi has been seen
    bool carry_owned = true;      // the record cut by the previous chunk's end started in front of the boundary
    bool complete = false;        // every record of the range has been handed out
    uint64_t carry_id = 0, tail_id = 0; // virtual offset of the record carried into / cut by the end of the current chunk
    uint64_t begin_voffset = 0, end_voffset = 0;
    bool have_begin = false;
    uint64_t n_own = 0;           // records handed out
    uint64_t header_bytes0 = 0;   // the handle's header_bytes when the ingest began (load_chunk consumes it)
    unsigned long long first_key = 0, last_key = 0; // refID << 32 | pos of the first / last record handed out
    double t_start = now_ms();    // (NGSQ_INGEST_TRACE)
    int cur_slot = 0;             // context of the chunk being handed out
    uint64_t carry_len = 0;       // bytes of the view in front of the current chunk's first byte
    bool last_chunk = false;      // the chunk being handed out is the range's last
    ngsq_bam_ingest_stats stats{}; // ngsq_bam_device_stats
    uint64_t chunk_first_record = 0; // records handed out before the current chunk (for the message of an invalid record)
    ~DeviceIngest() {
        {
            std::lock_guard<std::mutex> g(mu);
            stop = true;
        }
        cv.notify_all();
        if (reader.joinable()) reader.join();
        if (f) fclose(f);
        if (device >= 0) (void)hipSetDevice(device); // (the cache is per device)
        if (copy_stream) {
            (void)hipStreamSynchronize(copy_stream);
            pool_stream_put(false, copy_stream);
        }
        for (auto &q : inf_stream)
            if (q) {
                (void)hipStreamSynchronize(q);
                pool_stream_put(inf_low_priority, q);
            }
        for (auto &e : h2d_done) pool_event_put(e);
        for (auto &e : inf_done) pool_event_put(e);
        for (auto &e : raw_free) pool_event_put(e);
        for (auto &e : retired_ev) pool_event_put(e);
        for (auto &c : hc)
            if (c.pin) pool_pinned_free(c.pin, c.h_bytes);

    }
};

// a record cut by a chunk boundary is carried into the next chunk: room kept for it in the ingest buffer
constexpr uint64_t CARRY_MAX = (uint64_t)1 << 24;
// index_records: the chain starts at the first plausible record chain of the view (a shard that does not know yet)
constexpr uint64_t FIND_ENTRY = ~0ull;
// compressed bytes a shard's reader goes on behind its boundary for the end of its last record (any record the carry
// space can hold fits, stored blocks included)
constexpr uint64_t SHARD_EXTRA = CARRY_MAX + ((uint64_t)1 << 20);

} // namespace ngsq

namespace {

void free_ingest(DeviceIngest *d) { delete d; }

#define BHIP(expr)                                                                                          \
    do {                                                                                                    \
        hipError_t e_ = (expr);                                                                             \
        if (e_ != hipSuccess) return ngsq_bam_fail(NGSQ_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// The reader's pread workers: started once (they inherit the reader thread's CPU affinity), woken per request.
// (Threads created anew for every 64 MiB step spent a quarter of the step getting onto CPUs of their own.)  A request is a
// run of the file cut into pieces of 4 MiB that the workers take IN ORDER, one after the other: the bytes arrive as a
// growing prefix, the reader thread frames and sends what has arrived while the rest is on its way, and there is no point
// at which all the workers wait for the slowest of them -- with one barrier per 64 MiB step the storage under a file that is
// not in the page cache saw its queue drain sixty times per second (0.62 s for the cold 6 GB file, 0.53 s now).
struct ReadPool {
    static constexpr int NT_MAX = 32;
    // pieces of 4 MiB when the bytes come out of the page cache (the prefix grows smoothly: framing and the copies to the device
    // keep up with the reads), of 16 MiB when they come from storage (measured on the boxes' overlay filesystem: 12 GB/s with
    // 4 MiB requests, 15 with 16 MiB; from the page cache the large pieces cost a tenth of the scan) -- the reader thread
    // chooses by the rate of the request before
    static constexpr size_t PIECE = (size_t)4 << 20, PIECE_COLD = (size_t)16 << 20, MAX_PIECES = 1024; // a request is at most 4 GiB
    size_t piece = PIECE;
    int nt = 0, fd = -1;
    int fd_direct = -1;  // the same file opened O_DIRECT (-1: not available): requests marked `direct` read through it
    bool direct = false; // this request: whole 4 KiB sectors straight into the pinned buffer -- no page cache, no kernel copy
    std::thread th[NT_MAX];
    std::mutex mu;
    std::condition_variable cv_go, cv_done;
    uint64_t gen = 0;
    bool quit = false;
    uint8_t *dst = nullptr;
    uint64_t pos = 0;
    size_t want = 0, n_pieces = 0, next = 0, finished = 0;
    uint32_t got[MAX_PIECES] = {};
    bool done[MAX_PIECES] = {};
    bool bad = false;

    void open(int n, int file) {
        nt = n;
        fd = file;
        // O_DIRECT requests go straight to the storage's queue: two readers keep it full on the boxes measured (6 GB cold: 0.39 s
        // with one, 0.37 with two, 0.44 with fourteen -- and fourteen times the CPU); NGSQ_DIRECT_THREADS for other storage
        if (const char *e = getenv("NGSQ_DIRECT_THREADS")) direct_nt = std::max(1, atoi(e));
        for (int t = 0; t < nt; t++) th[t] = std::thread([this, t] { work(t); });
    }
    int direct_nt = 2;
    void work(int idx) {
        uint64_t seen = 0;
        std::unique_lock<std::mutex> g(mu);
        for (;;) {
            cv_go.wait(g, [&] { return quit || (gen != seen && next < n_pieces); });
            if (quit) return;
            if (direct && fd_direct >= 0 && idx >= direct_nt) { // this request is the first readers' alone
                seen = gen;
                continue;
            }
            const uint64_t my_gen = gen;
            while (gen == my_gen && next < n_pieces) {
                const size_t i = next++;
                const size_t lo = i * piece, hi = std::min(want, lo + piece);
                uint8_t *const to = dst;
                const uint64_t from = pos;
                const bool dio = direct && fd_direct >= 0;
                g.unlock();
                size_t n = 0;
                bool err = false, buffered = !dio;
                if (dio) {
                    // The buffer address of a file byte is congruent to its file offset mod 4096 (reader_main keeps it so), so the
                    // sectors that hold [from + lo, from + hi) can be read as they lie: up to 4095 bytes in front of the piece and
                    // behind it are read too -- bytes of the neighbouring pieces (or of the data in front of the request), which
                    // land on the addresses those same bytes have anyway.
                    const uint64_t a0 = (from + lo) & ~(uint64_t)4095, a1 = (from + hi + 4095) & ~(uint64_t)4095;
                    uint8_t *const at = to + lo - ((from + lo) - a0);
                    uint64_t got_to = a0; // file offset the sectors read so far reach
                    while (got_to < a1) {
                        const ssize_t r = pread(fd_direct, at + (got_to - a0), (size_t)(a1 - got_to), (off_t)got_to);
                        if (r < 0) {
                            if (got_to == a0 && (errno == EINVAL || errno == ENOTSUP)) { // this filesystem refuses after all: the ordinary way
                                got_to = 0;
                                break;
                            }
                            err = true;
                            break;
                        }
                        if (r == 0) break; // end of file
                        got_to += (uint64_t)r;
                    }
                    if (got_to) n = got_to > from + lo ? (size_t)std::min<uint64_t>(got_to - (from + lo), hi - lo) : 0;
                    else if (!err) buffered = true;
                }
                if (buffered) {
                    while (lo + n < hi) {
                        const ssize_t r = pread(fd, to + lo + n, hi - lo - n, (off_t)(from + lo + n));
                        if (r < 0) {
                            err = true;
                            break;
                        }
                        if (r == 0) break; // end of file
                        n += (size_t)r;
                    }
                }
                g.lock();
                got[i] = (uint32_t)n;
                done[i] = true;
                bad = bad || err;
                finished++;
                cv_done.notify_all();
            }
            seen = my_gen;
        }
    }
    // (the previous request must have been waited out: wait_prefix(...) returned all = true)
    void start(uint8_t *to, uint64_t file_pos, size_t n, bool cold) {
        {
            std::lock_guard<std::mutex> g(mu);
            dst = to;
            pos = file_pos;
            want = n;
            direct = cold;
            piece = cold ? PIECE_COLD : PIECE;
            n_pieces = (n + piece - 1) / piece;
            next = finished = 0;
            bad = false;
            for (size_t i = 0; i < n_pieces; i++) done[i] = false;
            gen++;
        }
        cv_go.notify_all();
    }
    // Wait until the bytes that have arrived without a gap reach `at_least` (or every piece is in): *avail = those bytes
    // (they end at the first short piece: the end of the file), *all = nothing is on its way any more.
    void wait_prefix(size_t at_least, size_t *avail, bool *all, bool *short_read) {
        std::unique_lock<std::mutex> g(mu);
        for (;;) {
            size_t a = 0, i = 0;
            bool cut = false;
            for (; i < n_pieces && done[i]; i++) {
                a += got[i];
                if (got[i] < std::min(want, (i + 1) * piece) - i * piece) { // a short piece: nothing behind it counts
                    cut = true;
                    break;
                }
            }
            const bool fin = finished == n_pieces;
            if (fin || a >= at_least) {
                *avail = a;
                *all = fin;
                *short_read = cut;
                return;
            }
            cv_done.wait(g);
        }
    }
    ~ReadPool() {
        {
            std::lock_guard<std::mutex> g(mu);
            quit = true;
        }
        cv_go.notify_all();
        for (int t = 0; t < nt; t++)
            if (th[t].joinable()) th[t].join();
    }
};

// Reader thread: fill the pinned buffers alternately with whole BGZF blocks (the bytes of a block
// cut by the end of a chunk start the next one).  A chunk is as many blocks as inflate to `out_limit`
// bytes; the file is read in steps of 64 MiB -- sixteen pread()s in parallel, one thread copying out
// of the page cache into pinned memory (~7 GB/s) would be slower than the GPU inflates -- and each step is
// framed while the next one is being read.  How many compressed bytes a chunk needs is estimated from the
// ratio seen so far, so that only a few blocks' worth of bytes are left over (they are copied to the other
// buffer: reading a fixed amount left 150 MB per chunk to copy twice).
void reader_main(DeviceIngest *d, std::string path) {
    std::vector<uint8_t> leftover;
    uint64_t file_pos = d->pos_lo; // next byte of the file to read
    bool eof = file_pos >= d->pos_end;
    const bool range_cut = d->pos_end < d->file_size; // the range ends inside the file: its last block may be cut, by design
    const size_t cap = 2 * d->comp_chunk;
    const uint64_t out_limit = d->raw_cap > 2 * CARRY_MAX ? d->raw_cap - CARRY_MAX : d->raw_cap / 2;
    constexpr size_t STEP = (size_t)32 << 20; // bytes framed and sent at a time
    const uint64_t ramp_first = env_mb("NGSQ_RAMP_FIRST_MB", 32), ramp_shift = getenv("NGSQ_RAMP_SHIFT") ? (uint64_t)atoi(getenv("NGSQ_RAMP_SHIFT")) : 2; // (measurement aids)
    constexpr int NT_MAX = ReadPool::NT_MAX;
    // leave two cores of the quota to the thread that drives the GPU and to this one (it frames while the others read)
    // (the workers of a sharded run share the quota: ngsq_bam_shard_begin sets reader_threads)
    // (NGSQ_READER_THREADS: a measurement aid for whole files -- tools/reader_threads.sh -- and the override for shards)
    const int env_nt = getenv("NGSQ_READER_THREADS") ? atoi(getenv("NGSQ_READER_THREADS")) : 0;
    const int NT = d->reader_threads > 0 ? std::min(NT_MAX, d->reader_threads)
                   : env_nt > 0          ? std::min(NT_MAX, env_nt)
                                         : std::max(4, std::min(NT_MAX, effective_cores() - 2));
    double ratio = 0.0; // compressed bytes per inflated byte, from the chunks framed so far
    bool cold = false;  // the last request came from storage, not from the page cache
    const int fd = fileno(d->f);
    // this thread, its pread workers and the pinned buffers they fill: all on the device's NUMA node
    pin_to_device_node(d->ctx->device);
    // ---- three ways for a chunk's bytes to reach the device (round 6; tools/reader_paths_probe.cpp):
    //   mapped   the bytes are in the page cache: the chunk IS the file's mapping -- framed in place, copied to the device straight
    //            from it (the runtime pins the pages it reads; no copy by the host: 30-38 GB per core-second against 12-14 through
    //            pread, one thread instead of four)
    //   direct   they are not: O_DIRECT reads into the pinned buffer -- no kernel copy, no page cache pollution (35 GB per
    //            core-second against 3.8 for buffered reads of a cold file, and the storage's full rate)
    //   pread    buffered reads into the pinned buffer (rounds 1-5): mappings or O_DIRECT not available, NGSQ_READER_PATH=pread
    const char *path_env = getenv("NGSQ_READER_PATH"); // measurement aid: mapped | direct | pread | auto
    const std::string path_mode = path_env ? path_env : "auto";
    const uint8_t *fmap = nullptr;
    if (path_mode == "auto" || path_mode == "mapped") {
        void *m = d->file_size ? mmap(nullptr, d->file_size, PROT_READ, MAP_PRIVATE, fd, 0) : MAP_FAILED;
        if (m != MAP_FAILED) fmap = static_cast<const uint8_t *>(m);
    }
    int fd_direct = -1;
    if (path_mode == "auto" || path_mode == "direct") fd_direct = open(path.c_str(), O_RDONLY | O_DIRECT | O_CLOEXEC);
    struct Closer {
        const uint8_t *m;
        uint64_t n;
        int fd;
        ~Closer() {
            if (m) munmap(const_cast<uint8_t *>(m), n);
            if (fd >= 0) close(fd);
        }
    } closer{fmap, d->file_size, fd_direct};
    // how much of file[lo, hi) the page cache holds: 32 pages spread over the range are looked at (mincore over a whole chunk walks
    // 65 k page-cache entries per call: it cost a warm 6 GB scan 0.09 of its 0.20 s)
    auto resident = [&](uint64_t lo, uint64_t hi) -> double {
        if (!fmap || hi <= lo) return 0.0;
        size_t seen = 0, in = 0;
        for (int k = 0; k < 32; k++) {
            const uint64_t a = (lo + (hi - lo) / 32 * (uint64_t)k) & ~(uint64_t)4095;
            unsigned char v = 0;
            if (a < d->file_size && mincore(const_cast<uint8_t *>(fmap) + a, 4096, &v) == 0) {
                seen++;
                in += v & 1;
            }
        }
        return seen ? (double)in / (double)seen : 0.0;
    };
    ReadPool pool;
    pool.fd_direct = fd_direct;
    bool pool_open = false; // (a file that is scanned out of the page cache never starts the pread workers)
    if (hipSetDevice(d->ctx->device) != hipSuccess) {
        {
            std::lock_guard<std::mutex> g(d->mu);
            d->hc[0].err = path + ": hipSetDevice failed in the reader thread";
            d->hc[0].last = true;
            d->hc[0].ready = true;
        }
        d->cv.notify_all();
        return;
    }
    uint64_t chunk_no = 0;
    for (int k = 0;; k = (k + 1) % DeviceIngest::NC, chunk_no++) {
        DeviceIngest::HostChunk &c = d->hc[k];
        // The pipeline fills gradually: nothing can be parsed before the first chunk has been read, copied and inflated, so
        // the first one is small (32 MiB of records) and each of the next is four times its predecessor until the full size
        // (32 / 128 / 512 MiB): the first records reach the facets 15 ms into the scan.  (Twice its predecessor until the
        // decoder got faster in round 3 -- a chunk's fixed costs, launches and host round trips, then weighed more than the
        // shorter waits: 0.217-0.221 s -> 0.198-0.213 s for the 6 GB file.)
        const uint64_t limit = chunk_no * ramp_shift < 12 ? std::min<uint64_t>(out_limit, (uint64_t)ramp_first << (chunk_no * ramp_shift)) : out_limit;
        {
            std::unique_lock<std::mutex> g(d->mu);
            d->cv.wait(g, [&] { return d->stop || !c.ready; });
            if (d->stop) return;
        }
        // ---- where this chunk's bytes are: in the page cache (then the chunk is the mapping), or not (pinned buffer, O_DIRECT)
        const uint64_t chunk_off = file_pos - leftover.size(); // file offset of the chunk's first byte
        {
            const uint64_t look = std::min<uint64_t>(d->pos_end, chunk_off + std::min<uint64_t>(cap, (uint64_t)256 << 20));
            const double in_cache = (path_mode == "direct" || path_mode == "pread") ? 0.0 : resident(chunk_off, look);
            // (auto does not take the mapped path: in the pipeline it is cheaper per byte than pread but slower on the wall -- 0.35-0.74 s
            // against 0.20-0.33 s for the 6 GB file -- see the table in DESIGN.md section 8)
            c.mapped = fmap && path_mode == "mapped";
            cold = !c.mapped && fd_direct >= 0 && (path_mode == "direct" || in_cache < 0.5);
        }
        if (!c.mapped) {
            if (!c.pin && ngsq::pool_pinned_alloc((void **)&c.pin, cap + 8192, &c.h_bytes) != hipSuccess) {
                {
                    std::lock_guard<std::mutex> g(d->mu);
                    c.err = path + ": hipHostMalloc of the ingest buffers failed";
                    c.last = true;
                    c.ready = true;
                }
                d->cv.notify_all();
                return;
            }
            if (!pool_open) {
                pool.open(NT, fd);
                pool_open = true;
            }
            c.h = c.pin + (chunk_off & 4095); // address == file offset (mod 4096): O_DIRECT reads whole sectors in place
        } else {
            c.h = const_cast<uint8_t *>(fmap) + chunk_off;
        }
        const double tr0 = now_ms();
        double t_frame = 0, t_send = 0, t_join = 0, t_spawn = 0;
        int n_steps = 0;
        if (!c.mapped) memcpy(c.h, leftover.data(), leftover.size()); // (a mapped chunk simply starts that many bytes earlier in the file)
        c.fill = leftover.size();
        c.file_off = file_pos - leftover.size();
        c.err.clear();
        c.blocks.clear();
        c.consumed = 0;
        c.total = 0;
        bool full = false; // the chunk holds all the blocks it may
        // the framed bytes cross PCIe step by step, on the copy stream, while the next step is read: the device
        // buffer of this slot is free (the consumer released the slot only after the kernels that read it had finished)
        size_t sent = 0;
        const uint8_t *slot_was = d->d_comp_slot[k].p;
        bool h2d_ok = d->copy_stream && hipSetDevice(d->ctx->device) == hipSuccess &&
                      d->d_comp_slot[k].reserve(cap + INFLATE_IN_SLACK) == hipSuccess;
        // (a new allocation is zeroed once: the decoders' input windows read up to INFLATE_IN_SLACK bytes behind the chunk's
        // last payload -- never used, but not left to whatever the memory held before)
        if (h2d_ok && d->d_comp_slot[k].p != slot_was)
            h2d_ok = hipMemsetAsync(d->d_comp_slot[k].p, 0, d->d_comp_slot[k].bytes, d->copy_stream) == hipSuccess;
        {   // the context's device buffers were last read by the chunk four in front: behind its retirement on the context's stream
            bool wait_ev;
            {
                std::lock_guard<std::mutex> g(d->mu);
                wait_ev = d->retired_set[k];
            }
            if (h2d_ok && wait_ev) h2d_ok = hipStreamWaitEvent(d->copy_stream, d->retired_ev[k], 0) == hipSuccess;
        }
        auto send = [&]() {
            if (h2d_ok && c.err.empty() && c.consumed > sent) {
                const double ts = now_ms();
                // (from the mapping: the runtime pins the pages it copies from and lets them go inside the call -- it returns when
                // the bytes have left.  Registering the pieces ourselves and copying asynchronously crashed in the runtime when the
                // consumer thread's calls ran beside it (round 6, not pursued: DESIGN.md section 8))
                h2d_ok = (c.mapped ? hipMemcpyAsync(d->d_comp_slot[k].p + sent, c.h + sent, c.consumed - sent, hipMemcpyHostToDevice, d->copy_stream)
                                   : ngsq::pool_pinned_h2d(d->d_comp_slot[k].p + sent, c.pin, (size_t)(c.h - c.pin) + sent, c.consumed - sent, d->copy_stream)) == hipSuccess;
                sent = c.consumed;
                t_send += now_ms() - ts;
            }
        };
        auto frame = [&]() { // blocks of c.h[consumed, fill) that fit the chunk
            const double tf = now_ms();
            const size_t n0 = c.blocks.size();
            size_t used = 0;
            std::string err;
            if (!bgzf_split(c.h + c.consumed, c.fill - c.consumed, &c.blocks, &used, &c.total, &err, limit)) {
                c.err = path + ": " + err;
                return;
            }
            for (size_t i = n0; i < c.blocks.size(); i++) c.blocks[i].in_off += c.consumed;
            c.consumed += used;
            // bgzf_split stops in front of the block that would pass the limit, or of an incomplete block
            if (c.fill - c.consumed >= 18) {
                const uint8_t *h = c.h + c.consumed;
                const uint32_t xlen = bgzf_rd16(h + 10);
                if (c.fill - c.consumed >= 18 + (size_t)xlen) { // the whole header is there: which of the two was it?
                    uint32_t bs = 0;
                    for (size_t q = 12; q + 6 <= 12 + (size_t)xlen;) {
                        const uint32_t sl = bgzf_rd16(h + q + 2);
                        if (q + 4 + sl > 12 + (size_t)xlen) break; // corrupt: the next bgzf_split reports it
                        if (h[q] == 'B' && h[q + 1] == 'C' && sl == 2) bs = bgzf_rd16(h + q + 4) + 1;
                        q += 4 + sl;
                    }
                    if (bs && c.fill - c.consumed >= bs) full = true; // complete, yet not taken: the limit
                }
            }
            t_frame += now_ms() - tf;
        };
        while (c.err.empty() && !full && !eof && c.fill < cap) {
            // bytes this chunk still needs, by the ratio so far (unknown at first: one step, then look again)
            size_t want = (size_t)std::min<uint64_t>(STEP, limit / 2);
            if (ratio > 0) {
                const double need = (double)(limit - c.total) * ratio * 1.02 + 2 * 65536.0;
                const size_t have = c.fill - c.consumed;
                want = need > (double)have ? (size_t)(need - (double)have) : 65536;
            }
            want = std::min(want, cap - c.fill);
            want = (size_t)std::min<uint64_t>(want, d->pos_end - file_pos);
            want = std::min(want, ReadPool::PIECE * ReadPool::MAX_PIECES);
            const double tsp = now_ms();
            n_steps++;
            const size_t fill0 = c.fill;
            size_t avail = 0;
            bool short_read = false;
            if (c.mapped) {
                // nothing to read: the bytes are where they are; framed and sent STEP bytes at a time (the copies pin as they go)
                const size_t have = (size_t)std::min<uint64_t>(want, d->file_size > file_pos ? d->file_size - file_pos : 0);
                short_read = have < want;
                while (avail < have) {
                    avail = std::min(have, avail + STEP);
                    c.fill = fill0 + avail;
                    if (c.err.empty() && !full) {
                        frame();
                        send();
                    }
                }
            } else {
                pool.start(c.h + c.fill, file_pos, want, cold);
                t_spawn += now_ms() - tsp;
                // what has arrived is framed and sent while the rest is read, STEP bytes at a time
                bool all = false;
                while (!all) {
                    const double tj = now_ms();
                    pool.wait_prefix(std::min(want, avail + STEP), &avail, &all, &short_read);
                    t_join += now_ms() - tj;
                    c.fill = fill0 + avail;
                    if (c.err.empty() && !full) {
                        frame();
                        send();
                    }
                }
                if (pool.bad) c.err = "read error on " + path;
                // (without O_DIRECT: below 1.2 GB/s per read thread the bytes did not come out of the page cache -- larger pieces then)
                if (fd_direct < 0 && avail >= ((size_t)32 << 20)) cold = (double)avail / ((now_ms() - tsp) * 1e-3) < 1.2e9 * NT;
            }
            if (avail < want || short_read) eof = true;
            file_pos += avail;
            if (file_pos >= d->pos_end) eof = true;
            if (c.total) ratio = (double)c.consumed / (double)c.total;
        }
        if (c.err.empty() && !full) frame();
        send();
        if (c.total) ratio = (double)c.consumed / (double)c.total;
        const double tr1 = now_ms();
        leftover.assign(c.h + c.consumed, c.h + c.fill);
        if (eof && range_cut && !full) leftover.clear(); // the block the range was cut in: the next shard's
        {   // file offset of every block (the blocks of a chunk are contiguous from h[0])
            c.coff.resize(c.blocks.size());
            uint64_t start = 0;
            for (size_t k = 0; k < c.blocks.size(); k++) {
                c.coff[k] = c.file_off + start;
                start = c.blocks[k].in_off + c.blocks[k].in_len + 8;
            }
        }
        if (c.err.empty() && c.blocks.empty() && !(eof && leftover.empty())) {
            if (eof && !leftover.empty()) c.err = path + ": truncated BGZF block at end of file";
            else if (!eof) c.err = path + ": BGZF block does not fit the ingest buffer";
        }
        c.last = !c.err.empty() || (eof && leftover.empty());
        const bool last = c.last;
        d->h2d_issued[k] = false;
        if (c.err.empty() && !c.blocks.empty() && h2d_ok && sent == c.consumed) {
            // the block table and the blocks' file offsets follow the bytes on the same stream (until round 4 the consumer sent
            // them with two copy kernels on the INFLATE stream, reading this memory across PCIe beside the DMA: 0.14 ms each)
            const size_t nb = c.blocks.size(), tb = nb * sizeof(BgzfBlock), ob = nb * sizeof(uint64_t);
            bool ok = c.tab.reserve(tb + ob) == hipSuccess && d->d_blocks_s[k].reserve(nb) == hipSuccess && d->d_coff_s[k].reserve(nb) == hipSuccess;
            if (ok) {
                memcpy(c.tab.h, c.blocks.data(), tb);
                memcpy(static_cast<uint8_t *>(c.tab.h) + tb, c.coff.data(), ob);
                ok = hipMemcpyAsync(d->d_blocks_s[k].p, c.tab.h, tb, hipMemcpyHostToDevice, d->copy_stream) == hipSuccess &&
                     hipMemcpyAsync(d->d_coff_s[k].p, static_cast<uint8_t *>(c.tab.h) + tb, ob, hipMemcpyHostToDevice, d->copy_stream) == hipSuccess;
            }
            // (the INFLATE_IN_SLACK bytes behind the chunk's last payload must be READABLE -- the decoders' input windows reach past a
            // block's end, into the next block's bytes everywhere but here -- not zero: no memset per chunk)
            ok = ok && hipEventRecord(d->h2d_done[k], d->copy_stream) == hipSuccess;
            d->h2d_issued[k] = ok; // on failure the consumer copies on its own stream (and reports errors)
        }
        if (trace_on())
            fprintf(stderr, "[ingest] +%.1f ms reader: slot %d, %.1f MB read and %zu blocks framed in %.1f ms (%d steps: starting the read threads %.1f, framing %.1f, "
                            "queueing the copies %.1f, waiting for the reads %.1f ms), %.1f MB left over\n",
                    now_ms() - d->t_start, k, c.fill / 1e6, c.blocks.size(), tr1 - tr0, n_steps, t_spawn, t_frame, t_send, t_join, leftover.size() / 1e6);
        {
            std::lock_guard<std::mutex> g(d->mu);
            c.ready = true;
        }
        d->cv.notify_all();
        if (last) return;
    }
}

// Offsets of every complete record of d->raw[0, raw_len) whose chain starts at `first` -> d_rec_off; sets
// d->tail_off to the offset of the cut record (or raw_len).  DESIGN.md section 9 "Record boundaries".
int index_records(ngsq_bam *b, DeviceIngest *d, uint64_t first, uint64_t *out_total, uint64_t *out_entry) {
    hipStream_t st = d->ctx->stream;
    const uint32_t n_seg = (uint32_t)((d->raw_len + REC_SEGMENT - 1) / REC_SEGMENT);
    const bool find = first == FIND_ENTRY;
    if (find) first = 0;
    *out_total = 0;
    *out_entry = find ? d->raw_len : first;
    if (!n_seg) return NGSQ_OK;
    d->stats.chunks += 1;
    d->stats.segments += n_seg;
    // (the candidate table is written straight into pinned host memory and the segments' verdicts are read from it: see
    // k_copy_words in bam_device.hip for why nothing here is a hipMemcpyAsync)
    BHIP(d->h_cand.reserve((size_t)n_seg * REC_CANDIDATES * sizeof(RecCandidate)));
    BHIP(d->h_seg.reserve((size_t)n_seg * 2 * sizeof(uint64_t)));
    BHIP(d->h_small.reserve(64 * sizeof(uint64_t)));
    const uint32_t n_pieces = (uint32_t)((d->raw_len + REC_PIECE - 1) / REC_PIECE);
    BHIP(d->d_pieces.reserve((size_t)n_seg * REC_CANDIDATES));
    {
        KernelTimer kt(d->ctx, K_REC_INDEX, d->raw_len);
        BHIP(launch_rec_candidates(d->raw, d->raw_len, first, n_seg, (int32_t)b->ref_names.size(), static_cast<RecCandidate *>(d->h_cand.dev),
                                   d->d_pieces.p, d->d_small.p, st));
    }
    BHIP(hipStreamSynchronize(st));
    const RecCandidate *const cand = static_cast<const RecCandidate *>(d->h_cand.h);
    // [n_seg] index of each segment's first record | [n_seg] its chain: 32-bit words (a chunk holds fewer than 2^32 records; 8 bytes
    // per segment cross PCIe to the device, read by a copy kernel)
    uint32_t *const seg = static_cast<uint32_t *>(d->h_seg.h);
    memset(seg, 0, (size_t)n_seg * 2 * sizeof(uint32_t));
    uint32_t *chosen = seg + n_seg;
    uint64_t *const h_small = static_cast<uint64_t *>(d->h_small.h);
    uint64_t *const h_small_dev = static_cast<uint64_t *>(d->h_small.dev);
    if (find) { // the first plausible chain of the view: an assumption (ngsq_bam_shard_verify confirms or corrects it)
        // -- preferably one whose landing offset is itself a candidate of the segment it lands in
        first = d->raw_len;
        bool any = false;
        for (size_t xi = 0; xi < (size_t)n_seg * REC_CANDIDATES; xi++) {
            const RecCandidate &x = cand[xi];
            if (!x.valid) continue;
            if (!any) first = x.start; // fallback: the very first
            any = true;
            const uint64_t sl = x.landing / REC_SEGMENT;
            bool agrees = sl >= n_seg;
            for (uint32_t k = 0; k < REC_CANDIDATES && !agrees; k++) {
                const RecCandidate &y = cand[(size_t)sl * REC_CANDIDATES + k];
                agrees = y.valid && y.start == x.landing;
            }
            if (agrees) {
                first = x.start;
                break;
            }
        }
        *out_entry = first;
    }
    uint64_t cur = first, total_rec = 0;
    bool chain_broken = false; // a segment's walk stopped at an invalid record
    for (uint32_t s = 0; s < n_seg; s++) {
        const uint64_t s0 = (uint64_t)s * REC_SEGMENT, s1 = std::min<uint64_t>(s0 + REC_SEGMENT, d->raw_len);
        chosen[s] = REC_NO_CHAIN;
        seg[s] = (uint32_t)total_rec;
        if (cur >= s1) continue;
        const RecCandidate *c = nullptr;
        for (uint32_t k = 0; k < REC_CANDIDATES; k++) {
            const RecCandidate &x = cand[(size_t)s * REC_CANDIDATES + k];
            if (x.valid && x.start == cur) {
                c = &x;
                chosen[s] = k;
            }
        }
        RecCandidate one{};
        if (!c) {
            // not in the table: one thread walks the segment with the host reader's rules (its piece table replaces candidate 0's)
            BHIP(launch_walk_one(d->raw, d->raw_len, cur, s0, s1, reinterpret_cast<RecCandidate *>(h_small_dev + 8),
                                 d->d_pieces.p + (size_t)s * REC_CANDIDATES, st));
            BHIP(hipStreamSynchronize(st));
            memcpy(&one, h_small + 8, sizeof one);
            c = &one; // an invalid record stops the walk: k_rec_offsets reports its index
            chosen[s] = 0;
            d->stats.walk_one += 1;
        }
        total_rec += c->count;
        // an invalid record ends the chain: the later segments get no entry (k_rec_offsets then reports
        // the record's index from the piece it lies in)
        if (!c->valid) chain_broken = true;
        cur = c->valid ? c->landing : d->raw_len;
    }
    d->tail_off = std::min(cur, d->raw_len);
    BHIP(d->d_rec_off.reserve(total_rec + 1));
    if (total_rec >> 32) return ngsq_bam_fail(NGSQ_ERR_LIMIT, "%s: more than 2^32 records in one chunk of the ingest", b->path.c_str());
    // (k_rec_offsets reads the verdicts where they are, in pinned host memory: 8 bytes per segment, 64 contiguous bytes per
    // wave and array.  Until late in round 4 a copy kernel brought them over first -- one launch per chunk that waited 90 us on
    // average, up to 1.3 ms, for a wave slot beside the decoders: 20 ms per 24 M-record scan for 256 KB)
    const uint32_t *const seg_dev = static_cast<const uint32_t *>(d->h_seg.dev);
    {
        KernelTimer kt(d->ctx, K_REC_INDEX, 0);
        BHIP(launch_rec_offsets(d->raw, d->raw_len, n_pieces, seg_dev + n_seg, seg_dev,
                                d->d_pieces.p, d->d_rec_off.p, d->d_small.p, st));
    }
    // (an invalid record k_rec_offsets alone notices: its index stays in the device word and the first batch of the chunk
    // reports it -- k_rec_fixed looks before it touches an offset; no copy and no wait here)
    d->chunk_first_record = b->n_read;
    if (chain_broken) { // the host's walk met one: say which
        BHIP(launch_copy_words(h_small_dev, d->d_small.p + W_BAD, sizeof(unsigned long long), st));
        BHIP(hipStreamSynchronize(st));
        if (h_small[0] != ~0ull)
            return ngsq_bam_fail(NGSQ_ERR_INVALID_ARGUMENT, "%s: malformed record %llu", b->path.c_str(),
                                 (unsigned long long)(b->n_read + h_small[0]));
    }
    *out_total = total_rec;
    return NGSQ_OK;
}

// Queue the inflate (+ CRC check) of the framed chunk in host slot `slot` on the inflate stream, into raw buffer
// `slot` behind its headroom.  The caller has made sure the reader thread marked the slot ready.
int issue_inflate(ngsq_bam *b, DeviceIngest *d, uint64_t j) {
    const int k = (int)(j % DeviceIngest::NC), rs = (int)(j % DeviceIngest::NR);
    DeviceIngest::HostChunk &c = d->hc[k];
    DeviceIngest::Pending &p = d->pend[k];
    p.rslot = rs;
    p.issued = true;
    p.err = c.err;
    p.last = c.last;
    p.n_blk = c.blocks.size();
    p.consumed = c.consumed;
    p.total = c.total;
    p.next_coff = c.file_off + c.consumed;
    p.blocks.clear();
    p.coff.clear();
    d->chunks_issued = j + 1;
    if (!p.err.empty() || !p.n_blk) return NGSQ_OK;
    p.blocks = c.blocks;
    p.coff = c.coff;
    const int si = d->inflate_streams > 1 ? (int)(j & 1) : 0;
    hipStream_t sb = d->inf_stream[si];
    BHIP(d->d_status_s[k].reserve(p.n_blk + 1));
    if (p.status_cap < p.n_blk) {
        const size_t cap = p.n_blk + p.n_blk / 4 + 1024;
        BHIP(p.pin.reserve((cap + 2) * (sizeof(uint32_t) + sizeof(BgzfBlock) + sizeof(uint64_t))));
        p.status_cap = cap;
        p.status = static_cast<uint32_t *>(p.pin.h);
        p.pin_blocks = reinterpret_cast<BgzfBlock *>(p.status + cap + (cap & 1)); // 8-byte aligned
        p.pin_coff = reinterpret_cast<uint64_t *>(p.pin_blocks + cap);
    }
    auto dev_of = [&](const void *host) { return static_cast<uint8_t *>(p.pin.dev) + (static_cast<const uint8_t *>(host) - static_cast<const uint8_t *>(p.pin.h)); };
    if (d->h2d_issued[k]) { // the bytes and the tables are on their way: the reader thread issued the copies
        if (trace_on()) {
            const double tw = now_ms();
            BHIP(hipEventSynchronize(d->h2d_done[k]));
            fprintf(stderr, "[ingest] waited %.1f ms for the host-to-device copy of context %d\n", now_ms() - tw, k);
        }
        BHIP(hipStreamWaitEvent(sb, d->h2d_done[k], 0));
    } else {
        if (d->retired_set[k]) BHIP(hipStreamWaitEvent(sb, d->retired_ev[k], 0));
        BHIP(d->d_blocks_s[k].reserve(p.n_blk));
        BHIP(d->d_coff_s[k].reserve(p.n_blk));
        memcpy(p.pin_blocks, p.blocks.data(), p.n_blk * sizeof(BgzfBlock));
        memcpy(p.pin_coff, p.coff.data(), p.n_blk * sizeof(uint64_t));
        {
            const uint8_t *slot_was = d->d_comp_slot[k].p;
            BHIP(d->d_comp_slot[k].reserve(p.consumed + INFLATE_IN_SLACK));
            if (d->d_comp_slot[k].p != slot_was) BHIP(hipMemsetAsync(d->d_comp_slot[k].p, 0, d->d_comp_slot[k].bytes, sb));
        }
        if (c.mapped) BHIP(hipMemcpyAsync(d->d_comp_slot[k].p, c.h, p.consumed, hipMemcpyHostToDevice, sb));
        else BHIP(ngsq::pool_pinned_h2d(d->d_comp_slot[k].p, c.pin, (size_t)(c.h - c.pin), p.consumed, sb));
        BHIP(launch_copy_words(d->d_blocks_s[k].p, dev_of(p.pin_blocks), p.n_blk * sizeof(BgzfBlock), sb));
        BHIP(launch_copy_words(d->d_coff_s[k].p, dev_of(p.pin_coff), p.n_blk * sizeof(uint64_t), sb));
    }
    if (d->raw_free_set[rs]) BHIP(hipStreamWaitEvent(sb, d->raw_free[rs], 0)); // the chunk three in front has left this buffer
    uint8_t *out = d->d_rawb[rs].p + CARRY_MAX;
    {   // algorithmic bytes of the inflate: compressed bytes read + inflated bytes written
        KernelTimer kt(d->ctx, K_INFLATE, p.consumed + p.total, sb);
        // (the decoders' block counter: a word per inflate stream that is never reset -- no memset per launch)
        BHIP(launch_bgzf_inflate(d->d_comp_slot[k].p, d->d_blocks_s[k].p, (uint32_t)p.n_blk, out, d->d_status_s[k].p,
                                 reinterpret_cast<uint32_t *>(d->d_small.p + (si ? W_INF1 : W_INF0)), false, sb, &d->inf_ctr_base[si]));
    }
    {   // (the decoders' verdicts reach the host through this kernel: it writes every block's final status into pinned memory)
        KernelTimer kt(d->ctx, K_INFLATE_CRC, p.total, sb);
        BHIP(launch_bgzf_crc(d->d_blocks_s[k].p, (uint32_t)p.n_blk, out, d->d_status_s[k].p, reinterpret_cast<uint32_t *>(dev_of(p.status)), sb));
    }
    BHIP(hipEventRecord(d->inf_done[k], sb));
    (void)b;
    return NGSQ_OK;
}

// The next chunk: wait for its inflate (queued one call earlier, while the chunk before it was parsed and scanned),
// move the record the previous chunk ended in to its front, queue the inflate of the chunk after it, index its records.
int load_chunk(ngsq_bam *b, DeviceIngest *d) {
    hipStream_t st = d->ctx->stream;
    const double t0 = now_ms();
    static thread_local double last_end = 0;
    const double batches_ms = last_end ? t0 - last_end : 0.0; // time the caller spent on the previous chunk's batches
    const uint64_t j = d->chunks_loaded; // this chunk
    const int slot = (int)(j % DeviceIngest::NC), rs = (int)(j % DeviceIngest::NR);
    DeviceIngest::HostChunk &c = d->hc[slot];
    DeviceIngest::Pending &p = d->pend[slot];
    // ---- 1. this chunk's inflate: queued two calls earlier unless the reader thread was late (or these are the first)
    if (!p.issued) {
        {
            std::unique_lock<std::mutex> g(d->mu);
            d->cv.wait(g, [&] { return c.ready; });
        }
        const int rc = issue_inflate(b, d, j);
        if (rc) return rc;
    }
    const double t1 = now_ms();
    // ---- 2. the cut record at the end of the previous chunk moves in front of this chunk's data (into the headroom of this
    // chunk's buffer: the inflate, which may still be running, writes behind it)
    const uint64_t carry = d->raw_len - d->tail_off;
    if (carry > CARRY_MAX)
        return ngsq_bam_fail(NGSQ_ERR_UNSUPPORTED, "%s: a record larger than the device ingest carry space (%llu MiB)", b->path.c_str(),
                             (unsigned long long)(CARRY_MAX >> 20));
    uint8_t *const view = d->d_rawb[rs].p + CARRY_MAX - carry;
    if (carry) BHIP(launch_copy_bytes(view, d->raw + d->tail_off, carry, st)); // (a kernel: a hipMemcpyAsync queues behind the reader's DMA)
    // everything that reads the previous chunk's buffers has been queued on the context's stream by now: the chunk is
    // retired -- its raw buffer may be inflated into again, its context goes back to the reader thread
    if (j) {
        const int kp = (int)((j - 1) % DeviceIngest::NC), rp = (int)((j - 1) % DeviceIngest::NR);
        BHIP(hipEventRecord(d->raw_free[rp], st));
        d->raw_free_set[rp] = true;
        BHIP(hipEventRecord(d->retired_ev[kp], st));
        d->h2d_issued[kp] = false;
        d->pend[kp].issued = false;
        {
            std::lock_guard<std::mutex> g(d->mu);
            d->retired_set[kp] = true;
            d->hc[kp].ready = false;
        }
        d->cv.notify_all();
    }
    d->chunks_loaded = j + 1;
    // ---- 3. the inflates of the two chunks after this one are queued BEFORE this thread waits for this chunk's: the
    // decoders go from one chunk to the next without waiting for the host, and those of chunk j + 1 (the other inflate
    // stream) take the wave slots this chunk's leave when its last blocks are in work
    auto issue_ahead = [&]() -> int {
        if (p.last) return NGSQ_OK;
        while (d->chunks_issued <= j + (uint64_t)d->inflate_ahead) {
            const uint64_t q = d->chunks_issued;
            const int kq = (int)(q % DeviceIngest::NC);
            if (q > j && d->pend[(q - 1) % DeviceIngest::NC].last) break; // (nothing behind the range's last chunk)
            bool ready;
            {
                std::lock_guard<std::mutex> g(d->mu);
                ready = d->hc[kq].ready && !d->pend[kq].issued;
            }
            if (!ready) break;
            const int rc = issue_inflate(b, d, q);
            if (rc) return rc;
        }
        return NGSQ_OK;
    };
    {
        const int rc = issue_ahead();
        if (rc) return rc;
    }
    // ---- 4. this chunk's inflate
    if (!p.err.empty()) return ngsq_bam_fail(NGSQ_ERR_INVALID_ARGUMENT, "%s", p.err.c_str());
    if (p.n_blk) {
        BHIP(hipEventSynchronize(d->inf_done[slot]));
        for (size_t k = 0; k < p.n_blk; k++)
            if (p.status[k] != INF_OK)
                return ngsq_bam_fail(NGSQ_ERR_INVALID_ARGUMENT, "%s: BGZF block %llu: %s", b->path.c_str(),
                                     (unsigned long long)(d->blocks_done + k), inflate_status_text(p.status[k]));
        d->blocks_done += p.n_blk;
    }
    const double t3 = now_ms();
    {   // (the reader thread was late a moment ago: look again)
        const int rc = issue_ahead();
        if (rc) return rc;
    }
    d->raw = view;
    d->raw_len = carry + p.total;
    d->n_rec = d->cursor = 0;
    d->tail_off = 0;
    d->cur_slot = slot;
    d->carry_len = carry;
    d->carry_id = d->tail_id;
    // view offset -> virtual offset, by this chunk's block table (blocks without data hold no byte)
    auto voffset_of = [&](uint64_t o) -> uint64_t {
        if (o < carry) return d->carry_id;
        const uint64_t u = o - carry;
        size_t lo_k = 0, hi_k = p.blocks.size();
        if (!hi_k || u >= p.total) return p.next_coff << 16; // behind the chunk's last byte: the next block's first
        while (hi_k - lo_k > 1) {
            const size_t mid = (lo_k + hi_k) / 2;
            if (p.blocks[mid].out_off <= u) lo_k = mid;
            else hi_k = mid;
        }
        return p.coff[lo_k] << 16 | (u - p.blocks[lo_k].out_off);
    };
    // records that start below `limit` are this range's own: everything in front of the first block at or behind pos_hi
    uint64_t limit = d->raw_len;
    if (d->boundary_passed) {
        limit = carry && d->carry_owned ? 1 : 0;
    } else {
        const size_t kb = (size_t)(std::lower_bound(p.coff.begin(), p.coff.end(), d->pos_hi) - p.coff.begin());
        if (kb < p.coff.size()) {
            limit = carry + p.blocks[kb].out_off;
            d->boundary_passed = true;
        } else if (p.last && d->sharded && d->pos_hi < d->file_size) {
            d->boundary_passed = true; // the range ended before a block of the next shard was seen (cut inside one)
        }
    }
    uint64_t first = 0;
    if (d->first_chunk) {
        if (d->entry_mode == DeviceIngest::ENTRY_HEADER) {
            first = b->header_bytes;
            if (d->raw_len < first) { // nothing but header bytes so far (the header may span the first chunks): drop them
                if (p.last) return ngsq_bam_fail(NGSQ_ERR_INVALID_ARGUMENT, "%s: the file ends inside the BAM header", b->path.c_str());
                if (limit < d->raw_len) return ngsq_bam_fail(NGSQ_ERR_UNSUPPORTED, "%s: the BAM header is larger than the first shard", b->path.c_str());
                b->header_bytes -= d->raw_len;
                d->tail_off = d->raw_len;
                return NGSQ_OK;
            }
        } else if (d->entry_mode == DeviceIngest::ENTRY_KNOWN) {
            first = d->entry_uoff;
            if (first > d->raw_len) return ngsq_bam_fail(NGSQ_ERR_INVALID_ARGUMENT, "%s: virtual offset outside its block", b->path.c_str());
        } else {
            first = FIND_ENTRY;
        }
        d->first_chunk = false;
    }
    // ---- 4. record index
    uint64_t total_rec = 0, entry = first;
    {
        const int rc = index_records(b, d, first, &total_rec, &entry);
        if (rc) return rc;
    }
    if (!d->have_begin) {
        d->begin_voffset = voffset_of(entry);
        d->have_begin = true;
    }
    d->tail_id = d->tail_off < d->raw_len ? voffset_of(d->tail_off) : 0;
    // ---- 5. which of them are this range's
    uint64_t own = total_rec;
    if (limit < d->raw_len && total_rec) {
        BHIP(launch_count_below_u64(d->d_rec_off.p, total_rec, limit, static_cast<unsigned long long *>(d->h_small.dev) + 12, st));
        BHIP(hipStreamSynchronize(st));
        own = static_cast<const uint64_t *>(d->h_small.h)[12];
    }
    d->n_rec = own;
    const bool cut = d->tail_off < d->raw_len; // the chunk ends inside a record
    if (own < total_rec) { // the first record of the next range
        BHIP(launch_copy_words(static_cast<uint64_t *>(d->h_small.dev) + 13, d->d_rec_off.p + own, sizeof(uint64_t), st));
        BHIP(hipStreamSynchronize(st));
        d->end_voffset = voffset_of(static_cast<const uint64_t *>(d->h_small.h)[13]);
        d->last_chunk = true;
    } else if (d->boundary_passed && cut && d->tail_off >= limit) { // the cut record is the next range's first
        d->end_voffset = d->tail_id;
        d->last_chunk = true;
    } else if (p.last) {
        if (cut)
            return ngsq_bam_fail(d->pos_end >= d->file_size ? NGSQ_ERR_INVALID_ARGUMENT : NGSQ_ERR_UNSUPPORTED,
                                 d->pos_end >= d->file_size ? "%s: truncated record" : "%s: a record reaches more than 17 MiB of BGZF blocks past its shard",
                                 b->path.c_str());
        d->end_voffset = p.next_coff << 16; // no record follows in what was read (the end of the file: file size << 16)
        d->last_chunk = true;
    } else {
        d->carry_owned = !d->boundary_passed || d->tail_off < limit;
    }
    if (d->last_chunk && !p.last) { // nothing behind this chunk is wanted: the reader thread may stop
        {
            std::lock_guard<std::mutex> g(d->mu);
            d->stop = true;
        }
        d->cv.notify_all();
    }
    if (trace_on())
        fprintf(stderr, "[ingest] +%.1f ms chunk: %zu blocks, %.1f MB -> %.1f MB, %llu records | batches of the previous chunk %.1f ms, "
                        "wait for reader %.1f ms, wait for inflate %.1f ms, index %.1f ms\n",
                now_ms() - d->t_start, p.n_blk, p.consumed / 1e6, p.total / 1e6, (unsigned long long)total_rec, batches_ms, t1 - t0, t3 - t1, now_ms() - t3);
    last_end = now_ms();
    return NGSQ_OK;
}

} // namespace

extern "C" {

} // extern "C" (reopened below)

// ---- sharded mode --------------------------------------------------------------------------------

namespace {

// first BGZF block start at or after `from`: the gzip/BGZF header bytes, confirmed by following BSIZE
// through the next blocks (compressed data can contain the magic by chance)
int find_block_start(FILE *f, uint64_t from, uint64_t file_size, uint64_t *out, std::string *err) {
    if (from == 0 || from >= file_size) {
        *out = std::min(from, file_size);
        return 0;
    }
    std::vector<uint8_t> buf((size_t)std::min<uint64_t>(file_size - from, (uint64_t)5 << 16));
    if (fseeko(f, (off_t)from, SEEK_SET) != 0 || fread(buf.data(), 1, buf.size(), f) != buf.size()) {
        *err = "read error while looking for a BGZF block boundary";
        return -1;
    }
    auto header_at = [&](size_t p, uint32_t *bsize) {
        if (p + 18 > buf.size()) return false;
        const uint8_t *c = buf.data() + p;
        if (c[0] != 31 || c[1] != 139 || c[2] != 8 || !(c[3] & 4)) return false;
        const uint32_t xlen = bgzf_rd16(c + 10);
        if (p + 12 + xlen > buf.size()) return false;
        for (size_t q = p + 12; q + 4 <= p + 12 + xlen;) {
            const uint32_t slen = bgzf_rd16(buf.data() + q + 2);
            if (q + 4 + slen > p + 12 + xlen) return false;
            if (buf[q] == 'B' && buf[q + 1] == 'C' && slen == 2) {
                *bsize = bgzf_rd16(buf.data() + q + 4) + 1;
                return *bsize >= 12 + xlen + 8;
            }
            q += 4 + slen;
        }
        return false;
    };
    for (size_t p = 0; p + 18 <= buf.size() && p < ((size_t)1 << 16) + 18; p++) {
        uint32_t bs = 0;
        if (!header_at(p, &bs)) continue;
        // follow the chain for up to three more blocks (or to the end of the file)
        size_t q = p + bs;
        int ok = 1;
        for (int k = 0; k < 3 && ok; k++) {
            if (from + q == file_size) break;
            uint32_t b2 = 0;
            if (q + 18 > buf.size()) break; // ran out of look-ahead: accept
            if (!header_at(q, &b2)) ok = 0;
            q += b2;
        }
        if (ok) {
            *out = from + p;
            return 0;
        }
    }
    if (from + buf.size() >= file_size) { // `from` lies inside the last block: the next boundary is the end of the file
        *out = file_size;
        return 0;
    }
    *err = "no BGZF block boundary found";
    return -1;
}

// Streams, events, the reader thread and the two raw buffers of an ingest over d's byte range.
int start_ingest(ngsq_bam *b, ngsq_ctx *c, DeviceIngest *d) {
    // Chunk size: every buffer of the pipeline (three raw buffers, four pinned and four device buffers for the compressed
    // bytes) is proportional to it and costs ~175 ms of allocation and pinning per GiB of chunk before the first record
    // is seen (the first file of a process; later ones take the blocks from the cache).  Measured in round 5 on a 100 M-record
    // aligner-style file (12.5 GB; median of five scans, profiles/r05_chunk_size.txt): 256 MiB 238 M records/s, 512 MiB 273-275 M,
    // 1 GiB 252 M -- the 1 GiB chunks round 2 chose for files beyond 16 GiB (their inflate launch runs fuller BY ITSELF) cost the
    // 150 M-record file of the bench 9 %.  NGSQ_INGEST_RAW_MB overrides.
    const uint64_t sz = d->pos_end > d->pos_lo ? d->pos_end - d->pos_lo : 0;
    const size_t dflt_mb = sz > ((uint64_t)4 << 30) ? 512 : 256;
    d->raw_cap = env_mb("NGSQ_INGEST_RAW_MB", dflt_mb);
    d->comp_chunk = std::max<size_t>(d->raw_cap / 4, (size_t)1 << 17);
    // (the two pinned buffers are allocated by the reader thread, on the device's NUMA node)
    // (streams and events come from the process's cache when an earlier ingest left them there: mem_pool.h)
    BHIP(ngsq::pool_stream_get(false, &d->copy_stream));
    {   // the inflate of the NEXT chunk runs beside the parse and the scan of this one, and it takes every
        // wave slot its LDS allows: give it the lowest priority so that the short kernels of the context's stream get
        // the slots its decoders free, instead of queueing behind all of them
        const char *e = getenv("NGSQ_INFLATE_PRIORITY");
        d->inf_low_priority = !(e && atoi(e) == 0); // =0: normal priority (A/B measurements)
        // (keeping the decoders off 8 / 16 / 32 compute units with a CU mask, so that the CRC and the parse kernels always find free
        // ones, was measured in round 5: 269 -> 210 / 210 / 204 M records/s on the plain file, 250 -> 194 / 198 / 200 M on the
        // aligner-style one -- masked streams cannot carry the low priority, and the decoders lose more than the others gain)
        for (auto &q : d->inf_stream) BHIP(ngsq::pool_stream_get(d->inf_low_priority, &q));
    }
    for (auto &e : d->h2d_done) BHIP(ngsq::pool_event_get(&e));
    for (auto &e : d->inf_done) BHIP(ngsq::pool_event_get(&e));
    for (auto &e : d->raw_free) BHIP(ngsq::pool_event_get(&e));
    for (auto &e : d->retired_ev) BHIP(ngsq::pool_event_get(&e));
    if (const char *e = getenv("NGSQ_INFLATE_AHEAD")) d->inflate_ahead = std::max(1, std::min(2, atoi(e)));
    if (const char *e = getenv("NGSQ_INFLATE_STREAMS")) d->inflate_streams = std::max(1, std::min(2, atoi(e)));
    // the reader starts pinning and reading at once; the two raw buffers are allocated meanwhile
    if (trace_on()) fprintf(stderr, "[ingest] +%.1f ms streams and events created\n", now_ms() - d->t_start);
    d->reader = std::thread(reader_main, d, b->path);
    for (auto &r : d->d_rawb) BHIP(r.reserve(CARRY_MAX + d->raw_cap + 64)); // headroom for the carried record | one chunk's inflated bytes
    if (trace_on()) fprintf(stderr, "[ingest] +%.1f ms raw buffers reserved\n", now_ms() - d->t_start);
    BHIP(d->d_small.reserve(REC_WORK_WORDS));
    {
        unsigned long long w[REC_WORK_WORDS];
        rec_work_init(w);
        BHIP(hipMemcpy(d->d_small.p, w, sizeof w, hipMemcpyHostToDevice));
        d->inf_ctr_base[0] = d->inf_ctr_base[1] = 0; // (the decoders' counters are among those words)
    }
    // the host side of this handle is done: release its buffers
    std::vector<uint8_t>().swap(b->comp);
    std::vector<uint8_t>().swap(b->data);
    (void)c;
    if (trace_on()) fprintf(stderr, "[ingest] +%.1f ms ingest started\n", now_ms() - d->t_start);
    return NGSQ_OK;
}

int open_ingest(ngsq_bam *b, ngsq_ctx *c, DeviceIngest **out) {
    DeviceIngest *d = new DeviceIngest();
    d->ctx = c;
    d->device = c->device;
    d->f = fopen(b->path.c_str(), "rb");
    if (!d->f) {
        delete d;
        return ngsq_bam_fail(NGSQ_ERR_INVALID_ARGUMENT, "opening BAM file: %s", b->path.c_str());
    }
    struct stat fst;
    if (fstat(fileno(d->f), &fst) != 0) {
        delete d;
        return ngsq_bam_fail(NGSQ_ERR_INVALID_ARGUMENT, "cannot stat %s", b->path.c_str());
    }
    d->file_size = (uint64_t)fst.st_size;
    d->pos_lo = 0;
    d->pos_hi = d->pos_end = d->file_size;
    *out = d;
    return NGSQ_OK;
}

unsigned long long sort_key(unsigned long long ref_pos) { // refID << 32 | pos -> ascending in a coordinate-sorted file
    const int32_t ref = (int32_t)(ref_pos >> 32), pos = (int32_t)(uint32_t)ref_pos;
    return ref < 0 ? ~0ull : (unsigned long long)(uint32_t)ref << 32 | (uint32_t)(pos + 1);
}

} // namespace

// ---- one file, several GPUs (include/ngsq_bam.h "sharded device ingest") ---------------------------------------
extern "C" int ngsq_bam_shard_begin(ngsq_bam *b, ngsq_ctx *c, uint32_t shard, uint32_t n_shards, uint64_t begin_voffset) {
    if (!b || !c || !n_shards || shard >= n_shards) return ngsq_bam_fail(NGSQ_ERR_INVALID_ARGUMENT, "bad argument");
    if (b->host_mode) return ngsq_bam_fail(NGSQ_ERR_STATE, "%s: the reader is in host ingest mode", b->path.c_str());
    BHIP(hipSetDevice(c->device));
    uint64_t header_bytes = b->header_bytes;
    if (b->dev) { // again, from a confirmed first record (ngsq_bam_shard_verify): the previous scan's state goes
        if (!b->dev->sharded || b->dev->ctx != c) return ngsq_bam_fail(NGSQ_ERR_STATE, "%s: the reader is already in use", b->path.c_str());
        header_bytes = b->dev->header_bytes0;
        b->dev_free(b->dev);
        b->dev = nullptr;
    }
    DeviceIngest *d = nullptr;
    int rc = open_ingest(b, c, &d);
    if (rc) return rc;
    b->dev = d;
    b->dev_free = free_ingest;
    b->n_read = 0;
    b->header_bytes = header_bytes;
    d->header_bytes0 = header_bytes;
    d->sharded = true;
    std::string err;
    uint64_t lo = 0, hi = d->file_size;
    if (find_block_start(d->f, d->file_size / n_shards * shard, d->file_size, &lo, &err) ||
        find_block_start(d->f, shard + 1 == n_shards ? d->file_size : d->file_size / n_shards * (shard + 1), d->file_size, &hi, &err))
        return ngsq_bam_fail(NGSQ_ERR_INVALID_ARGUMENT, "%s: %s", b->path.c_str(), err.c_str());
    d->pos_lo = lo;
    d->pos_hi = hi;
    d->pos_end = std::min(d->file_size, hi + SHARD_EXTRA);
    d->entry_mode = shard == 0 ? DeviceIngest::ENTRY_HEADER : DeviceIngest::ENTRY_FIND;
    if (begin_voffset) {
        if (shard == 0) return ngsq_bam_fail(NGSQ_ERR_INVALID_ARGUMENT, "shard 0 starts behind the header");
        const uint64_t coff = begin_voffset >> 16;
        if (coff < lo || coff > d->file_size)
            return ngsq_bam_fail(NGSQ_ERR_INVALID_ARGUMENT, "%s: virtual offset %llu lies in front of shard %u", b->path.c_str(),
                                 (unsigned long long)begin_voffset, shard);
        d->pos_lo = coff; // a block start, by the neighbour's word: the block chain is followed from there
        d->pos_end = std::max(d->pos_end, std::min(d->file_size, coff + SHARD_EXTRA));
        d->entry_mode = DeviceIngest::ENTRY_KNOWN;
        d->entry_uoff = begin_voffset & 0xFFFFu;
        if (coff >= hi) d->boundary_passed = true; // nothing of this shard's own (the first record lies behind it)
    }
    // the workers of one node share its cores: quota less one driving thread and one framing thread per worker
    if (const char *e = getenv("NGSQ_READER_THREADS")) d->reader_threads = atoi(e);
    if (d->reader_threads <= 0) d->reader_threads = std::max(2, (effective_cores() - 2 * (int)n_shards) / (int)n_shards);
    return start_ingest(b, c, d);
}

// What ngsq_bam_shard_verify needs of a shard whether or not its scan reached the end: *assumed = the scan ran from an
// ASSUMED first record (not the header, not an offset a neighbour confirmed) -- a failure of such a scan may be the
// assumption's fault (a chain of plausible records inside somebody's auxiliary data that dies later) and is forgiven
// once: the shard is scanned again from the confirmed offset.
int ngsq_bam_shard_peek(ngsq_bam *b, ngsq_bam_shard_info *out, int *assumed) {
    memset(out, 0, sizeof *out);
    *assumed = 0;
    if (!b->dev || !b->dev->sharded) return ngsq_bam_fail(NGSQ_ERR_STATE, "ngsq_bam_shard_begin first");
    DeviceIngest *d = b->dev;
    *assumed = d->entry_mode == DeviceIngest::ENTRY_FIND;
    if (!d->complete) return ngsq_bam_fail(NGSQ_ERR_STATE, "%s: the shard has not been scanned to its end", b->path.c_str());
    out->n_records = d->n_own;
    out->begin_voffset = d->begin_voffset;
    out->end_voffset = d->end_voffset;
    out->first_key = d->n_own ? sort_key(d->first_key) : 0;
    out->last_key = d->n_own ? sort_key(d->last_key) : 0;
    return NGSQ_OK;
}

extern "C" int ngsq_bam_shard_end(ngsq_bam *b, ngsq_bam_shard_info *out) {
    if (!b || !out) return ngsq_bam_fail(NGSQ_ERR_INVALID_ARGUMENT, "null argument");
    int assumed = 0;
    return ngsq_bam_shard_peek(b, out, &assumed);
}

extern "C" {

int ngsq_bam_next_batch_device(ngsq_bam *b, ngsq_ctx *c, uint64_t max_records, ngsq_batch *out) {
    if (!b || !c || !out) return ngsq_bam_fail(NGSQ_ERR_INVALID_ARGUMENT, "null argument");
    if (b->host_mode) return ngsq_bam_fail(NGSQ_ERR_STATE, "%s: this reader is in host ingest mode", b->path.c_str());
    memset(out, 0, sizeof *out);
    out->struct_size = sizeof *out;
    out->location = NGSQ_MEM_DEVICE;
    out->first_record_index = b->n_read;
    BHIP(hipSetDevice(c->device));
    DeviceIngest *d = b->dev;
    if (!d) { // the whole file
        int rc = open_ingest(b, c, &d);
        if (rc) return rc;
        b->dev = d;
        b->dev_free = free_ingest;
        d->header_bytes0 = b->header_bytes;
        rc = start_ingest(b, c, d);
        if (rc) return rc;
    }
    if (d->ctx != c) return ngsq_bam_fail(NGSQ_ERR_STATE, "%s: device ingest is bound to another context", b->path.c_str());
    if (max_records == 0) return NGSQ_OK;
    hipStream_t st = c->stream;
    while (d->cursor == d->n_rec) {
        if (d->complete) return NGSQ_OK; // clean end of the file (of the shard)
        if (d->last_chunk) {
            d->complete = true;
            return NGSQ_OK;
        }
        const int rc = load_chunk(b, d);
        if (rc) return rc;
    }
    const uint64_t n = std::min<uint64_t>(max_records, d->n_rec - d->cursor);
    const uint64_t *rec = d->d_rec_off.p + d->cursor;
    // ---- fixed-width columns and the layout decision (the rule of bam_reader.cpp)
    BHIP(d->d_flag.reserve(n + 64));
    BHIP(d->d_n_cigar.reserve(n + 64));
    BHIP(d->d_mapq.reserve(n + 64));
    BHIP(d->d_ref_id.reserve(n + 64));
    BHIP(d->d_pos.reserve(n + 64));
    BHIP(d->d_mate.reserve(n + 64));
    BHIP(d->d_tlen.reserve(n + 64));
    BHIP(d->d_l_seq.reserve(n + 64));
    BHIP(d->d_var_base.reserve(2 * n + 64));
    BHIP(d->d_record_id.reserve(n + 64));
    RecColumns col{};
    col.flag = d->d_flag.p;
    col.n_cigar = d->d_n_cigar.p;
    col.mapq = d->d_mapq.p;
    col.ref_id = d->d_ref_id.p;
    col.pos = d->d_pos.p;
    col.mate_ref_id = d->d_mate.p;
    col.tlen = d->d_tlen.p;
    col.l_seq = d->d_l_seq.p;
    BHIP(d->d_len.reserve(3 * (n + 1)));
    uint64_t *const sl = d->d_len.p, *const ql = sl + (n + 1), *const cl = ql + (n + 1);
    BHIP(d->h_small.reserve(64 * sizeof(uint64_t)));
    unsigned long long *const host_stats = static_cast<unsigned long long *>(d->h_small.h) + 16;
    {
        KernelTimer kt(d->ctx, K_REC_COLUMNS, n * 36);
        // the records' ids: their virtual offsets, by the block table of the chunk they come from
        const DeviceIngest::Pending &cp = d->pend[d->cur_slot];
        RecOrigin org{};
        org.record_id = d->d_record_id.p;
        org.blocks = d->d_blocks_s[d->cur_slot].p;
        org.coff = d->d_coff_s[d->cur_slot].p;
        org.n_blocks = (uint32_t)cp.blocks.size();
        org.carry = d->carry_len;
        org.carry_id = d->carry_id;
        BHIP(launch_rec_fixed(d->raw, rec, n, col, d->d_var_base.p, d->d_small.p, static_cast<unsigned long long *>(d->h_small.dev) + 16, org, cl, st));
    }
    // (the kernel's last block has written what the layout decision needs into pinned memory: no copy, one wait)
    BHIP(hipStreamSynchronize(st));
    if (host_stats[H_BAD] != ~0ull)
        return ngsq_bam_fail(NGSQ_ERR_INVALID_ARGUMENT, "%s: malformed record %llu", b->path.c_str(),
                             (unsigned long long)(d->chunk_first_record + host_stats[H_BAD]));
    d->stats.long_cigar_records += host_stats[H_LONG]; // (their operations come from the CG tag: k_rec_fixed)
    if (!d->n_own) d->first_key = host_stats[H_FIRST];
    d->last_key = host_stats[H_LAST];
    d->n_own += n;
    const uint32_t max_l = (uint32_t)host_stats[H_MAXL], max_ops = (uint32_t)host_stats[H_MAXOPS];
    const uint64_t sum_qual = host_stats[H_SUML];
    const uint32_t pitch_q = max_l, pitch_s = (max_l + 1) / 2;
    const bool fixed = max_l >= 1 && max_l <= 320 && (uint64_t)pitch_q * n <= sum_qual + sum_qual / 2 + 4096;
    const bool cig1 = max_ops <= 1;
    uint64_t so = (uint64_t)pitch_s * n, qo = (uint64_t)pitch_q * n, co = n;
    if (!fixed || !cig1) {
        size_t tmp_bytes = 0;
        BHIP(launch_exclusive_scan_u64(sl, n + 1, nullptr, &tmp_bytes, st));
        BHIP(d->d_scan_tmp.reserve(tmp_bytes + 256));
        if (!fixed) {
            // SEQ and QUAL lengths (absent qualities take no bytes: the kernel looks) -> offsets; their totals come back
            BHIP(hipMemsetAsync(sl, 0, 2 * (n + 1) * sizeof(uint64_t), st));
            BHIP(launch_rec_lengths(d->raw, rec, n, sl, ql, nullptr, st));
            for (int k = 0; k < 2; k++) {
                uint64_t *arr = d->d_len.p + (size_t)k * (n + 1);
                size_t tb = d->d_scan_tmp.cap;
                BHIP(launch_exclusive_scan_u64(arr, n + 1, d->d_scan_tmp.p, &tb, st));
                BHIP(launch_copy_words(static_cast<uint64_t *>(d->h_small.dev) + 24 + k, arr + n, sizeof(uint64_t), st));
            }
            BHIP(hipStreamSynchronize(st));
            so = static_cast<const uint64_t *>(d->h_small.h)[24];
            qo = static_cast<const uint64_t *>(d->h_small.h)[25];
            col.seq_off = sl;
            col.qual_off = ql;
        }
        if (!cig1) { // the operations per record are there (k_rec_fixed), their sum too: the offsets need no wait
            size_t tb = d->d_scan_tmp.cap;
            BHIP(launch_exclusive_scan_u64(cl, n + 1, d->d_scan_tmp.p, &tb, st));
            co = host_stats[H_SUMOPS];
            col.cigar_off = cl;
        }
    }
    BHIP(d->d_seq.reserve(so + 64));
    BHIP(d->d_qual.reserve(qo + 64));
    BHIP(d->d_cigar.reserve(co + 16));
    col.seq = d->d_seq.p;
    col.qual = d->d_qual.p;
    col.cigar = d->d_cigar.p;
    col.seq_pitch = pitch_s;
    col.qual_pitch = pitch_q;
    {
        KernelTimer kt(d->ctx, K_REC_COLUMNS, 2 * (so + qo + co * 4));
        BHIP(launch_rec_var(d->raw, d->d_var_base.p, n, col, so, qo, st));
    }
    d->cursor += n;
    b->n_read += n;
    out->n_records = n;
    out->flag = col.flag;
    out->mapq = col.mapq;
    out->ref_id = col.ref_id;
    out->pos = col.pos;
    out->mate_ref_id = col.mate_ref_id;
    out->tlen = col.tlen;
    out->l_seq = col.l_seq;
    out->n_cigar = col.n_cigar;
    out->seq = col.seq;
    out->qual = col.qual;
    out->cigar = col.cigar;
    out->record_id = d->d_record_id.p;
    out->max_l_seq = max_l;
    out->seq_bytes = so;
    out->qual_bytes = qo;
    out->cigar_ops = co;
    if (fixed) {
        out->seq_stride = pitch_s;
        out->qual_stride = pitch_q;
    } else {
        out->seq_off = col.seq_off;
        out->qual_off = col.qual_off;
    }
    if (cig1) out->cigar_stride = 1;
    else out->cigar_off = col.cigar_off;
    d->stats.batches += 1;
    d->stats.batches_fixed_rows += fixed;
    d->stats.batches_one_op += cig1;
    return NGSQ_OK;
}

int ngsq_bam_device_stats(const ngsq_bam *b, ngsq_bam_ingest_stats *out) {
    if (!b || !out) return ngsq_bam_fail(NGSQ_ERR_INVALID_ARGUMENT, "null argument");
    if (!b->dev) return ngsq_bam_fail(NGSQ_ERR_STATE, "%s: no device ingest on this handle", b->path.c_str());
    *out = b->dev->stats;
    return NGSQ_OK;
}
int ngsq_bgzf_inflate_device(ngsq_ctx *c, const uint8_t *comp, uint64_t comp_len, uint8_t *out, uint64_t out_cap,
                             uint64_t *out_len, int check_crc) {
    if (!c || !comp || !out_len) return dfail(c, NGSQ_ERR_INVALID_ARGUMENT, "null argument");
    std::vector<BgzfBlock> blocks;
    size_t consumed = 0;
    uint64_t total = 0;
    std::string err;
    if (!bgzf_split(comp, comp_len, &blocks, &consumed, &total, &err))
        return dfail(c, NGSQ_ERR_INVALID_ARGUMENT, "%s", err.c_str());
    if (consumed != comp_len) return dfail(c, NGSQ_ERR_INVALID_ARGUMENT, "truncated BGZF block at end of buffer");
    *out_len = total;
    if (total > out_cap) return dfail(c, NGSQ_ERR_INVALID_ARGUMENT, "output buffer too small: %llu > %llu",
                                      (unsigned long long)total, (unsigned long long)out_cap);
    if (blocks.empty()) return NGSQ_OK;
    DHIP(c, hipSetDevice(c->device));
    uint8_t *d_comp = nullptr, *d_out = nullptr;
    BgzfBlock *d_blocks = nullptr;
    uint32_t *d_status = nullptr;
    int rc = NGSQ_OK;
    std::vector<uint32_t> status(blocks.size());
    auto cleanup = [&]() {
        (void)hipFree(d_comp);
        (void)hipFree(d_out);
        (void)hipFree(d_blocks);
        (void)hipFree(d_status);
    };
#define TRY(expr)                                                                                   \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess) {                                                                     \
            cleanup();                                                                              \
            return dfail(c, NGSQ_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(e_));                  \
        }                                                                                           \
    } while (0)
    TRY(hipMalloc((void **)&d_comp, comp_len + INFLATE_IN_SLACK));
    TRY(hipMalloc((void **)&d_out, total + 64));
    TRY(hipMalloc((void **)&d_blocks, blocks.size() * sizeof(BgzfBlock)));
    TRY(hipMalloc((void **)&d_status, (blocks.size() + 1) * sizeof(uint32_t)));
    TRY(hipMemcpyAsync(d_comp, comp, comp_len, hipMemcpyHostToDevice, c->stream));
    TRY(hipMemsetAsync(d_comp + comp_len, 0, INFLATE_IN_SLACK, c->stream));
    TRY(hipMemcpyAsync(d_blocks, blocks.data(), blocks.size() * sizeof(BgzfBlock), hipMemcpyHostToDevice, c->stream));
    TRY(launch_bgzf_inflate(d_comp, d_blocks, (uint32_t)blocks.size(), d_out, d_status, d_status + blocks.size(), check_crc != 0, c->stream));
    TRY(hipMemcpyAsync(status.data(), d_status, blocks.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    if (total) TRY(hipMemcpyAsync(out, d_out, total, hipMemcpyDeviceToHost, c->stream));
    TRY(hipStreamSynchronize(c->stream));
#undef TRY
    for (size_t k = 0; k < blocks.size() && rc == NGSQ_OK; k++)
        if (status[k] != INF_OK)
            rc = dfail(c, NGSQ_ERR_INVALID_ARGUMENT, "BGZF block %zu: %s", k, inflate_status_text(status[k]));
    cleanup();
    return rc;
}

} // extern "C"
