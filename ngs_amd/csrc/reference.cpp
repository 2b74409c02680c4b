// reference.cpp -- include/ngsq_reference.h: the reference FASTA of the Edits facet, from file to the two packed copies the
// kernels compare with.  Replaces EditsFacet::setup's "open the FASTA, read records until the name matches, keep the sequence"
// per @SQ (src/qc/sequence_based/edits.rs:177-215) and the per-read Base::try_from of edits.rs:257-261.
//
// Host work per byte of FASTA: one pread() into pinned memory -- and, without a .fai, one memchr for '>' through a mapping
// of the file, which ngsq_fasta_open starts on its own threads before the device is even initialised.  Everything else --
// dropping the line terminators, letters to 4-bit codes, packing -- is done by HIP kernels (reference_kernels.hip).
#include <fcntl.h>
#include <hip/hip_runtime_api.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/ngsq_reference.h"
#include "context.h"
#include "mem_pool.h"
#include "reference_kernels.h"

namespace {

thread_local std::string g_fa_err;

int fa_fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_fa_err = buf;
    return code;
}

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct FaRecord {
    std::string name;
    uint64_t text_begin = 0, text_end = 0; // the record's sequence lines in the file
};

int default_threads() {
    int n = (int)std::thread::hardware_concurrency();
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) { // the cgroup's CPU quota, as the readers of ngsq_bam.h count cores
        char quota[32] = {0};
        long period = 0;
        if (fscanf(f, "%31s %ld", quota, &period) == 2 && strcmp(quota, "max") != 0 && period > 0) {
            const long q = atol(quota) / period;
            if (q >= 1 && q < n) n = (int)q;
        }
        fclose(f);
    }
    return std::max(1, std::min(4, n - 2));
}

} // namespace

struct ngsq_fasta {
    std::string path;
    int fd = -1;
    uint64_t size = 0;
    const uint8_t *map = nullptr;
    int n_threads = 1;
    std::vector<FaRecord> recs;
    std::thread indexer;
    std::mutex mu;
    bool joined = false;
    std::string index_err;
    double index_s = 0;
    bool from_fai = false;
};

namespace {

// the definition line that starts at `pos`: name = the text behind '>' up to the first blank; *text = the byte behind the line
void parse_definition(const uint8_t *d, uint64_t size, uint64_t pos, std::string *name, uint64_t *text) {
    const uint8_t *eol = static_cast<const uint8_t *>(memchr(d + pos, '\n', size - pos));
    const uint64_t end = eol ? (uint64_t)(eol - d) : size;
    uint64_t q = pos + 1;
    while (q < end && d[q] != ' ' && d[q] != '\t' && d[q] != '\r') q++;
    name->assign(reinterpret_cast<const char *>(d + pos + 1), q - pos - 1);
    *text = std::min(end + 1, size);
}

// <path>.fai (samtools faidx: name, length, offset, linebases, linewidth) in place of the scan -- when every line of it points
// at the byte behind a definition line of that name, and the records it lists are all the file has up to the last of them
bool index_from_fai(ngsq_fasta *f) {
    FILE *fi = fopen((f->path + ".fai").c_str(), "r");
    if (!fi) return false;
    std::vector<FaRecord> recs;
    char line[4096];
    bool ok = true;
    uint64_t prev_end = 0;
    while (ok && fgets(line, sizeof line, fi)) {
        char name[2048];
        unsigned long long len = 0, off = 0, lb = 0, lw = 0;
        if (sscanf(line, "%2047[^\t]\t%llu\t%llu\t%llu\t%llu", name, &len, &off, &lb, &lw) != 5 || lw < lb || off > f->size) {
            ok = false;
            break;
        }
        const uint64_t text = lb ? len / lb * lw + len % lb : 0;
        FaRecord r;
        r.name = name;
        r.text_begin = off;
        r.text_end = std::min<uint64_t>(f->size, off + text);
        // the definition line in front of `off`: "...\n>name[ description]\n" with nothing but that line since the previous record's text
        if (off == 0 || f->map[off - 1] != '\n') { ok = false; break; }
        uint64_t p = off - 1;
        while (p > 0 && f->map[p - 1] != '\n') p--;
        const size_t nl = strlen(name);
        if (f->map[p] != '>' || off - 1 - p < 1 + nl || memcmp(f->map + p + 1, name, nl) != 0) { ok = false; break; }
        const uint8_t after = f->map[p + 1 + nl];
        if (after != '\n' && after != ' ' && after != '\t' && after != '\r') { ok = false; break; }
        // between the previous record's text (by the index's arithmetic) and this definition line: at most line terminators
        for (uint64_t q = prev_end; q < p && ok; q++) ok = f->map[q] == '\n' || f->map[q] == '\r';
        if (p < prev_end) ok = false;
        prev_end = r.text_end;
        recs.push_back(std::move(r));
    }
    fclose(fi);
    if (!ok || recs.empty()) return false;
    // behind the last record the index knows: nothing but line terminators (a record appended after `samtools faidx` ran
    // would be missed)
    for (uint64_t q = prev_end; q < f->size; q++)
        if (f->map[q] != '\n' && f->map[q] != '\r') return false;
    // text_end by the next definition line rather than by arithmetic: the device strips the terminators either way
    for (size_t i = 0; i + 1 < recs.size(); i++) {
        uint64_t p = recs[i + 1].text_begin - 1;
        while (p > 0 && f->map[p - 1] != '\n') p--;
        recs[i].text_end = p;
    }
    recs.back().text_end = f->size;
    f->recs.swap(recs);
    return true;
}

void index_main(ngsq_fasta *f) {
    const double t0 = now_s();
    if (f->size == 0) {
        f->index_s = 0;
        return;
    }
    const char *no_fai = getenv("NGSQ_FASTA_NO_FAI");
    if (!(no_fai && atoi(no_fai)) && index_from_fai(f)) {
        f->from_fai = true;
        f->index_s = now_s() - t0;
        return;
    }
    // every '>' that starts a line, found by n_threads memchr loops over disjoint ranges of the mapping
    const int nt = f->n_threads;
    std::vector<std::vector<uint64_t>> found((size_t)nt);
    std::vector<std::thread> th;
    const uint64_t per = (f->size + nt - 1) / nt;
    for (int t = 0; t < nt; t++)
        th.emplace_back([f, t, per, &found] {
            const uint64_t lo = std::min(f->size, per * (uint64_t)t), hi = std::min(f->size, lo + per);
            const uint8_t *d = f->map;
            for (uint64_t p = lo; p < hi;) {
                const uint8_t *q = static_cast<const uint8_t *>(memchr(d + p, '>', hi - p));
                if (!q) break;
                const uint64_t pos = (uint64_t)(q - d);
                if (pos == 0 || d[pos - 1] == '\n') found[(size_t)t].push_back(pos);
                p = pos + 1;
            }
        });
    for (auto &x : th) x.join();
    std::vector<uint64_t> starts;
    for (auto &v : found) starts.insert(starts.end(), v.begin(), v.end());
    f->recs.resize(starts.size());
    for (size_t i = 0; i < starts.size(); i++) {
        parse_definition(f->map, f->size, starts[i], &f->recs[i].name, &f->recs[i].text_begin);
        f->recs[i].text_end = i + 1 < starts.size() ? starts[i + 1] : f->size;
        if (f->recs[i].text_begin > f->recs[i].text_end) f->recs[i].text_begin = f->recs[i].text_end;
    }
    if (starts.empty() || starts[0] != 0) {
        // noodles-fasta: the first line of a record must be a definition ('>' ...)
        uint64_t p = 0;
        while (p < f->size && (f->map[p] == '\n' || f->map[p] == '\r')) p++;
        if (p < f->size && (starts.empty() || starts[0] != p)) f->index_err = "invalid FASTA: sequence data before the first definition line";
    }
    f->index_s = now_s() - t0;
}

void wait_index(ngsq_fasta *f) {
    std::lock_guard<std::mutex> g(f->mu);
    if (!f->joined) {
        if (f->indexer.joinable()) f->indexer.join();
        f->joined = true;
    }
}

} // namespace

extern "C" {

const char *ngsq_fasta_last_error(void) { return g_fa_err.c_str(); }

int ngsq_fasta_base_code(uint8_t byte) { return ngsq::fasta_base_code(byte); }

int ngsq_fasta_open(const char *path, int n_threads, ngsq_fasta **out) {
    if (!path || !out) return fa_fail(NGSQ_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    const int fd = open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) return fa_fail(NGSQ_ERR_INVALID_ARGUMENT, "%s (os error %d)", strerror(errno), errno);
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) {
        close(fd);
        return fa_fail(NGSQ_ERR_INVALID_ARGUMENT, "not a regular file");
    }
    ngsq_fasta *f = new ngsq_fasta();
    f->path = path;
    f->fd = fd;
    f->size = (uint64_t)st.st_size;
    f->n_threads = n_threads > 0 ? std::min(n_threads, 32) : default_threads();
    if (f->size) {
        void *m = mmap(nullptr, f->size, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) {
            close(fd);
            delete f;
            return fa_fail(NGSQ_ERR_INVALID_ARGUMENT, "mmap failed (os error %d)", errno);
        }
        f->map = static_cast<const uint8_t *>(m);
    }
    f->indexer = std::thread(index_main, f);
    *out = f;
    return NGSQ_OK;
}

void ngsq_fasta_close(ngsq_fasta *f) {
    if (!f) return;
    wait_index(f);
    if (f->map) munmap(const_cast<uint8_t *>(f->map), f->size);
    if (f->fd >= 0) close(f->fd);
    delete f;
}

int64_t ngsq_fasta_n_records(ngsq_fasta *f) {
    if (!f) return NGSQ_ERR_INVALID_ARGUMENT;
    wait_index(f);
    if (!f->index_err.empty()) return fa_fail(NGSQ_ERR_INVALID_ARGUMENT, "%s", f->index_err.c_str());
    return (int64_t)f->recs.size();
}
const char *ngsq_fasta_record_name(ngsq_fasta *f, uint32_t i) {
    if (!f) return nullptr;
    wait_index(f);
    return i < f->recs.size() ? f->recs[i].name.c_str() : nullptr;
}
uint64_t ngsq_fasta_record_text_bytes(ngsq_fasta *f, uint32_t i) {
    if (!f) return 0;
    wait_index(f);
    return i < f->recs.size() ? f->recs[i].text_end - f->recs[i].text_begin : 0;
}
double ngsq_fasta_index_seconds(ngsq_fasta *f) {
    if (!f) return 0;
    wait_index(f);
    return f->index_s;
}
int ngsq_fasta_index_from_fai(ngsq_fasta *f) {
    if (!f) return 0;
    wait_index(f);
    return f->from_fai ? 1 : 0;
}

} // extern "C"

// ---------------------------------------------------------------------------------------------------------------------
// the load
// ---------------------------------------------------------------------------------------------------------------------
namespace {

struct Loader {
    ngsq_ctx *c = nullptr;
    ngsq_fasta *f = nullptr;
    std::vector<std::string> names;
    std::vector<uint8_t> wanted;
    std::thread th;
    bool joined = false;
    int rc = NGSQ_OK;
    std::string err;
    ngsq_reference_stats stats{};
};

int lfail(Loader *L, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    L->err = buf;
    L->rc = code;
    return code;
}

#define LHIP(expr)                                                                                         \
    do {                                                                                                   \
        hipError_t e_ = (expr);                                                                            \
        if (e_ != hipSuccess) return lfail(L, NGSQ_ERR_DEVICE, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

constexpr size_t SLOT = (size_t)32 << 20; // one pinned slot = one piece of the file = one host-to-device copy
constexpr int N_SLOTS = 6;

struct Piece {
    uint64_t file_off, dev_off, len;
};

// the pieces are read by `nt` threads into a ring of pinned slots and copied to the device from there; piece i uses slot
// i % N_SLOTS once the copy of piece i - N_SLOTS has completed
struct Uploader {
    int fd;
    uint8_t *d_text;
    hipStream_t stream;
    int device;
    std::vector<Piece> pieces;
    void *slot[N_SLOTS] = {};
    size_t slot_bytes[N_SLOTS] = {};
    hipEvent_t ev[N_SLOTS] = {};
    std::mutex mu;
    std::condition_variable cv;
    int64_t issued[N_SLOTS]; // the last piece whose copy has been queued from the slot (-1: none)
    std::atomic<size_t> next{0};
    std::atomic<bool> bad{false};
    std::string err;

    void fail(const std::string &why) {
        std::lock_guard<std::mutex> g(mu);
        if (err.empty()) err = why;
        bad = true;
        cv.notify_all();
    }
    void work() {
        if (hipSetDevice(device) != hipSuccess) return fail("hipSetDevice failed in a FASTA reader thread");
        for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= pieces.size() || bad) return;
            const int s = (int)(i % N_SLOTS);
            {
                std::unique_lock<std::mutex> g(mu);
                cv.wait(g, [&] { return bad || issued[s] == (int64_t)i - N_SLOTS; });
                if (bad) return;
            }
            if (i >= (size_t)N_SLOTS && hipEventSynchronize(ev[s]) != hipSuccess) return fail("hipEventSynchronize failed (FASTA upload)");
            const Piece &p = pieces[i];
            uint8_t *dst = static_cast<uint8_t *>(slot[s]);
            for (uint64_t n = 0; n < p.len;) {
                const ssize_t r = pread(fd, dst + n, p.len - n, (off_t)(p.file_off + n));
                if (r <= 0) return fail(r < 0 ? std::string("read error on the reference FASTA: ") + strerror(errno) : "the reference FASTA got shorter while it was read");
                n += (uint64_t)r;
            }
            if (ngsq::pool_pinned_h2d(d_text + p.dev_off, slot[s], 0, p.len, stream) != hipSuccess || hipEventRecord(ev[s], stream) != hipSuccess)
                return fail("hipMemcpyAsync failed (FASTA upload)");
            {
                std::lock_guard<std::mutex> g(mu);
                issued[s] = (int64_t)i;
            }
            cv.notify_all();
        }
    }
};

int load_main(Loader *L) {
    ngsq_ctx *c = L->c;
    ngsq_fasta *f = L->f;
    const double t_begin = now_s();
    ngsq_reference_stats &S = L->stats;
    // ---- the index (started by ngsq_fasta_open)
    wait_index(f);
    S.index_wait_s = now_s() - t_begin;
    if (!f->index_err.empty()) return lfail(L, NGSQ_ERR_INVALID_ARGUMENT, "%s", f->index_err.c_str());
    const uint32_t nr = c->st.n_refs;
    // record.name() == seq_name: the FIRST record of that name (edits.rs:196-203 breaks at it)
    std::unordered_map<std::string, uint32_t> by_name;
    by_name.reserve(f->recs.size() * 2);
    for (uint32_t i = 0; i < f->recs.size(); i++) by_name.emplace(f->recs[i].name, i);
    struct Want {
        uint32_t ref, rec;
        uint64_t dev_off;
    };
    std::vector<Want> want;
    for (uint32_t r = 0; r < nr; r++) {
        if (!L->wanted.empty() && !L->wanted[r]) continue;
        if (c->edits_off[r] == ngsq::NO_DEPTH) continue;
        auto it = by_name.find(L->names[r]);
        if (it == by_name.end()) return lfail(L, NGSQ_ERR_INVALID_ARGUMENT, "sequence %s not found in reference FASTA.", L->names[r].c_str()); // edits.rs:207-209
        want.push_back(Want{r, it->second, 0});
    }
    // in file order: the reads run forward through the file
    std::sort(want.begin(), want.end(), [&](const Want &a, const Want &b) { return f->recs[a.rec].text_begin < f->recs[b.rec].text_begin; });
    uint64_t total = 0, n_tiles = 0;
    std::vector<ngsq::FastaSeqDev> seqs(want.size());
    std::vector<uint64_t> tile_first(want.size() + 1, 0);
    for (size_t k = 0; k < want.size(); k++) {
        const FaRecord &rec = f->recs[want[k].rec];
        const uint64_t len = rec.text_end - rec.text_begin;
        if (len >= (1ull << 32)) return lfail(L, NGSQ_ERR_LIMIT, "sequence %s: %llu bytes of FASTA text; the loader takes records below 4 GiB", rec.name.c_str(), (unsigned long long)len);
        want[k].dev_off = total;
        seqs[k] = ngsq::FastaSeqDev{total, len};
        tile_first[k] = n_tiles;
        n_tiles += (len + ngsq::FASTA_TILE - 1) / ngsq::FASTA_TILE;
        total += (len + 255) & ~255ull;
    }
    tile_first[want.size()] = n_tiles;
    S.text_bytes = 0;
    for (auto &s : seqs) S.text_bytes += s.text_len;
    S.sequences = (uint32_t)want.size();

    LHIP(hipSetDevice(c->device));
    hipStream_t stream = nullptr;
    LHIP(ngsq::pool_stream_get(false, &stream));
    uint8_t *d_text = nullptr, *d_codes = nullptr;
    size_t text_got = 0, codes_got = 0;
    void *d_small = nullptr; // seqs | tile_first | seq_len | n_bad | bad_list
    uint32_t *d_counts = nullptr;
    uint64_t *d_tile_base = nullptr;
    constexpr uint32_t BAD_CAP = 1u << 16;
    const size_t nseq = want.size();
    const size_t off_seqs = 0, off_tf = off_seqs + nseq * sizeof(ngsq::FastaSeqDev), off_len = off_tf + (nseq + 1) * 8, off_nbad = off_len + nseq * 8,
                 off_bad = off_nbad + 8, small_bytes = off_bad + (size_t)BAD_CAP * 8;
    std::vector<unsigned long long> h_back; // seq_len | n_bad | bad_list, after the kernels
    int rc = NGSQ_OK;
    Uploader up;
    auto cleanup = [&]() {
        (void)hipStreamSynchronize(stream);
        ngsq::pool_device_free(d_text, text_got);
        ngsq::pool_device_free(d_codes, codes_got);
        (void)hipFree(d_small);
        (void)hipFree(d_counts);
        (void)hipFree(d_tile_base);
        for (int s = 0; s < N_SLOTS; s++) {
            if (up.slot[s]) ngsq::pool_pinned_free(up.slot[s], up.slot_bytes[s]);
            if (up.ev[s]) ngsq::pool_event_put(up.ev[s]);
        }
        ngsq::pool_stream_put(false, stream);
    };
#define LTRY(expr)                                                                                                 \
    do {                                                                                                           \
        hipError_t e_ = (expr);                                                                                    \
        if (e_ != hipSuccess) {                                                                                    \
            rc = lfail(L, NGSQ_ERR_DEVICE, "%s failed: %s", #expr, hipGetErrorString(e_));                         \
            cleanup();                                                                                             \
            return rc;                                                                                             \
        }                                                                                                          \
    } while (0)
    if (nseq) {
        LTRY(ngsq::pool_device_alloc((void **)&d_text, total + 256, &text_got));
        LTRY(ngsq::pool_device_alloc((void **)&d_codes, total + 256, &codes_got));
        LTRY(hipMalloc(&d_small, small_bytes));
        LTRY(hipMalloc((void **)&d_counts, (n_tiles + 1) * 4));
        LTRY(hipMalloc((void **)&d_tile_base, (n_tiles + 1) * 8));
        LTRY(hipMemsetAsync(static_cast<uint8_t *>(d_small) + off_len, 0, small_bytes - off_len, stream));
        LTRY(hipMemcpyAsync(static_cast<uint8_t *>(d_small) + off_seqs, seqs.data(), nseq * sizeof(ngsq::FastaSeqDev), hipMemcpyHostToDevice, stream));
        LTRY(hipMemcpyAsync(static_cast<uint8_t *>(d_small) + off_tf, tile_first.data(), (nseq + 1) * 8, hipMemcpyHostToDevice, stream));
        // ---- the text: file -> pinned slots -> device, a few threads
        const double t_read = now_s();
        up.fd = f->fd;
        up.d_text = d_text;
        up.stream = stream;
        up.device = c->device;
        for (size_t k = 0; k < nseq; k++) {
            const FaRecord &rec = f->recs[want[k].rec];
            for (uint64_t o = 0; o < seqs[k].text_len; o += SLOT)
                up.pieces.push_back(Piece{rec.text_begin + o, seqs[k].text_off + o, std::min<uint64_t>(SLOT, seqs[k].text_len - o)});
        }
        for (int s = 0; s < N_SLOTS; s++) {
            up.issued[s] = (int64_t)s - N_SLOTS;
            if ((size_t)s < up.pieces.size()) {
                LTRY(ngsq::pool_pinned_alloc(&up.slot[s], SLOT, &up.slot_bytes[s]));
                LTRY(ngsq::pool_event_get(&up.ev[s]));
            }
        }
        {
            const int nt = (int)std::min<size_t>((size_t)f->n_threads, up.pieces.size());
            std::vector<std::thread> th;
            for (int t = 0; t < nt; t++) th.emplace_back([&up] { up.work(); });
            for (auto &x : th) x.join();
        }
        if (up.bad) {
            rc = lfail(L, NGSQ_ERR_DEVICE, "%s", up.err.c_str());
            cleanup();
            return rc;
        }
        S.read_s = now_s() - t_read;
        // ---- text -> codes -> the two packed copies of every sequence
        hipEvent_t e0 = nullptr, e1 = nullptr;
        LTRY(hipEventCreate(&e0));
        LTRY(hipEventCreate(&e1));
        LTRY(hipEventRecord(e0, stream));
        uint8_t *sm = static_cast<uint8_t *>(d_small);
        LTRY(ngsq::launch_fasta_convert(c->li, d_text, reinterpret_cast<ngsq::FastaSeqDev *>(sm + off_seqs), (uint32_t)nseq, reinterpret_cast<uint64_t *>(sm + off_tf),
                                        n_tiles, d_counts, d_tile_base, reinterpret_cast<unsigned long long *>(sm + off_len), d_codes,
                                        reinterpret_cast<unsigned long long *>(sm + off_nbad), reinterpret_cast<unsigned long long *>(sm + off_bad), BAD_CAP, stream));
        h_back.resize(nseq + 1);
        LTRY(hipMemcpyAsync(h_back.data(), sm + off_len, (nseq + 1) * 8, hipMemcpyDeviceToHost, stream));
        LTRY(hipStreamSynchronize(stream)); // the lengths decide how much of each sequence is packed
        uint8_t *bases = c->d_ref_bases;
        for (size_t k = 0; k < nseq; k++) {
            const uint32_t r = want[k].ref;
            const uint64_t Lr = c->ref_len[r], have = std::min<uint64_t>(h_back[k], Lr);
            // (the context's copies were zeroed when it was created: bytes behind `have` read as 0, as k_pack_reference writes them)
            LTRY(ngsq::launch_pack_reference(c->li, d_codes + seqs[k].text_off, have, bases + c->bases_off[r], bases + c->nbases + c->bases_off[r], Lr / 2 + 1,
                                             reinterpret_cast<unsigned long long *>(sm + off_nbad) + 0 /* codes are <= 15 by construction */, stream));
        }
        LTRY(hipEventRecord(e1, stream));
        LTRY(hipStreamSynchronize(stream));
        float ms = 0;
        if (hipEventElapsedTime(&ms, e0, e1) == hipSuccess) S.device_s = ms * 1e-3;
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
    }
    // ---- what the FASTA held: lengths, bytes that are no base letters
    std::vector<uint32_t> lens(2 * (size_t)nr, 0); // reach | fast-path bound (sequences that were not wanted: 0 -- never compared)
    unsigned long long n_bad = 0;
    std::vector<std::vector<uint32_t>> bad((size_t)nr);
    if (nseq) {
        n_bad = h_back[nseq];
        // (k_pack_reference's `bad` counter shares the word: it cannot fire, every code is <= 15)
        if (n_bad > BAD_CAP) {
            rc = lfail(L, NGSQ_ERR_LIMIT, "the reference FASTA holds %llu bytes that are no base letters; the loader keeps the positions of %u", n_bad, BAD_CAP);
            cleanup();
            return rc;
        }
        if (n_bad) {
            std::vector<unsigned long long> list((size_t)n_bad);
            LTRY(hipMemcpy(list.data(), static_cast<uint8_t *>(d_small) + off_bad, (size_t)n_bad * 8, hipMemcpyDeviceToHost));
            for (unsigned long long v : list) {
                const size_t k = (size_t)(v >> 40);
                const uint64_t pos = v & ((1ull << 40) - 1);
                if (k < nseq && pos <= 0xFFFFFFFFull) bad[want[k].ref].push_back((uint32_t)pos); // (beyond LN too: a read may end there, ed_walk_record)
            }
        }
    }
    for (size_t k = 0; k < nseq; k++) {
        const uint32_t r = want[k].ref;
        const uint64_t Lr = c->ref_len[r], have = std::min<uint64_t>(h_back[k], Lr);
        S.bases += have;
        S.shorter += h_back[k] < Lr;
        S.longer += h_back[k] > Lr;
        lens[r] = (uint32_t)std::min<uint64_t>(h_back[k], 0xFFFFFFFFull); // the FASTA's own count: a read may end beyond LN inside a longer sequence
        std::sort(bad[r].begin(), bad[r].end());
        // a sequence with listed positions INSIDE what the window lanes may reach: every record takes the walk, which looks them up
        // (positions beyond LN are only ever under reads that end beyond LN -- the walk's anyway)
        lens[nr + r] = bad[r].empty() || bad[r].front() > have ? (uint32_t)have : 0u;
    }
    S.invalid_bytes = n_bad;
    if (nr) {
        if (!c->d_edits_len) LTRY(hipMalloc((void **)&c->d_edits_len, lens.size() * 4));
        LTRY(hipMemcpy(c->d_edits_len, lens.data(), lens.size() * 4, hipMemcpyHostToDevice));
        c->st.ref_edits_len = c->d_edits_len;
        c->st.ref_fast_len = c->d_edits_len + nr;
        bool any_bad = false;
        for (auto &v : bad) any_bad = any_bad || !v.empty();
        if (any_bad) {
            std::vector<uint32_t> off(nr + 1, 0), flat;
            for (uint32_t r = 0; r < nr; r++) {
                off[r] = (uint32_t)flat.size();
                flat.insert(flat.end(), bad[r].begin(), bad[r].end());
            }
            off[nr] = (uint32_t)flat.size();
            LTRY(hipMalloc((void **)&c->d_bad_off, off.size() * 4));
            LTRY(hipMalloc((void **)&c->d_bad_pos, flat.size() * 4));
            LTRY(hipMemcpy(c->d_bad_off, off.data(), off.size() * 4, hipMemcpyHostToDevice));
            LTRY(hipMemcpy(c->d_bad_pos, flat.data(), flat.size() * 4, hipMemcpyHostToDevice));
            c->st.ref_bad_off = c->d_bad_off;
            c->st.ref_bad_pos = c->d_bad_pos;
        }
    }
    cleanup();
    S.total_s = now_s() - t_begin;
    return NGSQ_OK;
#undef LTRY
}

} // namespace

namespace ngsq {

int reference_join(ngsq_ctx *c) {
    Loader *L = static_cast<Loader *>(c->ref_loader);
    if (!L) return NGSQ_ERR_STATE;
    if (!L->joined) {
        if (L->th.joinable()) L->th.join();
        L->joined = true;
    }
    if (L->rc != NGSQ_OK) {
        c->err = L->err;
        return L->rc;
    }
    c->ref_ready = true;
    return NGSQ_OK;
}

void reference_abandon(ngsq_ctx *c) {
    Loader *L = static_cast<Loader *>(c->ref_loader);
    if (!L) return;
    if (!L->joined && L->th.joinable()) L->th.join();
    delete L;
    c->ref_loader = nullptr;
}

} // namespace ngsq

extern "C" {

int ngsq_reference_load(ngsq_ctx *c, ngsq_fasta *f, const char *const *ref_names, const uint8_t *wanted) {
    if (!c) return NGSQ_ERR_INVALID_ARGUMENT;
    auto cfail = [&](int code, const char *msg) {
        c->err = msg;
        return code;
    };
    if (!f || !ref_names) return cfail(NGSQ_ERR_INVALID_ARGUMENT, "null argument");
    if (!(c->cfg.facets & NGSQ_FACET_EDITS) || !c->ref_deferred)
        return cfail(NGSQ_ERR_STATE, "ngsq_reference_load needs a context with NGSQ_FACET_EDITS and ngsq_config.ref_bases_deferred");
    if (c->ref_loader) return cfail(NGSQ_ERR_STATE, "the reference of this context has been loaded already");
    Loader *L = new Loader();
    L->c = c;
    L->f = f;
    const uint32_t nr = c->st.n_refs;
    L->names.resize(nr);
    for (uint32_t r = 0; r < nr; r++) L->names[r] = ref_names[r] ? ref_names[r] : "";
    if (wanted) L->wanted.assign(wanted, wanted + nr);
    c->ref_loader = L;
    L->th = std::thread([L] { (void)load_main(L); });
    return NGSQ_OK;
}

int ngsq_reference_wait(ngsq_ctx *c) {
    if (!c) return NGSQ_ERR_INVALID_ARGUMENT;
    if (!c->ref_loader) {
        c->err = "ngsq_reference_load was not called";
        return NGSQ_ERR_STATE;
    }
    return ngsq::reference_join(c);
}

int ngsq_reference_get_stats(ngsq_ctx *c, ngsq_reference_stats *out) {
    if (!c || !out) return NGSQ_ERR_INVALID_ARGUMENT;
    Loader *L = static_cast<Loader *>(c->ref_loader);
    if (!L || !L->joined) return NGSQ_ERR_STATE;
    *out = L->stats;
    return NGSQ_OK;
}

} // extern "C"
