// context.cpp -- the C ABI of include/ngsq.h: context lifecycle, batch staging,
// kernel sequencing on one HIP stream, teardown and integer result download.
//
// Host-side mirror of the reference driver's facet lifecycle
// (src/qc/command.rs:288-418): process (batches) -> summarize/teardown ->
// aggregate.  There is NO CPU fallback: without a usable GPU ngsq_create fails
// with NGSQ_ERR_NO_DEVICE.
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/ngsq.h"
#include "../../include/ngsq_shared.h"
#include "context.h"
#include "mem_pool.h"

using namespace ngsq;

namespace ngsq {
void free_exchange_scratch(void *p); // exchange.cpp
}

static thread_local std::string g_err;

// roctx ranges with the names of ngsq_kernel_timing around the same launches, so that a rocprofv3 --marker-trace
// lines up with the library's own timing table.  The marker library is looked up once at run time; without it
// (or without a profiler attached) the calls cost nothing worth measuring.
namespace {
struct Roctx {
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
    Roctx() {
        for (const char *n : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
            if (void *h = dlopen(n, RTLD_NOW | RTLD_LOCAL)) {
                push = reinterpret_cast<int (*)(const char *)>(dlsym(h, "roctxRangePushA"));
                pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
                if (push && pop) return;
                push = nullptr;
                pop = nullptr;
            }
        }
    }
};
const Roctx &roctx() {
    static const Roctx r;
    return r;
}
} // namespace


static int fail(ngsq_ctx *c, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c)
        c->err = buf;
    else
        g_err = buf;
    return code;
}

#define HIP_TRY(c, expr)                                                                        \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return fail((c), NGSQ_ERR_DEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                                    \
    } while (0)

static const char *const KERNEL_NAMES[K_COUNT] = {"fields", "gc", "qual", "edits", "cov_scan", "edits_vaf", "h2d", "features", "cov_stream", "bgzf_inflate", "bgzf_crc",
                                                 "rec_index", "rec_columns"};

extern "C" {

uint32_t ngsq_abi_version(void) { return NGSQ_ABI_VERSION; }

uint64_t ngsq_release_cached_memory(void) { return (uint64_t)ngsq::pool_trim(); }

int ngsq_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int ngsq_device_pci_bus_id(int device, char *buf, size_t cap) {
    char tmp[64] = {0};
    if (!buf || !cap || hipDeviceGetPCIBusId(tmp, (int)sizeof tmp, device) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    for (char *p = tmp; *p; p++)
        if (*p >= 'A' && *p <= 'F') *p = (char)(*p - 'A' + 'a'); // sysfs spells the address in lower case
    snprintf(buf, cap, "%s", tmp);
    return (int)strlen(buf);
}

const char *ngsq_facet_name(uint32_t bit) {
    switch (bit) {
    case NGSQ_FACET_GENERAL: return "General";
    case NGSQ_FACET_TEMPLATE_LENGTH: return "Template Length";
    case NGSQ_FACET_GC_CONTENT: return "GC Content";
    case NGSQ_FACET_QUALITY_SCORE: return "Quality Score";
    case NGSQ_FACET_COVERAGE: return "Coverage";
    case NGSQ_FACET_EDITS: return "Edits";
    case NGSQ_FACET_FEATURES: return "Genomic Features";
    default: return nullptr;
    }
}

const char *ngsq_last_global_error(void) { return g_err.c_str(); }
const char *ngsq_last_error(const ngsq_ctx *c) { return c ? c->err.c_str() : g_err.c_str(); }

uint32_t ngsq_gc_offset(uint64_t seed, uint64_t idx, uint32_t l) { return ngsq_gc_offset_fn(seed, idx, l); }

static uint64_t round_up(uint64_t v, uint64_t m) { return (v + m - 1) / m * m; }

// NGSQ_STEP_LEGACY=1 (measurement aid): ngsq_reset and ngsq_finalize as they were until round 4 -- one hipMemsetAsync per small
// block, one device-to-host copy per result block -- for an A/B of the time a step spends outside its kernels
static bool step_legacy() {
    static const bool v = [] { const char *e = getenv("NGSQ_STEP_LEGACY"); return e && atoi(e) != 0; }();
    return v;
}

int ngsq_create(const ngsq_config *cfg, ngsq_ctx **out) {
    if (!cfg || !out) return fail(nullptr, NGSQ_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    if (cfg->struct_size != sizeof(ngsq_config))
        return fail(nullptr, NGSQ_ERR_INVALID_ARGUMENT, "ngsq_config.struct_size %u != %zu",
                    cfg->struct_size, sizeof(ngsq_config));
    if (cfg->n_refs && !cfg->ref_len)
        return fail(nullptr, NGSQ_ERR_INVALID_ARGUMENT, "ref_len is null");
    if (cfg->facets & ~(NGSQ_FACETS_RECORD_BASED | NGSQ_FACETS_SEQUENCE_BASED))
        return fail(nullptr, NGSQ_ERR_INVALID_ARGUMENT, "unknown facet bits 0x%x", cfg->facets);
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, NGSQ_ERR_NO_DEVICE,
                    "no HIP device available (%s); the ngs qc hot path has no CPU fallback",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (cfg->device < 0 || cfg->device >= ndev)
        return fail(nullptr, NGSQ_ERR_INVALID_ARGUMENT, "device %d out of range (0..%d)", cfg->device,
                    ndev - 1);

    ngsq_ctx *c = new ngsq_ctx();
    c->cfg = *cfg;
    if (!c->cfg.bin_size) c->cfg.bin_size = 50000;
    if (!c->cfg.tlen_cap) c->cfg.tlen_cap = 1024;
    if (!c->cfg.cov_cap) c->cfg.cov_cap = 2048;
    if (!c->cfg.max_read_len) c->cfg.max_read_len = 512;
    if (c->cfg.max_read_len > NGSQ_QUALITY_ROWS_LIMIT || c->cfg.tlen_cap > 15000 || c->cfg.cov_cap > 15000) {
        delete c;
        return fail(nullptr, NGSQ_ERR_INVALID_ARGUMENT, "max_read_len/tlen_cap/cov_cap too large");
    }
    const uint32_t nr = cfg->n_refs;
    c->ref_len.assign(cfg->ref_len, cfg->ref_len + nr);
    c->primary.resize(nr);
    for (uint32_t r = 0; r < nr; r++) c->primary[r] = cfg->ref_is_primary ? cfg->ref_is_primary[r] : 1;
    c->cfg.ref_len = nullptr;
    c->cfg.ref_is_primary = nullptr;
    c->cfg.ref_bases = nullptr;
    c->device = cfg->device;

#define CTX_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) {                                                                \
            int rc_ = fail(nullptr, NGSQ_ERR_DEVICE, "%s failed: %s", #expr, hipGetErrorString(e_)); \
            ngsq_destroy(c);                                                                   \
            return rc_;                                                                        \
        }                                                                                      \
    } while (0)

    CTX_TRY(hipSetDevice(c->device));
    // NGSQ_BLOCKING_SYNC=1: a thread that waits for the device sleeps instead of spinning -- for hosts that give a process
    // few cores (`ngs qc --gpus N` sets it for its workers when the CPU quota leaves a worker fewer than six: the reader's
    // pread threads need the core the driving thread would spin on).  Refused once the device is in
    // use by this process: not an error.
    if (const char *e = getenv("NGSQ_BLOCKING_SYNC"))
        if (atoi(e)) (void)hipSetDeviceFlags(hipDeviceScheduleBlockingSync);
    hipDeviceProp_t prop;
    CTX_TRY(hipGetDeviceProperties(&prop, c->device));
    c->li.n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (cfg->stream) {
        c->stream = (hipStream_t)cfg->stream;
        c->own_stream = false;
    } else {
        CTX_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        c->own_stream = true;
    }
    CTX_TRY(hipEventCreateWithFlags(&c->copy_done, hipEventDisableTiming));

    // ---- counters block
    DeviceState &st = c->st;
    st.tlen_cap = c->cfg.tlen_cap;
    st.max_read_len = c->cfg.max_read_len;
    st.n_refs = nr;
    st.cov_cap = c->cfg.cov_cap;
    st.gc_seed = c->cfg.gc_seed;
    st.off_tlen = OFF_TLEN_HIST;
    // (the quality table comes LAST: it grows with the longest read met -- quality_scores.rs:18 keeps a map per
    // position, any length works -- and growing then leaves every other offset where it was)
    st.off_edits1 = (uint32_t)round_up(st.off_tlen + st.tlen_cap + 1, 8);
    st.off_edits2 = st.off_edits1 + (uint32_t)round_up(NGSQ_EDITS_BINS, 8);
    st.off_seen = st.off_edits2 + (uint32_t)round_up(NGSQ_EDITS_BINS, 8);
    st.off_eseen = st.off_seen + nr;
    st.off_qual = (uint32_t)round_up((uint64_t)st.off_eseen + nr, 8);
    c->n_counters = round_up((uint64_t)st.off_qual + (uint64_t)st.max_read_len * QUAL_BINS, 8);
    CTX_TRY(hipMalloc((void **)&st.counters, c->n_counters * 8));
    CTX_TRY(hipMemsetAsync(st.counters, 0, c->n_counters * 8, c->stream));
    c->h_counters.assign(c->n_counters, 0);

    // ---- per-sequence tables
    // (ref_bases_deferred: the bases come later, from a file -- ngsq_reference_load; every sequence gets Edits state and room)
    const bool deferred = (c->cfg.facets & NGSQ_FACET_EDITS) && cfg->ref_bases_deferred && !cfg->ref_bases;
    c->ref_deferred = deferred;
    std::vector<uint64_t> depth_off(nr, NO_DEPTH), edits_off(nr, NO_DEPTH), bases_off(nr, NO_DEPTH);
    std::vector<uint32_t> first_chunk(nr + 1, 0);
    uint64_t nd = 0, ne = 0, nbases = 0;
    c->bin_off.assign(nr + 1, 0);
    uint64_t max_len = 0;
    for (uint32_t r = 0; r < nr; r++) {
        const uint64_t L = c->ref_len[r];
        if (L > max_len) max_len = L;
        first_chunk[r] = (uint32_t)(nd / COV_CHUNK);
        if ((c->cfg.facets & NGSQ_FACET_COVERAGE) && c->primary[r]) {
            depth_off[r] = nd;
            nd += round_up(L + 2, COV_CHUNK); // chunks never straddle sequences
        }
        const uint64_t nb = 1 + L / c->cfg.bin_size + (L % c->cfg.bin_size != 0);
        c->bin_off[r + 1] = c->bin_off[r] + ((c->cfg.facets & NGSQ_FACET_COVERAGE) && c->primary[r] ? nb : 0);
        if ((c->cfg.facets & NGSQ_FACET_EDITS) && (deferred || (cfg->ref_bases && cfg->ref_bases[r]))) {
            edits_off[r] = ne;
            ne += round_up(2 * (L + 1), 4);
            bases_off[r] = nbases;
            nbases += round_up(L / 2 + 1 + 32, 16); // packed, two bases per byte; slack for the 16-byte loads past a read's last base
        }
    }
    first_chunk[nr] = (uint32_t)(nd / COV_CHUNK);
    c->depth_off = depth_off;
    c->edits_off = edits_off;
    c->bases_off = bases_off;
    c->nbases = nbases;
    // depth block = difference arrays | one sum per chunk | one sum per COV_SUPER chunks
    c->n_diff = nd;
    c->n_chunks = nd / COV_CHUNK;
    const uint64_t n_chunk_pad = round_up(c->n_chunks, 4), n_super_pad = round_up((c->n_chunks + COV_SUPER - 1) / COV_SUPER, 4);
    if (nd) nd += n_chunk_pad + n_super_pad;
    c->n_depth = nd;
    c->n_edits = ne;
    if (nr) {
        CTX_TRY(hipMalloc((void **)&c->d_ref_len, nr * 4));
        CTX_TRY(hipMemcpy(c->d_ref_len, c->ref_len.data(), nr * 4, hipMemcpyHostToDevice));
        CTX_TRY(hipMalloc((void **)&c->d_depth_off, nr * 8));
        CTX_TRY(hipMemcpy(c->d_depth_off, depth_off.data(), nr * 8, hipMemcpyHostToDevice));
        CTX_TRY(hipMalloc((void **)&c->d_edits_off, nr * 8));
        CTX_TRY(hipMemcpy(c->d_edits_off, edits_off.data(), nr * 8, hipMemcpyHostToDevice));
        CTX_TRY(hipMalloc((void **)&c->d_bases_off, nr * 8));
        CTX_TRY(hipMemcpy(c->d_bases_off, bases_off.data(), nr * 8, hipMemcpyHostToDevice));
        CTX_TRY(hipMalloc((void **)&c->d_first_chunk, (nr + 1) * 4));
        CTX_TRY(hipMemcpy(c->d_first_chunk, first_chunk.data(), (nr + 1) * 4, hipMemcpyHostToDevice));
        CTX_TRY(hipMalloc((void **)&c->d_bin_off, (nr + 1) * 8));
        CTX_TRY(hipMemcpy(c->d_bin_off, c->bin_off.data(), (nr + 1) * 8, hipMemcpyHostToDevice));
    }
    st.ref_len = c->d_ref_len;
    st.ref_depth_off = c->d_depth_off;
    st.ref_edits_off = c->d_edits_off;
    st.ref_bases_off = c->d_bases_off;
    if (nd) {
        CTX_TRY(hipMalloc((void **)&st.depth, nd * 4));
        CTX_TRY(hipMemsetAsync(st.depth, 0, nd * 4, c->stream));
        st.chunk_sums = st.depth + c->n_diff;
        st.super_sums = st.chunk_sums + n_chunk_pad;
        (void)max_len;
        c->n_cov_hist = (uint64_t)nr * (c->cfg.cov_cap + 2);
        c->h_cov_hist.assign(c->n_cov_hist, 0);
        c->h_bin_totals.assign(c->bin_off[nr] + 1, 0);
        c->scan_hi = c->n_chunks;
    }
    {   // teardown block: [depth histograms | bin totals | VAF histogram]
        const uint64_t n_bins = nd ? c->bin_off[nr] + 1 : 0;
        c->n_td = round_up(c->n_cov_hist + n_bins + NGSQ_VAF_BINS, 8);
        CTX_TRY(hipMalloc((void **)&c->d_td, c->n_td * 8));
        CTX_TRY(hipMemsetAsync(c->d_td, 0, c->n_td * 8, c->stream));
        c->d_cov_hist = c->d_td;
        c->d_bin_totals = c->d_td + c->n_cov_hist;
        c->d_vaf = c->d_td + c->n_cov_hist + n_bins;
        CTX_TRY(hipMalloc((void **)&c->d_touched, 16));
        CTX_TRY(hipMemcpyAsync(c->d_touched, c->h_touched, 16, hipMemcpyHostToDevice, c->stream));
        st.touched = c->d_touched;
    }
    c->stream_cov = c->cfg.sorted_input && nd;
    if (c->stream_cov) {
        const uint64_t nr4 = round_up(nr ? nr : 1, 4);
        CTX_TRY(hipMalloc((void **)&c->d_stream_u32, (7 * nr4 + 4) * 4));
        CTX_TRY(hipMalloc((void **)&c->d_last_key, 8));
        CTX_TRY(hipMalloc((void **)&c->d_chunk_flags, round_up(c->n_chunks ? c->n_chunks : 1, 16)));
        st.end_acc = c->d_stream_u32;
        st.batch_span = c->d_stream_u32 + 7 * nr4;
        CovStreamArgs &sa = c->csa;
        sa.prev_end = c->d_stream_u32 + nr4;
        sa.plan_a = c->d_stream_u32 + 2 * nr4;
        sa.plan_z = c->d_stream_u32 + 3 * nr4;
        sa.plan_h = c->d_stream_u32 + 4 * nr4;
        sa.plan_t = c->d_stream_u32 + 5 * nr4;
        sa.guard_until = c->d_stream_u32 + 6 * nr4;
        sa.last_key = c->d_last_key;
        sa.chunk_flags = c->d_chunk_flags;
        sa.hist = c->d_cov_hist;
        sa.bin_totals = c->d_bin_totals;
        sa.bin_off = c->d_bin_off;
        sa.bin_size = c->cfg.bin_size;
        sa.cov_cap = c->cfg.cov_cap;
        sa.head_guard = c->cfg.cov_head_guard;
        CTX_TRY(hipMemsetAsync(c->d_stream_u32, 0, (7 * nr4 + 4) * 4, c->stream));
        CTX_TRY(hipMemsetAsync(sa.plan_a, 0xFF, nr4 * 4, c->stream));
        CTX_TRY(hipMemsetAsync(c->d_last_key, 0, 8, c->stream));
        CTX_TRY(hipMemsetAsync(c->d_chunk_flags, 0, c->n_chunks ? c->n_chunks : 1, c->stream));
    }
    if (ne) {
        // (+ 64 bytes: the alts flush of k_edits_rows adds pairs of entries with 64-bit atomics; the pair that holds a sequence's
        // last entry may reach one entry further -- adding zero there)
        CTX_TRY(hipMalloc((void **)&st.edits, ne * 4 + 64));
        CTX_TRY(hipMemsetAsync(st.edits, 0, ne * 4 + 64, c->stream));
        // the reference, packed 4-bit, twice (edits_kernel.hip): [copy from base 0 | copy from base 1]
        uint8_t *bases = nullptr;
        // (+ 256: the window lanes of k_edits_rows read 16 bytes at up to 80 + 16 bytes behind a read's last compared base)
        CTX_TRY(hipMalloc((void **)&bases, 2 * nbases + 256));
        CTX_TRY(hipMemsetAsync(bases, 0, 2 * nbases + 256, c->stream));
        st.ref_bases = bases;
        st.ref_bases_odd = bases + nbases;
        c->d_ref_bases = bases;
        uint8_t *codes = nullptr;
        unsigned long long *d_bad = nullptr, h_bad = 0;
        CTX_TRY(hipMalloc((void **)&codes, (deferred ? 0 : max_len) + 64));
        CTX_TRY(hipMalloc((void **)&d_bad, 8));
        CTX_TRY(hipMemsetAsync(d_bad, 0, 8, c->stream));
        uint64_t n_carry = 0;
        c->edits_carry_off.assign(nr, 0);
        std::vector<uint32_t> have(2 * (size_t)nr, 0); // bases of each sequence: what a read may reach | the fast paths' bound
        bool lens_differ = false;
        for (uint32_t r = 0; r < nr; r++) {
            if (bases_off[r] == NO_DEPTH) continue;
            const uint64_t L = c->ref_len[r];
            // (ref_bases_len: the FASTA's sequence may be shorter or longer than @SQ LN says -- edits.rs:257-261 slices the FASTA's)
            const uint64_t have_r = cfg->ref_bases_len && !deferred ? std::min<uint64_t>(cfg->ref_bases_len[r], L) : L;
            have[nr + r] = (uint32_t)have_r;
            have[r] = cfg->ref_bases_len && !deferred ? cfg->ref_bases_len[r] : (uint32_t)L; // (the FASTA's own length: a read may end beyond LN inside it)
            lens_differ = lens_differ || have[r] != L;
            if (!deferred && have_r) {
                CTX_TRY(hipMemcpyAsync(codes, cfg->ref_bases[r], have_r, hipMemcpyHostToDevice, c->stream));
                CTX_TRY(launch_pack_reference(c->li, codes, have_r, bases + bases_off[r], bases + nbases + bases_off[r], L / 2 + 1, d_bad, c->stream));
                CTX_TRY(hipStreamSynchronize(c->stream)); // `codes` is reused (and the host buffer is the caller's)
            }
            c->edits_carry_off[r] = n_carry;
            n_carry += edits_teardown_carry_words(L + 1);
        }
        if (lens_differ) {
            CTX_TRY(hipMalloc((void **)&c->d_edits_len, have.size() * 4));
            CTX_TRY(hipMemcpy(c->d_edits_len, have.data(), have.size() * 4, hipMemcpyHostToDevice));
            st.ref_edits_len = c->d_edits_len;
            st.ref_fast_len = c->d_edits_len + nr;
        }
        CTX_TRY(hipStreamSynchronize(c->stream)); // (d_bad was zeroed on this stream, which the copy below does not wait for by itself)
        CTX_TRY(hipMemcpy(&h_bad, d_bad, 8, hipMemcpyDeviceToHost));
        (void)hipFree(codes);
        (void)hipFree(d_bad);
        if (h_bad) {
            ngsq_destroy(c);
            return fail(nullptr, NGSQ_ERR_INVALID_ARGUMENT, "ngsq_config.ref_bases: every byte must be a 4-bit BAM base code (0..15)");
        }
        CTX_TRY(hipMalloc((void **)&c->d_edits_carry, (n_carry + 4) * 4));
        c->edits_conv_lo.assign(nr, 0);
        c->edits_conv_hi.assign(nr, 0);
    }
    c->h_vaf.assign(NGSQ_VAF_BINS, 0);
    for (int k = 0; k < K_COUNT; k++) c->timing[k] = {KERNEL_NAMES[k], 0, 0.0, 0};
    CTX_TRY(hipStreamSynchronize(c->stream));
#undef CTX_TRY
    *out = c;
    return NGSQ_OK;
}

void ngsq_destroy(ngsq_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    ngsq::reference_abandon(c); // (a reference load still on its way writes into this context's buffers)
    (void)ngsq::pool_trim(); // blocks a finished device ingest left for the next file (mem_pool.h)
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    (void)hipFree(c->d_ft_idx);
    (void)hipFree(c->d_ft_starts);
    (void)hipFree(c->d_ft_stops);
    (void)hipFree(c->d_ft_primary);
    for (const ngsq_ctx::DeferredFeatures &d : c->ft_deferred) (void)hipFree(d.buf);
    for (auto &p : c->pending) {
        (void)hipEventDestroy(p.a);
        (void)hipEventDestroy(p.b);
    }
    for (auto &ev : c->event_pool) (void)hipEventDestroy(ev);
    for (int k = 0; k < 2; k++) {
        if (c->stage[k].buf) (void)hipFree(c->stage[k].buf);
        if (c->stage[k].done) (void)hipEventDestroy(c->stage[k].done);
    }
    if (c->copy_done) (void)hipEventDestroy(c->copy_done);
    (void)hipFree(c->st.counters);
    (void)hipFree(c->st.depth);
    (void)hipFree(c->st.edits);
    (void)hipFree(c->d_ref_bases);
    (void)hipFree(c->d_edits_len);
    (void)hipFree(c->d_bad_off);
    (void)hipFree(c->d_bad_pos);
    (void)hipFree(c->d_edits_carry);
    (void)hipFree(c->d_edits_td);
    (void)hipFree(c->d_edits_defer);
    (void)hipFree(c->d_ref_len);
    (void)hipFree(c->d_depth_off);
    (void)hipFree(c->d_edits_off);
    (void)hipFree(c->d_bases_off);
    (void)hipFree(c->d_first_chunk);
    (void)hipFree(c->d_bin_off);
    (void)hipFree(c->d_td);
    (void)hipFree(c->d_touched);
    (void)hipFree(c->d_cov_end);
    (void)hipFree(c->d_stream_u32);
    (void)hipFree(c->d_last_key);
    (void)hipFree(c->d_chunk_flags);
    if (c->pin_results) (void)hipHostFree(c->pin_results);
    if (c->xchg_scratch) ngsq::free_exchange_scratch(c->xchg_scratch);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

// ---- kernel timing brackets -------------------------------------------------

static hipEvent_t get_event(ngsq_ctx *c) {
    if (!c->event_pool.empty()) {
        hipEvent_t e = c->event_pool.back();
        c->event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

ngsq::KernelTimer::KernelTimer(ngsq_ctx *c_, int id_, uint64_t bytes, hipStream_t stream) : c(c_), id(id_), s(stream ? stream : c_->stream) {
    if (roctx().push) (void)roctx().push(KERNEL_NAMES[id]);
    c->timing[id].launches += 1;
    c->timing[id].algo_bytes += bytes;
    if (c->cfg.timing) {
        a = get_event(c);
        b = get_event(c);
        (void)hipEventRecord(a, s);
    }
}
ngsq::KernelTimer::~KernelTimer() {
    if (roctx().pop) (void)roctx().pop();
    if (c->cfg.timing) {
        (void)hipEventRecord(b, s);
        c->pending.push_back({id, a, b});
    }
}
typedef ngsq::KernelTimer Bracket;

static void resolve_timing(ngsq_ctx *c) {
    for (auto &p : c->pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) c->timing[p.id].total_ms += ms;
        c->event_pool.push_back(p.a);
        c->event_pool.push_back(p.b);
    }
    c->pending.clear();
}

// ---- batches ----------------------------------------------------------------

// The per-cycle quality table has one row per cycle of the longest read met so far (the reference: a map entry per
// position, quality_scores.rs:37-49).  Rows are added by moving the counters block into a larger allocation; nothing but
// the block's size changes (the table is its last part).
// exact: to `rows` rows, not beyond (ngsq_exchange: every shard takes the size of the largest table among them -- a size this policy
// produced on some rank; growing "by half again" from it would leave the ranks with tables of two sizes: found by the eight-rank
// file test of round 6 on a 300 kb read, refused by the exchange's layout check instead of being summed misaligned)
static int grow_rows(ngsq_ctx *c, uint64_t rows, bool exact = false) {
    if (rows <= c->st.max_read_len) return NGSQ_OK;
    if (rows > NGSQ_QUALITY_ROWS_LIMIT)
        return fail(c, NGSQ_ERR_LIMIT, "implementation limit: a read of %llu bases (the quality table holds up to %u cycles)",
                    (unsigned long long)rows, (unsigned)NGSQ_QUALITY_ROWS_LIMIT);
    // short reads: to the next multiple of 64 (what the fast kernels are specialised for depends on the rows); long reads: by
    // half again, so that a file of ever longer reads does not move the block for each of them
    const uint64_t want = exact ? rows : std::min<uint64_t>(rows <= 1024 ? round_up(rows, 64) : round_up(rows + rows / 2, 1024), NGSQ_QUALITY_ROWS_LIMIT);
    const uint64_t n_new = round_up((uint64_t)c->st.off_qual + want * QUAL_BINS, 8);
    unsigned long long *bigger = nullptr;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMalloc((void **)&bigger, n_new * 8));
    HIP_TRY(c, hipMemcpyAsync(bigger, c->st.counters, c->n_counters * 8, hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(c, hipMemsetAsync(bigger + c->n_counters, 0, (n_new - c->n_counters) * 8, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream)); // kernels queued earlier still use the old block
    (void)hipFree(c->st.counters);
    c->st.counters = bigger;
    c->st.max_read_len = (uint32_t)want;
    c->cfg.max_read_len = (uint32_t)want;
    c->n_counters = n_new;
    c->h_counters.resize(n_new, 0);
    return NGSQ_OK;
}

// the longest read of a batch, where the host can tell: fixed-pitch rows (the pitch), host columns (l_seq), or the
// reader's word for it (ngsq_batch.max_l_seq); 0 = unknown
static uint64_t batch_longest_read(const ngsq_batch *b) {
    if (!b->qual) return 0;
    if (!b->qual_off) return b->qual_stride;
    if (b->max_l_seq) return b->max_l_seq;
    if (b->location != NGSQ_MEM_HOST) return 0;
    uint64_t longest = 0;
    for (uint64_t i = 0; i < b->n_records; i++) longest = std::max<uint64_t>(longest, b->qual_off[i + 1] - b->qual_off[i]);
    return longest;
}

static int check_batch(ngsq_ctx *c, const ngsq_batch *b, uint32_t facets) {
    if (b->struct_size != sizeof(ngsq_batch))
        return fail(c, NGSQ_ERR_INVALID_ARGUMENT, "ngsq_batch.struct_size %u != %zu", b->struct_size,
                    sizeof(ngsq_batch));
    if (b->location != NGSQ_MEM_HOST && b->location != NGSQ_MEM_DEVICE)
        return fail(c, NGSQ_ERR_INVALID_ARGUMENT, "bad batch location %u", b->location);
    if (!b->n_records) return NGSQ_OK;
    if (!b->flag) return fail(c, NGSQ_ERR_INVALID_ARGUMENT, "flag column is null");
    if (b->location == NGSQ_MEM_DEVICE) {
        const void *cols[] = {b->flag, b->mapq, b->ref_id, b->pos, b->mate_ref_id, b->tlen, b->l_seq, b->n_cigar, b->cigar};
        for (const void *p : cols)
            if (((uintptr_t)p & 15) != 0)
                return fail(c, NGSQ_ERR_INVALID_ARGUMENT, "device batch columns must be 16-byte aligned");
    }
    if ((facets & NGSQ_FACET_GENERAL) && (!b->mapq || !b->ref_id || !b->mate_ref_id || !b->n_cigar))
        return fail(c, NGSQ_ERR_INVALID_ARGUMENT, "General needs mapq, ref_id, mate_ref_id, n_cigar");
    if ((facets & NGSQ_FACET_FEATURES) && (!b->ref_id || !b->pos || !b->n_cigar))
        return fail(c, NGSQ_ERR_INVALID_ARGUMENT, "Genomic Features needs flag, ref_id, pos, n_cigar (+ cigar)");
    if ((facets & NGSQ_FACET_TEMPLATE_LENGTH) && !b->tlen)
        return fail(c, NGSQ_ERR_INVALID_ARGUMENT, "Template Length needs tlen");
    if ((facets & (NGSQ_FACET_GC_CONTENT | NGSQ_FACET_EDITS)) && (!b->l_seq || !b->seq))
        return fail(c, NGSQ_ERR_INVALID_ARGUMENT, "GC Content / Edits need l_seq and seq");
    if ((facets & NGSQ_FACET_GC_CONTENT) && b->location == NGSQ_MEM_DEVICE && b->seq_off && !b->seq_bytes) {
        // 0 is also what a batch of reads without bases has (found by the parity sweep: a slice of records with l_seq 0, uploaded):
        // the offsets say which -- eight bytes read back, on this path only
        uint64_t end = 0;
        if (hipMemcpy(&end, b->seq_off + b->n_records, sizeof end, hipMemcpyDeviceToHost) != hipSuccess || end != 0)
            return fail(c, NGSQ_ERR_INVALID_ARGUMENT, "device batch with seq_off needs seq_bytes");
    }
    if ((facets & NGSQ_FACET_QUALITY_SCORE) && (!b->qual || (!b->qual_off && !b->l_seq)))
        return fail(c, NGSQ_ERR_INVALID_ARGUMENT, "Quality Score needs qual (+ l_seq or qual_off)");
    if ((facets & (NGSQ_FACET_COVERAGE | NGSQ_FACET_EDITS)) && (!b->ref_id || !b->pos || !b->n_cigar))
        return fail(c, NGSQ_ERR_INVALID_ARGUMENT, "Coverage / Edits need ref_id, pos, n_cigar");
    return NGSQ_OK;
}

// bump allocator over one staging buffer
struct Bump {
    uint8_t *base;
    uint64_t off = 0;
    explicit Bump(uint8_t *b) : base(b) {}
    uint8_t *take(uint64_t n) {
        uint8_t *p = base + off;
        off += round_up(n ? n : 1, 256);
        return p;
    }
};

struct ColumnSizes {
    uint64_t seq_bytes, qual_bytes, cigar_ops;
};

static int column_sizes(ngsq_ctx *c, const ngsq_batch *b, ColumnSizes *cs) {
    const uint64_t n = b->n_records;
    const bool host = b->location == NGSQ_MEM_HOST;
    cs->seq_bytes = b->seq_bytes;
    cs->qual_bytes = b->qual_bytes;
    cs->cigar_ops = b->cigar_ops;
    if (!b->seq) cs->seq_bytes = 0;
    else if (!b->seq_off) cs->seq_bytes = n * b->seq_stride;
    else if (host) cs->seq_bytes = b->seq_off[n];
    if (!b->qual) cs->qual_bytes = 0;
    else if (!b->qual_off) cs->qual_bytes = n * b->qual_stride;
    else if (host) cs->qual_bytes = b->qual_off[n];
    if (!b->cigar) cs->cigar_ops = 0;
    else if (!b->cigar_off) cs->cigar_ops = n * b->cigar_stride;
    else if (host) cs->cigar_ops = b->cigar_off[n];
    (void)c;
    return NGSQ_OK;
}

static int launch_all(ngsq_ctx *c, const DeviceBatch &db, const ColumnSizes &cs, uint32_t pass_mask) {
    const uint32_t facets = c->cfg.facets;
    const uint64_t n = db.n;
    const uint32_t rec_f = (pass_mask & NGSQ_PASS_RECORD) ? (facets & NGSQ_FACETS_RECORD_BASED) : 0;
    const uint32_t seq_f = (pass_mask & NGSQ_PASS_SEQUENCE) ? (facets & NGSQ_FACETS_SEQUENCE_BASED) : 0;
    if ((rec_f & (NGSQ_FACET_GENERAL | NGSQ_FACET_TEMPLATE_LENGTH)) || (seq_f & NGSQ_FACET_COVERAGE)) {
        const bool walk = (rec_f & NGSQ_FACET_GENERAL) || (seq_f & NGSQ_FACET_COVERAGE);
        const bool cov = (seq_f & NGSQ_FACET_COVERAGE) != 0;
        if (cov && c->stream_cov) { // scratch column of the streaming pass
            const uint64_t need = round_up(n, 1024) + 1024;
            if (c->cov_end_cap < need) {
                HIP_TRY(c, hipStreamSynchronize(c->stream));
                (void)hipFree(c->d_cov_end);
                c->d_cov_end = nullptr;
                c->cov_end_cap = 0;
                HIP_TRY(c, hipMalloc((void **)&c->d_cov_end, need * 4));
                c->cov_end_cap = need;
                c->st.cov_end = c->d_cov_end;
            }
            // the batch's largest-span word: two words used in turn, each zeroed by the OTHER batch's k_cov_plan_refs (both by
            // ngsq_reset) -- until round 4 a 4-byte memset per batch
            uint32_t *const spans = c->d_stream_u32 + 7 * round_up(c->st.n_refs ? c->st.n_refs : 1, 4);
            c->st.batch_span = spans + (c->span_turn & 1);
            c->csa.span_next = spans + ((c->span_turn & 1) ^ 1);
            c->span_turn += 1;
        }
        {
            Bracket br(c, K_FIELDS, n * 25 + (walk ? cs.cigar_ops * 4 : 0));
            HIP_TRY(c, launch_fields(c->li, c->st, db, rec_f, cov ? (c->stream_cov ? 2 : 1) : 0, c->stream));
        }
        if (cov && c->stream_cov) {
            // (on a stream of its own beside k_gc / k_qual it hides its 0.7 ms and costs k_qual_perm 1.2: DESIGN appendix A.8; confined to
            // 16-96 compute units by a CU mask, with or without the other kernels kept off them, 5.74-16.7 ms per step against 5.55: A.9)
            Bracket br(c, K_COV_STREAM, n * 8); // pos + cov_end of every record
            HIP_TRY(c, launch_cov_stream(c->li, c->st, db, c->csa, c->stream));
        }
    }
    // GC Content and Edits in one pass of the batch: k_edits_rows tallies the GC window from the sequence bytes it compares (the
    // column -- 75 of a record's 254 bytes -- is read once instead of twice; get_qc_facets builds both facets for one scan: qc.rs:44-126)
    const bool gc_in_edits = (rec_f & NGSQ_FACET_GC_CONTENT) && (seq_f & NGSQ_FACET_EDITS) && edits_can_take_gc(c->st, db);
    if ((rec_f & NGSQ_FACET_GC_CONTENT) && !gc_in_edits) {
        Bracket br(c, K_GC, n * 6 + cs.seq_bytes);
        HIP_TRY(c, launch_gc(c->li, c->st, db, cs.seq_bytes, c->stream));
    }
    if (rec_f & NGSQ_FACET_QUALITY_SCORE) {
        Bracket br(c, K_QUAL, cs.qual_bytes);
        HIP_TRY(c, launch_qual(c->li, c->st, db, c->stream));
    }
    if (rec_f & NGSQ_FACET_FEATURES) {
        // (on a stream of its own beside the other facets' kernels it hides nothing: 10.9 ms per all-facets pass either way, round 5)
        if (!c->have_features) {
            // the gene model is not there yet (the host is still reading the GFF): keep what the facet needs of these records --
            // 16 bytes each -- and look them up when it arrives (ngsq_set_features); ngsq_finalize fails if it never does
            ngsq_ctx::DeferredFeatures d{nullptr, 0, n};
            const uint64_t n16 = round_up(n, 16);
            HIP_TRY(c, ngsq::pool_device_alloc((void **)&d.buf, n16 * 16 + 256, &d.bytes));
            c->ft_deferred.push_back(d);
            Bracket br(c, K_FEATURES, n * 12 + cs.cigar_ops * 4);
            HIP_TRY(c, launch_features_defer(c->li, db, reinterpret_cast<uint16_t *>(d.buf + n16 * 12), reinterpret_cast<int32_t *>(d.buf),
                                             reinterpret_cast<int32_t *>(d.buf + n16 * 4), reinterpret_cast<uint16_t *>(d.buf + n16 * 14),
                                             reinterpret_cast<uint32_t *>(d.buf + n16 * 8), c->stream));
        } else {
            Bracket br(c, K_FEATURES, n * 12 + cs.cigar_ops * 4);
            HIP_TRY(c, launch_features(c->li, c->st, db, c->ft, c->stream));
        }
    }
    if (seq_f & NGSQ_FACET_EDITS) {
        if (c->ref_deferred && !c->ref_ready) {
            // the reference comes from a file (ngsq_reference_load): everything above has been queued meanwhile; the first Edits
            // kernel waits for the loader's last kernel (the loader thread synchronises its stream before it reports)
            if (!c->ref_loader) return fail(c, NGSQ_ERR_STATE, "ngsq_config.ref_bases_deferred is set but ngsq_reference_load was not called");
            const int rc = reference_join(c);
            if (rc != NGSQ_OK) return rc;
        }
        Bracket br(c, K_EDITS, n * 16 + cs.cigar_ops * 4 + cs.seq_bytes);
        const uint64_t words = (n + 63) / 64 + 16;
        if (c->edits_defer_cap < words) {
            HIP_TRY(c, hipStreamSynchronize(c->stream)); // (the previous batch's launches may still read the old block)
            (void)hipFree(c->d_edits_defer);
            c->d_edits_defer = nullptr;
            c->edits_defer_cap = 0;
            HIP_TRY(c, hipMalloc((void **)&c->d_edits_defer, (words + words / 4) * 8));
            c->edits_defer_cap = words + words / 4;
        }
        HIP_TRY(c, launch_edits(c->li, c->st, db, c->d_edits_defer, gc_in_edits, c->stream));
    }
    return NGSQ_OK;
}

int ngsq_set_features(ngsq_ctx *c, const ngsq_features *f) {
    if (!c || !f) return fail(c, NGSQ_ERR_INVALID_ARGUMENT, "null argument");
    if (f->struct_size != sizeof(ngsq_features))
        return fail(c, NGSQ_ERR_INVALID_ARGUMENT, "ngsq_features.struct_size %u != %zu", f->struct_size, sizeof(ngsq_features));
    if (f->n && (!f->ref_id || !f->name || !f->start || !f->stop)) return fail(c, NGSQ_ERR_INVALID_ARGUMENT, "null interval column");
    if (f->n > 0xFFFFFFF0ull) return fail(c, NGSQ_ERR_UNSUPPORTED, "too many feature intervals");
    for (int r = 0; r < 5; r++)
        if (f->role_name[r] >= 5) return fail(c, NGSQ_ERR_INVALID_ARGUMENT, "role_name[%d] = %u: name ids are < 5", r, f->role_name[r]);
    const uint32_t n_refs = c->cfg.n_refs;
    // bucket by (name id, sequence), then sort starts and stops of each bucket independently
    std::vector<uint32_t> idx((size_t)5 * n_refs + 1, 0);
    for (uint64_t i = 0; i < f->n; i++) {
        if (f->ref_id[i] >= n_refs || f->name[i] >= 5)
            return fail(c, NGSQ_ERR_INVALID_ARGUMENT, "feature interval %llu: sequence %u / name id %u out of range",
                        (unsigned long long)i, f->ref_id[i], f->name[i]);
        if (f->start[i] > f->stop[i])
            return fail(c, NGSQ_ERR_INVALID_ARGUMENT, "feature interval %llu: start %u > end %u", (unsigned long long)i,
                        f->start[i], f->stop[i]);
        idx[(size_t)f->name[i] * n_refs + f->ref_id[i] + 1] += 1;
    }
    for (size_t k = 1; k < idx.size(); k++) idx[k] += idx[k - 1];
    std::vector<uint32_t> starts(f->n + 1), stops(f->n + 1), fill(idx.begin(), idx.end() - 1);
    for (uint64_t i = 0; i < f->n; i++) {
        const uint32_t slot = fill[(size_t)f->name[i] * n_refs + f->ref_id[i]]++;
        starts[slot] = f->start[i];
        stops[slot] = f->stop[i];
    }
    {   // (a GENCODE-sized model -- 2.9 M intervals in 975 buckets -- took one thread 0.25 s of the command's wall clock: round 6)
        const size_t n_buckets = idx.size() - 1;
        std::atomic<size_t> next{0};
        auto work = [&]() {
            for (;;) {
                const size_t k = next.fetch_add(1);
                if (k >= 2 * n_buckets) return;
                std::vector<uint32_t> &v = k < n_buckets ? starts : stops;
                const size_t q = k < n_buckets ? k : k - n_buckets;
                std::sort(v.begin() + idx[q], v.begin() + idx[q + 1]);
            }
        };
        const unsigned hw = std::thread::hardware_concurrency();
        const size_t nt = f->n < 100000 ? 1 : std::min<size_t>(8, hw ? hw : 1);
        std::vector<std::thread> th;
        for (size_t t = 1; t < nt; t++) th.emplace_back(work);
        work();
        for (auto &t : th) t.join();
    }
    HIP_TRY(c, hipSetDevice(c->device));
    (void)hipFree(c->d_ft_idx);
    (void)hipFree(c->d_ft_starts);
    (void)hipFree(c->d_ft_stops);
    (void)hipFree(c->d_ft_primary);
    c->d_ft_idx = c->d_ft_starts = c->d_ft_stops = nullptr;
    c->d_ft_primary = nullptr;
    HIP_TRY(c, hipMalloc((void **)&c->d_ft_idx, idx.size() * 4));
    HIP_TRY(c, hipMalloc((void **)&c->d_ft_starts, starts.size() * 4));
    HIP_TRY(c, hipMalloc((void **)&c->d_ft_stops, stops.size() * 4));
    HIP_TRY(c, hipMalloc((void **)&c->d_ft_primary, n_refs + 1 + 16 + (ngsq::FT_SLOTS * 16 + 1) * 8)); // | the kernel's scratch (8-byte aligned)
    HIP_TRY(c, hipMemcpy(c->d_ft_idx, idx.data(), idx.size() * 4, hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->d_ft_starts, starts.data(), starts.size() * 4, hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->d_ft_stops, stops.data(), stops.size() * 4, hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->d_ft_primary, c->primary.data(), n_refs, hipMemcpyHostToDevice));
    c->ft.idx = c->d_ft_idx;
    c->ft.starts = c->d_ft_starts;
    c->ft.stops = c->d_ft_stops;
    c->ft.primary = c->d_ft_primary;
    c->ft.scratch = reinterpret_cast<unsigned long long *>(c->d_ft_primary + ((n_refs + 1 + 15) & ~15ull));
    HIP_TRY(c, hipMemset(c->ft.scratch, 0, (ngsq::FT_SLOTS * 16 + 1) * 8));
    c->ft.n_refs = n_refs;
    for (int r = 0; r < 5; r++) c->ft.role_name[r] = f->role_name[r];
    c->have_features = true;
    // the batches that came first
    for (const ngsq_ctx::DeferredFeatures &d : c->ft_deferred) {
        const uint64_t n16 = round_up(d.n, 16);
        DeviceBatch db{};
        db.n = d.n;
        db.ref_id = reinterpret_cast<const int32_t *>(d.buf);
        db.pos = reinterpret_cast<const int32_t *>(d.buf + n16 * 4);
        db.cigar = reinterpret_cast<const uint32_t *>(d.buf + n16 * 8);
        db.flag = reinterpret_cast<const uint16_t *>(d.buf + n16 * 12);
        db.n_cigar = reinterpret_cast<const uint16_t *>(d.buf + n16 * 14);
        db.cigar_stride = 1;
        Bracket br(c, K_FEATURES, d.n * 16);
        HIP_TRY(c, launch_features(c->li, c->st, db, c->ft, c->stream));
    }
    if (!c->ft_deferred.empty()) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        for (const ngsq_ctx::DeferredFeatures &d : c->ft_deferred) ngsq::pool_device_free(d.buf, d.bytes);
        c->ft_deferred.clear();
    }
    return NGSQ_OK;
}

int ngsq_process_batch(ngsq_ctx *c, const ngsq_batch *b, uint32_t pass_mask) {
    if (!c || !b) return fail(c, NGSQ_ERR_INVALID_ARGUMENT, "null argument");
    if (c->finalized) return fail(c, NGSQ_ERR_STATE, "context already finalized; call ngsq_reset");
    const uint32_t facets = c->cfg.facets;
    int rc = check_batch(c, b, facets);
    if (rc != NGSQ_OK) return rc;
    const uint64_t n = b->n_records;
    if (!n) return NGSQ_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    ColumnSizes cs;
    column_sizes(c, b, &cs);
    if ((facets & NGSQ_FACET_QUALITY_SCORE) && (pass_mask & NGSQ_PASS_RECORD)) {
        rc = grow_rows(c, batch_longest_read(b));
        if (rc != NGSQ_OK) return rc;
    }

    DeviceBatch db{};
    db.n = n;
    db.first_record_index = b->first_record_index;
    db.seq_stride = b->seq_stride;
    db.qual_stride = b->qual_stride;
    db.cigar_stride = b->cigar_stride;
    db.qual_bytes = cs.qual_bytes;

    if (b->location == NGSQ_MEM_DEVICE) {
        db.flag = b->flag;
        db.mapq = b->mapq;
        db.ref_id = b->ref_id;
        db.pos = b->pos;
        db.mate_ref_id = b->mate_ref_id;
        db.tlen = b->tlen;
        db.l_seq = b->l_seq;
        db.n_cigar = b->n_cigar;
        db.seq = b->seq;
        db.seq_off = b->seq_off;
        db.qual = b->qual;
        db.qual_off = b->qual_off;
        db.cigar = b->cigar;
        db.cigar_off = b->cigar_off;
        db.record_id = b->record_id;
        return launch_all(c, db, cs, pass_mask);
    }

    // host batch: copy the columns into one of two device staging buffers
    const uint64_t need = round_up(n * 2, 256) + round_up(n, 256) + 5 * round_up(n * 4, 256) +
                          round_up(n * 2, 256) + round_up(cs.seq_bytes + 16, 256) +
                          round_up(cs.qual_bytes + 16, 256) + round_up(cs.cigar_ops * 4 + 16, 256) +
                          3 * round_up((n + 1) * 8, 256) + round_up(n * 8, 256) + 4096;
    Staging &sg = c->stage[c->stage_next];
    c->stage_next ^= 1;
    if (sg.done) HIP_TRY(c, hipEventSynchronize(sg.done)); // kernels of its previous batch finished
    if (sg.cap < need) {
        if (sg.buf) HIP_TRY(c, hipFree(sg.buf));
        sg.buf = nullptr;
        sg.cap = 0;
        HIP_TRY(c, hipMalloc((void **)&sg.buf, need));
        sg.cap = need;
    }
    if (!sg.done) HIP_TRY(c, hipEventCreateWithFlags(&sg.done, hipEventDisableTiming));
    Bump bump(sg.buf);
    uint64_t copied = 0;
    {
        Bracket br(c, K_H2D, 0);
#define STAGE(field, type, count)                                                             \
    do {                                                                                      \
        if (b->field) {                                                                       \
            const uint64_t bytes_ = (uint64_t)(count) * sizeof(type);                         \
            type *d_ = (type *)bump.take(bytes_);                                             \
            if (bytes_) HIP_TRY(c, hipMemcpyAsync(d_, b->field, bytes_, hipMemcpyHostToDevice, c->stream)); \
            db.field = d_;                                                                    \
            copied += bytes_;                                                                 \
        }                                                                                     \
    } while (0)
        STAGE(flag, uint16_t, n);
        STAGE(mapq, uint8_t, n);
        STAGE(ref_id, int32_t, n);
        STAGE(pos, int32_t, n);
        STAGE(mate_ref_id, int32_t, n);
        STAGE(tlen, int32_t, n);
        STAGE(l_seq, uint32_t, n);
        STAGE(n_cigar, uint16_t, n);
        STAGE(seq, uint8_t, cs.seq_bytes);
        STAGE(seq_off, uint64_t, n + 1);
        STAGE(qual, uint8_t, cs.qual_bytes);
        STAGE(qual_off, uint64_t, n + 1);
        STAGE(cigar, uint32_t, cs.cigar_ops);
        STAGE(cigar_off, uint64_t, n + 1);
        STAGE(record_id, uint64_t, n);
#undef STAGE
        c->timing[K_H2D].algo_bytes += copied;
    }
    HIP_TRY(c, hipEventRecord(c->copy_done, c->stream));
    rc = launch_all(c, db, cs, pass_mask);
    if (rc != NGSQ_OK) return rc;
    HIP_TRY(c, hipEventRecord(sg.done, c->stream));
    // the caller may reuse its (possibly pinned) buffers once the copies have landed (NGSQ_PASS_NOWAIT: the caller waits itself)
    if (!(pass_mask & NGSQ_PASS_NOWAIT)) HIP_TRY(c, hipEventSynchronize(c->copy_done));
    return NGSQ_OK;
}

int ngsq_synchronize(ngsq_ctx *c) {
    if (!c) return NGSQ_ERR_INVALID_ARGUMENT;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    resolve_timing(c);
    return NGSQ_OK;
}

void *ngsq_stream(ngsq_ctx *c) { return c ? (void *)c->stream : nullptr; }

// every sequence that has Edits slots counts as written (the words at off_eseen of the counters block; kept non-zero ones)
static int mark_edits_written(ngsq_ctx *c) {
    const uint32_t nr = c->st.n_refs;
    if (!c->n_edits || !nr) return NGSQ_OK;
    std::vector<unsigned long long> eseen(nr);
    HIP_TRY(c, hipMemcpyAsync(eseen.data(), c->st.counters + c->st.off_eseen, nr * 8ull, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (uint32_t r = 0; r < nr; r++)
        if (c->edits_off[r] != NO_DEPTH && !eseen[r]) eseen[r] = 1;
    HIP_TRY(c, hipMemcpyAsync(c->st.counters + c->st.off_eseen, eseen.data(), nr * 8ull, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return NGSQ_OK;
}

int ngsq_teardown(ngsq_ctx *c) {
    if (!c) return NGSQ_ERR_INVALID_ARGUMENT;
    if (c->finalized || c->torn_down) return fail(c, NGSQ_ERR_STATE, "already torn down");
    if (!c->ft_deferred.empty())
        return fail(c, NGSQ_ERR_STATE, "NGSQ_FACET_FEATURES: batches were scanned but ngsq_set_features was never called");
    HIP_TRY(c, hipSetDevice(c->device));
    const uint32_t facets = c->cfg.facets;
    const uint32_t nr = c->st.n_refs;
    if (c->edits_uploaded) { // see ngsq_state_upload: every sequence with Edits slots counts as written
        const int rc = mark_edits_written(c);
        if (rc) return rc;
    }
    if ((facets & NGSQ_FACET_COVERAGE) && c->n_chunks) {
        // one launch tears down every sequence that has an entry (coverage.rs:187-193 skips the rest)
        CovScanArgs a{};
        a.depth = c->st.depth;
        a.n_chunks = c->n_chunks;
        a.c_begin = c->scan_lo;
        a.c_end = c->scan_hi;
        a.carry_in = c->scan_carry;
        a.carry_words = c->scan_words;
        a.carry_mask = c->scan_front;
        a.chunk_sums = c->st.chunk_sums;
        a.super_sums = c->st.super_sums;
        a.ref_first_chunk = c->d_first_chunk;
        a.ref_len = c->d_ref_len;
        a.seen = c->st.counters + c->st.off_seen;
        a.hist = c->d_cov_hist;
        a.bin_totals = c->d_bin_totals;
        a.bin_off = c->d_bin_off;
        a.n_refs = nr;
        a.bin_size = c->cfg.bin_size;
        a.cov_cap = c->cfg.cov_cap;
        a.reset = 1;
        a.chunk_flags = c->stream_cov ? c->d_chunk_flags : nullptr;
        Bracket br(c, K_COV_SCAN, 0);
        HIP_TRY(c, launch_cov_scan(c->li, a, c->stream));
    }
    if (facets & NGSQ_FACET_EDITS) {
        // The slot of refs holds the difference array of the `M` cover (edits_kernel.hip): sums of every 4096-entry chunk and of every
        // 256 chunks, then -- chunk by chunk, the carry from those sums -- the VAF histogram of the covered positions.  A sharded run
        // splits the chunks of every sequence evenly over the ranks (ngsq_exchange); the partial histograms are summed with the other
        // teardown results.  The slots keep the difference array until somebody asks for the positions (ngsq_get_edits_positions
        // converts then).  All sequences in THREE launches (round 6; sequences Edits wrote nothing for -- their word at off_eseen is
        // zero -- cost their blocks one load each): a real header has 195, and one launch per sequence and step was 585.
        std::vector<EditsSeq> seqs;
        std::vector<uint32_t> first[3];
        uint64_t nb[3] = {0, 0, 0};
        uint64_t algo = 0;
        for (uint32_t r = 0; r < nr; r++) {
            if (c->edits_off[r] == NO_DEPTH) continue;
            const uint64_t L1 = (uint64_t)c->ref_len[r] + 1, nc = edits_teardown_chunks(L1);
            const uint64_t c0 = nc * c->vaf_part / c->vaf_parts, c1 = nc * (c->vaf_part + 1) / c->vaf_parts;
            c->edits_conv_lo[r] = c->edits_conv_hi[r] = c0;
            EditsSeq e{};
            e.edits_off = c->edits_off[r];
            e.n_entries = L1;
            e.carry_off = c->edits_carry_off[r];
            e.chunk0 = (uint32_t)c0;
            e.chunk1 = (uint32_t)c1;
            e.ref = r;
            seqs.push_back(e);
            for (int k = 0; k < 3; k++) first[k].push_back((uint32_t)nb[k]);
            nb[0] += nc;
            nb[1] += (nc + 255) / 256; // (EDS = 256 chunks per super sum: edits_kernel.hip)
            nb[2] += std::min<uint64_t>(c1 - c0, 4096);
            algo += L1 * 4 + std::min<uint64_t>(L1, (c1 - c0) * 4096) * 8;
        }
        if (!seqs.empty()) {
            for (int k = 0; k < 3; k++) first[k].push_back((uint32_t)nb[k]);
            const size_t n_seq = seqs.size(), tab = (n_seq + 1) * 4, bytes = n_seq * sizeof(EditsSeq) + 3 * tab;
            if (c->edits_td_cap < bytes) {
                HIP_TRY(c, hipStreamSynchronize(c->stream));
                (void)hipFree(c->d_edits_td);
                c->d_edits_td = nullptr;
                c->edits_td_cap = 0;
                HIP_TRY(c, hipMalloc((void **)&c->d_edits_td, bytes));
                c->edits_td_cap = bytes;
            }
            // (the tables travel from a host vector that lives until the copies have been made: a pageable copy returns then)
            c->h_edits_td.resize(bytes);
            uint8_t *h = c->h_edits_td.data();
            memcpy(h, seqs.data(), n_seq * sizeof(EditsSeq));
            for (int k = 0; k < 3; k++) memcpy(h + n_seq * sizeof(EditsSeq) + (size_t)k * tab, first[k].data(), tab);
            HIP_TRY(c, hipMemcpyAsync(c->d_edits_td, h, bytes, hipMemcpyHostToDevice, c->stream));
            const EditsSeq *d_seqs = reinterpret_cast<const EditsSeq *>(c->d_edits_td);
            const uint32_t *d_first = reinterpret_cast<const uint32_t *>(c->d_edits_td + n_seq * sizeof(EditsSeq));
            Bracket br(c, K_EDITS_VAF, algo);
            HIP_TRY(c, launch_edits_teardown_all(d_seqs, (uint32_t)n_seq, d_first, (uint32_t)nb[0], d_first + (n_seq + 1), (uint32_t)nb[1], d_first + 2 * (n_seq + 1),
                                                 (uint32_t)nb[2], c->st.edits, c->d_edits_carry, c->d_vaf, c->st.counters + c->st.off_eseen, c->stream));
        }
    }
    c->torn_down = true;
    return NGSQ_OK;
}

int ngsq_finalize(ngsq_ctx *c) {
    if (!c) return NGSQ_ERR_INVALID_ARGUMENT;
    if (c->finalized) return fail(c, NGSQ_ERR_STATE, "already finalized");
    HIP_TRY(c, hipSetDevice(c->device));
    const uint32_t facets = c->cfg.facets;
    const uint32_t nr = c->st.n_refs;
    if (!c->torn_down) {
        int rc = ngsq_teardown(c);
        if (rc != NGSQ_OK) return rc;
    }
    // the integer results in ONE launch, written by the kernel into pinned host memory the device addresses (until round 4: five
    // device-to-host copies into pageable vectors, each a host round trip of its own)
    const bool cov = (facets & NGSQ_FACET_COVERAGE) && c->n_chunks;
    const uint64_t n_bins = cov ? c->bin_off[nr] + 1 : 0, n_hist = cov ? c->n_cov_hist : 0, n_vaf = (facets & NGSQ_FACET_EDITS) ? NGSQ_VAF_BINS : 0;
    const uint64_t need = c->n_counters + n_hist + n_bins + n_vaf + 2;
    if (step_legacy()) {
        if (cov) {
            HIP_TRY(c, hipMemcpyAsync(c->h_cov_hist.data(), c->d_cov_hist, c->n_cov_hist * 8, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipMemcpyAsync(c->h_bin_totals.data(), c->d_bin_totals, c->bin_off[nr] * 8 + 8, hipMemcpyDeviceToHost, c->stream));
        }
        if (n_vaf) HIP_TRY(c, hipMemcpyAsync(c->h_vaf.data(), c->d_vaf, NGSQ_VAF_BINS * 8, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipMemcpyAsync(c->h_touched, c->d_touched, 16, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipMemcpyAsync(c->h_counters.data(), c->st.counters, c->n_counters * 8, hipMemcpyDeviceToHost, c->stream));
    } else {
        if (c->pin_words < need) {
            if (c->pin_results) (void)hipHostFree(c->pin_results);
            c->pin_results = nullptr;
            c->pin_words = 0;
            HIP_TRY(c, hipHostMalloc((void **)&c->pin_results, (need + need / 4) * 8, hipHostMallocMapped));
            HIP_TRY(c, hipHostGetDevicePointer((void **)&c->pin_results_dev, c->pin_results, 0));
            c->pin_words = need + need / 4;
        }
        unsigned long long *d = c->pin_results_dev;
        StateSpans sp;
        sp.copy(d, c->st.counters, c->n_counters * 2);
        sp.copy(d + c->n_counters, c->d_cov_hist, n_hist * 2);
        sp.copy(d + c->n_counters + n_hist, c->d_bin_totals, n_bins * 2);
        sp.copy(d + c->n_counters + n_hist + n_bins, c->d_vaf, n_vaf * 2);
        sp.copy(d + c->n_counters + n_hist + n_bins + n_vaf, c->d_touched, 4);
        HIP_TRY(c, launch_state_spans(sp, c->stream));
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (!step_legacy()) {
        const unsigned long long *h = c->pin_results;
        memcpy(c->h_counters.data(), h, c->n_counters * 8);
        if (n_hist) memcpy(c->h_cov_hist.data(), h + c->n_counters, n_hist * 8);
        if (n_bins) memcpy(c->h_bin_totals.data(), h + c->n_counters + n_hist, n_bins * 8);
        if (n_vaf) memcpy(c->h_vaf.data(), h + c->n_counters + n_hist + n_bins, n_vaf * 8);
        memcpy(c->h_touched, h + c->n_counters + n_hist + n_bins + n_vaf, 16);
    }
    resolve_timing(c);
    c->finalized = true;
    if ((facets & NGSQ_FACET_COVERAGE) && c->n_chunks) { // algorithmic bytes of the scan: 8 B per torn-down position
        if (c->scan_partial) {
            c->timing[K_COV_SCAN].algo_bytes += (c->scan_hi - c->scan_lo) * (uint64_t)COV_CHUNK * 8;
        } else {
            for (uint32_t r = 0; r < nr; r++)
                if (c->depth_off[r] != NO_DEPTH && c->h_counters[c->st.off_seen + r])
                    c->timing[K_COV_SCAN].algo_bytes += ((uint64_t)c->ref_len[r] + 2) * 8;
        }
    }
    if (c->stream_cov && c->h_counters[C_COV_UNSORTED])
        return fail(c, NGSQ_ERR_UNSORTED,
                    "sorted_input was set but %llu adjacent record pair(s) are out of coordinate order; "
                    "create the context without sorted_input for such input",
                    c->h_counters[C_COV_UNSORTED]);
    const unsigned long long *err = c->h_counters.data() + C_ERR;
    {
        bool other = c->h_counters[C_FEAT_ERR_REF] || c->h_counters[C_FEAT_ERR_POS];
        for (int k = 0; k < 8; k++) other = other || (k != E_READ_TOO_LONG && err[k]);
        if (err[E_READ_TOO_LONG] && !other)
            return fail(c, NGSQ_ERR_LIMIT,
                        "implementation limit: %llu read(s) longer than the quality table's %u cycles in a batch that did not "
                        "announce them (ngsq_batch.max_l_seq; the reference has no such limit)",
                        err[E_READ_TOO_LONG], c->cfg.max_read_len);
    }
    for (int k = 0; k < 8; k++)
        if (err[k])
            return fail(c, NGSQ_ERR_MALFORMED_RECORD,
                        "malformed record(s): the reference would abort this run "
                        "(missing_ref_id=%llu bad_quality=%llu read_too_long=%llu edits_bad_ref=%llu "
                        "edits_record_short=%llu edits_not_consumed=%llu edits_too_many=%llu bad_cigar_op=%llu)",
                        err[0], err[1], err[2], err[3], err[4], err[5], err[6], err[7]);
    if (c->h_counters[C_FEAT_ERR_REF] || c->h_counters[C_FEAT_ERR_POS])
        return fail(c, NGSQ_ERR_MALFORMED_RECORD,
                    "malformed record(s): the reference would abort this run (Genomic Features: mapped records "
                    "without a reference sequence id=%llu, without an alignment start=%llu)",
                    c->h_counters[C_FEAT_ERR_REF], c->h_counters[C_FEAT_ERR_POS]);
    return NGSQ_OK;
}

int ngsq_depth_layout(ngsq_ctx *c, uint64_t *n_diff, uint64_t *n_chunks, uint64_t *touched_lo, uint64_t *touched_hi) {
    if (!c) return NGSQ_ERR_INVALID_ARGUMENT;
    HIP_TRY(c, hipSetDevice(c->device));
    unsigned long long t[2];
    HIP_TRY(c, hipMemcpyAsync(t, c->d_touched, 16, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (n_diff) *n_diff = c->n_diff;
    if (n_chunks) *n_chunks = c->n_chunks;
    if (touched_lo) *touched_lo = t[0] == ~0ull ? 0 : t[0];
    if (touched_hi) *touched_hi = t[0] == ~0ull ? 0 : t[1];
    return NGSQ_OK;
}

int ngsq_set_scan_range(ngsq_ctx *c, uint64_t chunk_lo, uint64_t chunk_hi, uint32_t carry_in) {
    if (!c) return NGSQ_ERR_INVALID_ARGUMENT;
    if (c->torn_down || c->finalized) return fail(c, NGSQ_ERR_STATE, "set the scan range before teardown");
    if (chunk_lo > chunk_hi || chunk_hi > c->n_chunks)
        return fail(c, NGSQ_ERR_INVALID_ARGUMENT, "scan range [%llu,%llu) outside [0,%llu)", (unsigned long long)chunk_lo,
                    (unsigned long long)chunk_hi, (unsigned long long)c->n_chunks);
    c->scan_lo = chunk_lo;
    c->scan_hi = chunk_hi;
    c->scan_carry = carry_in;
    c->scan_words = nullptr;
    c->scan_front = 0;
    c->scan_partial = !(chunk_lo == 0 && chunk_hi == c->n_chunks && carry_in == 0);
    return NGSQ_OK;
}

int ngsq_state_teardown(ngsq_ctx *c, void **p, uint64_t *n) {
    if (!c) return NGSQ_ERR_INVALID_ARGUMENT;
    *p = c->d_td;
    *n = c->n_td;
    return NGSQ_OK;
}

int ngsq_state_chunk_flags(ngsq_ctx *c, void **p, uint64_t *n) {
    if (!c) return NGSQ_ERR_INVALID_ARGUMENT;
    *p = c->d_chunk_flags;
    *n = c->stream_cov ? c->n_chunks : 0;
    return NGSQ_OK;
}

int ngsq_reset(ngsq_ctx *c) {
    if (!c) return NGSQ_ERR_INVALID_ARGUMENT;
    HIP_TRY(c, hipSetDevice(c->device));
    const uint64_t nr4 = round_up(c->st.n_refs ? c->st.n_refs : 1, 4);
    if (step_legacy()) {
        HIP_TRY(c, hipMemsetAsync(c->st.counters, 0, c->n_counters * 8, c->stream));
        if (c->n_depth && c->finalized) HIP_TRY(c, hipMemsetAsync(c->st.chunk_sums, 0, (c->n_depth - c->n_diff) * 4, c->stream));
        HIP_TRY(c, hipMemsetAsync(c->d_td, 0, c->n_td * 8, c->stream));
        if (c->stream_cov) {
            HIP_TRY(c, hipMemsetAsync(c->d_stream_u32, 0, (7 * nr4 + 4) * 4, c->stream));
            HIP_TRY(c, hipMemsetAsync(c->csa.plan_a, 0xFF, nr4 * 4, c->stream));
            HIP_TRY(c, hipMemsetAsync(c->d_last_key, 0, 8, c->stream));
            HIP_TRY(c, hipMemsetAsync(c->d_chunk_flags, 0, c->n_chunks ? c->n_chunks : 1, c->stream));
        }
    } else {
        // every small block in one launch (kernels.hip k_state_spans); the two large ones below stay memsets
        StateSpans sp;
        sp.fill(c->st.counters, c->n_counters * 2, 0);
        if (c->n_depth && c->finalized) sp.fill(c->st.chunk_sums, c->n_depth - c->n_diff, 0);
        sp.fill(c->d_td, c->n_td * 2, 0);
        if (c->stream_cov) {
            sp.fill(c->d_stream_u32, 2 * nr4, 0);            // end_acc | prev_end
            sp.fill(c->csa.plan_a, nr4, 0xFFFFFFFFu);        // CS_NONE
            sp.fill(c->csa.plan_z, 4 * nr4 + 4, 0);          // plan_z | plan_h | plan_t | guard_until | the two largest-span words
            sp.fill(c->d_last_key, 2, 0);
            sp.fill(c->d_chunk_flags, (c->n_chunks + 3) / 4, 0);
        }
        sp.fill(c->d_touched, 2, 0xFFFFFFFFu);               // {~0, 0}
        sp.fill(c->d_touched + 1, 2, 0);
        HIP_TRY(c, launch_state_spans(sp, c->stream));
    }
    if (c->n_depth) {
        if (!c->finalized) {
            HIP_TRY(c, hipMemsetAsync(c->st.depth, 0, c->n_depth * 4, c->stream));
        } else if (c->scan_partial && c->h_touched[0] != ~0ull && c->h_touched[1] > c->h_touched[0]) {
            // the scan zeroed the difference arrays behind itself; what is left are the chunk / super-chunk sums (above) and, after
            // a partial teardown, this shard's own entries outside the chunk range it tore down (h_touched: read back by ngsq_finalize)
            const uint64_t lo = c->h_touched[0], hi = c->h_touched[1] < c->n_diff ? c->h_touched[1] : c->n_diff;
            if (hi > lo) HIP_TRY(c, hipMemsetAsync(c->st.depth + lo, 0, (hi - lo) * 4, c->stream));
        }
    }
    if (c->n_edits) {
        if (!c->finalized || c->edits_uploaded) { // (after an upload the host's words say nothing certain about the slots)
            HIP_TRY(c, hipMemsetAsync(c->st.edits, 0, c->n_edits * 4, c->stream));
        } else { // only the sequences Edits wrote something for (h_counters: read back by ngsq_finalize)
            for (uint32_t r = 0; r < c->st.n_refs; r++)
                if (c->edits_off[r] != NO_DEPTH && c->h_counters[c->st.off_eseen + r])
                    HIP_TRY(c, hipMemsetAsync(c->st.edits + c->edits_off[r], 0, round_up(2 * ((uint64_t)c->ref_len[r] + 1), 4) * 4, c->stream));
        }
    }
    c->h_touched[0] = ~0ull;
    c->h_touched[1] = 0;
    if (step_legacy()) HIP_TRY(c, hipMemcpyAsync(c->d_touched, c->h_touched, 16, hipMemcpyHostToDevice, c->stream));
    for (const ngsq_ctx::DeferredFeatures &d : c->ft_deferred) ngsq::pool_device_free(d.buf, d.bytes); // (frees behind a device sync)
    c->ft_deferred.clear();
    c->span_turn = 0;
    c->finalized = false;
    c->edits_uploaded = false;
    c->torn_down = false;
    c->scan_lo = 0;
    c->scan_hi = c->n_chunks;
    c->scan_carry = 0;
    c->scan_partial = false;
    c->scan_words = nullptr;
    c->scan_front = 0;
    c->vaf_part = 0;
    c->vaf_parts = 1;
    return NGSQ_OK;
}

// ---- getters ------------------------------------------------------------------

#define NEED_FINAL(c)                                                                         \
    do {                                                                                      \
        if (!(c)) return NGSQ_ERR_INVALID_ARGUMENT;                                           \
        if (!(c)->finalized) return NGSQ_ERR_STATE;                                           \
    } while (0)

int ngsq_get_error_counts(const ngsq_ctx *c, ngsq_error_counts *out) {
    NEED_FINAL(c);
    static_assert(sizeof(ngsq_error_counts) == 10 * 8, "layout");
    memcpy(out, c->h_counters.data() + C_ERR, 8 * 8);
    out->features_missing_reference_id = c->h_counters[C_FEAT_ERR_REF];
    out->features_missing_position = c->h_counters[C_FEAT_ERR_POS];
    return NGSQ_OK;
}

int ngsq_get_features(const ngsq_ctx *c, ngsq_features_metrics *out) {
    NEED_FINAL(c);
    static_assert(sizeof(ngsq_features_metrics) == 9 * 8, "layout");
    memcpy(out, c->h_counters.data() + C_FEAT, sizeof *out);
    return NGSQ_OK;
}

int ngsq_get_general(const ngsq_ctx *c, ngsq_general_metrics *out) {
    NEED_FINAL(c);
    static_assert(sizeof(ngsq_general_metrics) == (16 + 18) * 8, "layout");
    memcpy(out, c->h_counters.data() + C_GENERAL, sizeof *out);
    return NGSQ_OK;
}

int ngsq_get_template_length(const ngsq_ctx *c, uint64_t *hist, size_t n_bins, uint64_t *processed,
                             uint64_t *ignored) {
    NEED_FINAL(c);
    if (n_bins < (size_t)c->st.tlen_cap + 1) return NGSQ_ERR_BUFFER_TOO_SMALL;
    memcpy(hist, c->h_counters.data() + c->st.off_tlen, ((size_t)c->st.tlen_cap + 1) * 8);
    *processed = c->h_counters[C_TLEN_PROCESSED];
    *ignored = c->h_counters[C_TLEN_IGNORED];
    return NGSQ_OK;
}

int ngsq_get_gc_content(const ngsq_ctx *c, ngsq_gc_metrics *out) {
    NEED_FINAL(c);
    memcpy(out->histogram, c->h_counters.data() + OFF_GC_HIST, sizeof out->histogram);
    out->total_gc_count = c->h_counters[C_GC_GC];
    out->total_at_count = c->h_counters[C_GC_AT];
    out->total_other_count = c->h_counters[C_GC_OTHER];
    out->processed = c->h_counters[C_GC_PROCESSED];
    out->ignored_flags = c->h_counters[C_GC_IGN_FLAGS];
    out->ignored_too_short = c->h_counters[C_GC_IGN_SHORT];
    return NGSQ_OK;
}

int ngsq_get_quality_scores(const ngsq_ctx *c, uint64_t *scores, size_t n_rows) {
    NEED_FINAL(c);
    if (n_rows < c->st.max_read_len) return NGSQ_ERR_BUFFER_TOO_SMALL;
    memcpy(scores, c->h_counters.data() + c->st.off_qual, (size_t)c->st.max_read_len * QUAL_BINS * 8);
    return NGSQ_OK;
}

uint32_t ngsq_n_refs(const ngsq_ctx *c) { return c ? c->st.n_refs : 0; }
uint32_t ngsq_max_read_len(const ngsq_ctx *c) { return c ? c->st.max_read_len : 0; }
uint32_t ngsq_tlen_bins(const ngsq_ctx *c) { return c ? c->st.tlen_cap + 1 : 0; }
uint32_t ngsq_cov_bins(const ngsq_ctx *c) { return c ? c->st.cov_cap + 1 : 0; }

uint64_t ngsq_coverage_n_bins(const ngsq_ctx *c, uint32_t ref) {
    if (!c || ref >= c->st.n_refs) return 0;
    const uint64_t L = c->ref_len[ref], b = c->cfg.bin_size;
    return 1 + L / b + (L % b != 0);
}

int ngsq_get_coverage_sequence(const ngsq_ctx *c, uint32_t ref, int *seen, uint64_t *hist, size_t n_hist_bins,
                               uint64_t *ignored, uint64_t *bin_totals, size_t n_bins) {
    NEED_FINAL(c);
    if (ref >= c->st.n_refs) return NGSQ_ERR_INVALID_ARGUMENT;
    const bool has = (c->cfg.facets & NGSQ_FACET_COVERAGE) && c->depth_off[ref] != NO_DEPTH &&
                     c->h_counters[c->st.off_seen + ref] != 0;
    *seen = has ? 1 : 0;
    *ignored = 0;
    if (!has) return NGSQ_OK;
    const uint64_t nb = ngsq_coverage_n_bins(c, ref);
    if (n_hist_bins < (size_t)c->st.cov_cap + 1 || n_bins < nb) return NGSQ_ERR_BUFFER_TOO_SMALL;
    const unsigned long long *h = c->h_cov_hist.data() + (uint64_t)ref * (c->st.cov_cap + 2);
    memcpy(hist, h, ((size_t)c->st.cov_cap + 1) * 8);
    *ignored = h[c->st.cov_cap + 1];
    memcpy(bin_totals, c->h_bin_totals.data() + c->bin_off[ref], nb * 8);
    return NGSQ_OK;
}

int ngsq_get_coverage_nonsensical(const ngsq_ctx *c, uint64_t *n) {
    NEED_FINAL(c);
    *n = c->h_counters[C_COV_NONSENSICAL];
    return NGSQ_OK;
}

int ngsq_get_edits(const ngsq_ctx *c, uint64_t *r1, uint64_t *r2, size_t n_edit_bins, uint64_t *vaf,
                   size_t n_vaf_bins) {
    NEED_FINAL(c);
    if (n_edit_bins < NGSQ_EDITS_BINS || n_vaf_bins < NGSQ_VAF_BINS) return NGSQ_ERR_BUFFER_TOO_SMALL;
    memcpy(r1, c->h_counters.data() + c->st.off_edits1, NGSQ_EDITS_BINS * 8);
    memcpy(r2, c->h_counters.data() + c->st.off_edits2, NGSQ_EDITS_BINS * 8);
    memcpy(vaf, c->h_vaf.data(), NGSQ_VAF_BINS * 8);
    return NGSQ_OK;
}

int ngsq_get_edits_positions(ngsq_ctx *c, uint32_t ref, uint32_t *refs, uint32_t *alts, size_t n) {
    NEED_FINAL(c);
    if (ref >= c->cfg.n_refs || !refs || !alts) return fail(c, NGSQ_ERR_INVALID_ARGUMENT, "bad argument");
    if (!(c->cfg.facets & NGSQ_FACET_EDITS) || c->edits_off[ref] == NO_DEPTH)
        return fail(c, NGSQ_ERR_STATE, "sequence %u has no Edits state", ref);
    const size_t L1 = (size_t)c->ref_len[ref] + 1;
    if (n < L1) return NGSQ_ERR_BUFFER_TOO_SMALL;
    HIP_TRY(c, hipSetDevice(c->device));
    if (!c->h_counters[c->st.off_eseen + ref]) { // Edits wrote nothing for the sequence (its teardown was skipped: no chunk sums either)
        memset(refs, 0, L1 * 4);
        memset(alts, 0, L1 * 4);
        return NGSQ_OK;
    }
    uint32_t *base = c->st.edits + c->edits_off[ref];
    {   // the chunks this context's teardown did not turn into refs (a sharded run: the other ranks' slices), without the tally
        const uint64_t nc = edits_teardown_chunks(L1);
        const uint32_t *carry = c->d_edits_carry + c->edits_carry_off[ref];
        HIP_TRY(c, launch_edits_refs(base, base + L1, L1, carry, 0, c->edits_conv_lo[ref], nullptr, nullptr, true, c->stream));
        HIP_TRY(c, launch_edits_refs(base, base + L1, L1, carry, c->edits_conv_hi[ref], nc, nullptr, nullptr, true, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        c->edits_conv_lo[ref] = 0;
        c->edits_conv_hi[ref] = nc;
    }
    HIP_TRY(c, hipMemcpy(refs, base, L1 * 4, hipMemcpyDeviceToHost));
    HIP_TRY(c, hipMemcpy(alts, base + L1, L1 * 4, hipMemcpyDeviceToHost));
    return NGSQ_OK;
}

// ---- measurement -----------------------------------------------------------------

int ngsq_kernel_timing_count(const ngsq_ctx *c) { return c ? K_COUNT : 0; }

int ngsq_kernel_timing(const ngsq_ctx *c, int index, ngsq_kernel_time *out) {
    if (!c || index < 0 || index >= K_COUNT || !out) return NGSQ_ERR_INVALID_ARGUMENT;
    *out = c->timing[index];
    return NGSQ_OK;
}

int ngsq_kernel_timing_reset(ngsq_ctx *c) {
    if (!c) return NGSQ_ERR_INVALID_ARGUMENT;
    resolve_timing(c);
    for (int k = 0; k < K_COUNT; k++) c->timing[k] = {KERNEL_NAMES[k], 0, 0.0, 0};
    return NGSQ_OK;
}

// ---- multi-GPU exchange points ---------------------------------------------------

int ngsq_state_counters(ngsq_ctx *c, void **p, uint64_t *n) {
    if (!c) return NGSQ_ERR_INVALID_ARGUMENT;
    *p = c->st.counters;
    *n = c->n_counters;
    return NGSQ_OK;
}
int ngsq_state_depth(ngsq_ctx *c, void **p, uint64_t *n) {
    if (!c) return NGSQ_ERR_INVALID_ARGUMENT;
    *p = c->st.depth;
    *n = c->n_depth;
    return NGSQ_OK;
}
int ngsq_state_edits(ngsq_ctx *c, void **p, uint64_t *n) {
    if (!c) return NGSQ_ERR_INVALID_ARGUMENT;
    *p = c->st.edits;
    *n = c->n_edits;
    return NGSQ_OK;
}

static int state_block(ngsq_ctx *c, int which, void **p, uint64_t *bytes) {
    switch (which) {
    case 0: *p = c->st.counters; *bytes = c->n_counters * 8; return NGSQ_OK;
    case 1: *p = c->st.depth; *bytes = c->n_depth * 4; return NGSQ_OK;
    case 2: *p = c->st.edits; *bytes = c->n_edits * 4; return NGSQ_OK;
    case 3: *p = c->d_td; *bytes = c->n_td * 8; return NGSQ_OK;
    case 4: *p = c->d_chunk_flags; *bytes = c->stream_cov ? c->n_chunks : 0; return NGSQ_OK;
    default: return NGSQ_ERR_INVALID_ARGUMENT;
    }
}

int ngsq_state_download(ngsq_ctx *c, int which, void *dst, uint64_t n_bytes) {
    if (!c) return NGSQ_ERR_INVALID_ARGUMENT;
    void *p;
    uint64_t bytes;
    int rc = state_block(c, which, &p, &bytes);
    if (rc) return rc;
    if (n_bytes != bytes) return fail(c, NGSQ_ERR_BUFFER_TOO_SMALL, "state block is %llu bytes", (unsigned long long)bytes);
    HIP_TRY(c, hipSetDevice(c->device));
    if (bytes) HIP_TRY(c, hipMemcpyAsync(dst, p, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return NGSQ_OK;
}

int ngsq_state_upload(ngsq_ctx *c, int which, const void *src, uint64_t n_bytes) {
    if (!c) return NGSQ_ERR_INVALID_ARGUMENT;
    void *p;
    uint64_t bytes;
    int rc = state_block(c, which, &p, &bytes);
    if (rc) return rc;
    if (n_bytes != bytes) return fail(c, NGSQ_ERR_BUFFER_TOO_SMALL, "state block is %llu bytes", (unsigned long long)bytes);
    HIP_TRY(c, hipSetDevice(c->device));
    if (bytes) HIP_TRY(c, hipMemcpyAsync(p, src, bytes, hipMemcpyHostToDevice, c->stream));
    if (which == 1 && c->n_diff) { // uploaded differences may sit anywhere
        c->h_touched[0] = 0;
        c->h_touched[1] = c->n_diff;
        HIP_TRY(c, hipMemcpyAsync(c->d_touched, c->h_touched, 16, hipMemcpyHostToDevice, c->stream));
    }
    // uploaded cover / alts may sit on any sequence: the teardown, ngsq_get_edits_positions and the reset behind a finalize act
    // on the sequences whose "Edits wrote here" word (counters block, off_eseen) is set.  ngsq_teardown sets them all when an
    // edits block came in from outside (whatever the order of the uploads), the next ngsq_reset clears every slot (a
    // host that restored or merged only this block got an empty VAF histogram, and the next run inherited the data)
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (which == 2 && c->n_edits) {
        c->edits_uploaded = true;
        return mark_edits_written(c); // (now, for an exchange that sums the words before the teardown; again there, for a counters block uploaded later)
    }
    return NGSQ_OK;
}

// ---- device memory helpers -------------------------------------------------------

int ngsq_device_malloc(ngsq_ctx *c, uint64_t n, void **p) {
    if (!c || !p) return NGSQ_ERR_INVALID_ARGUMENT;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMalloc(p, n + NGSQ_DEVICE_COLUMN_SLACK)); // the kernels' 16-byte loads read past a column's last row (ngsq.h)
    return NGSQ_OK;
}
int ngsq_device_free(ngsq_ctx *c, void *p) {
    if (!c) return NGSQ_ERR_INVALID_ARGUMENT;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipFree(p));
    return NGSQ_OK;
}
int ngsq_memcpy_h2d(ngsq_ctx *c, void *d, const void *h, uint64_t n) {
    if (!c) return NGSQ_ERR_INVALID_ARGUMENT;
    HIP_TRY(c, hipSetDevice(c->device));
    if (n) HIP_TRY(c, hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return NGSQ_OK;
}
int ngsq_memcpy_d2h(ngsq_ctx *c, void *h, const void *d, uint64_t n) {
    if (!c) return NGSQ_ERR_INVALID_ARGUMENT;
    HIP_TRY(c, hipSetDevice(c->device));
    if (n) HIP_TRY(c, hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return NGSQ_OK;
}
int ngsq_host_malloc_pinned(uint64_t n, void **p) {
    if (!p) return NGSQ_ERR_INVALID_ARGUMENT;
    hipError_t e = hipHostMalloc(p, n ? n : 1, hipHostMallocDefault);
    if (e != hipSuccess) return fail(nullptr, NGSQ_ERR_DEVICE, "hipHostMalloc: %s", hipGetErrorString(e));
    return NGSQ_OK;
}
int ngsq_host_free_pinned(void *p) {
    hipError_t e = hipHostFree(p);
    if (e != hipSuccess) return fail(nullptr, NGSQ_ERR_DEVICE, "hipHostFree: %s", hipGetErrorString(e));
    return NGSQ_OK;
}

} // extern "C"

namespace ngsq {
int grow_quality_table(ngsq_ctx *c, uint64_t rows) { return grow_rows(c, rows, true); }
} // namespace ngsq
