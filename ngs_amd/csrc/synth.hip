// synth.hip -- synthetic record batches (include/ngsq_synth.h): the same pure
// functions of (seed, record index) evaluated by a host loop or by HIP kernels
// that write the SoA columns in place in HBM.
#include <hip/hip_runtime.h>

#include <thread>
#include <vector>

#include "../../include/ngsq_synth.h"
#include "bam_reader.h"
#include "context.h"

namespace {

struct Cols {
    uint16_t *flag;
    uint8_t *mapq;
    int32_t *ref_id;
    int32_t *pos;
    int32_t *mate_ref_id;
    int32_t *tlen;
    uint32_t *l_seq;
    uint16_t *n_cigar;
    uint8_t *seq;
    const uint64_t *seq_off;
    uint8_t *qual;
    const uint64_t *qual_off;
    uint32_t *cigar;
    const uint64_t *cigar_off;
    uint32_t seq_stride, qual_stride, cigar_stride;
};

Cols cols_of(const ngsq_batch *b) {
    Cols c;
    c.flag = const_cast<uint16_t *>(b->flag);
    c.mapq = const_cast<uint8_t *>(b->mapq);
    c.ref_id = const_cast<int32_t *>(b->ref_id);
    c.pos = const_cast<int32_t *>(b->pos);
    c.mate_ref_id = const_cast<int32_t *>(b->mate_ref_id);
    c.tlen = const_cast<int32_t *>(b->tlen);
    c.l_seq = const_cast<uint32_t *>(b->l_seq);
    c.n_cigar = const_cast<uint16_t *>(b->n_cigar);
    c.seq = const_cast<uint8_t *>(b->seq);
    c.seq_off = b->seq_off;
    c.qual = const_cast<uint8_t *>(b->qual);
    c.qual_off = b->qual_off;
    c.cigar = const_cast<uint32_t *>(b->cigar);
    c.cigar_off = b->cigar_off;
    c.seq_stride = b->seq_stride;
    c.qual_stride = b->qual_stride;
    c.cigar_stride = b->cigar_stride;
    return c;
}

__host__ __device__ inline void fill_fixed_fields(const ngsq_synth_config &cfg, const Cols &c, uint64_t first,
                                                  uint64_t i) {
    ngsq_synth_record r;
    ngsq_synth_record_at(&cfg, first + i, &r);
    c.flag[i] = r.flag;
    c.mapq[i] = r.mapq;
    c.ref_id[i] = r.ref_id;
    c.pos[i] = r.pos;
    c.mate_ref_id[i] = r.mate_ref_id;
    c.tlen[i] = r.tlen;
    c.l_seq[i] = r.l_seq;
    c.n_cigar[i] = r.n_cigar;
    const uint64_t cb = c.cigar_off ? c.cigar_off[i] : i * (uint64_t)c.cigar_stride;
    const uint32_t room = c.cigar_off ? r.n_cigar : c.cigar_stride;
    for (uint32_t k = 0; k < room && k < NGSQ_SYNTH_MAX_OPS; k++) c.cigar[cb + k] = k < r.n_cigar ? r.cigar[k] : 0u;
}

// one thread per record: fixed-width fields + cigar
__global__ __launch_bounds__(256) void k_synth_fields(ngsq_synth_config cfg, Cols c, uint64_t first, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) fill_fixed_fields(cfg, c, first, i);
}

// FIXED mode: one thread per output dword of the dense seq / qual streams
__global__ __launch_bounds__(256) void k_synth_seq_fixed(ngsq_synth_config cfg, uint8_t *seq, uint32_t stride,
                                                         uint64_t first, uint64_t n_bytes) {
    const uint64_t stride_t = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w * 4 < n_bytes; w += stride_t) {
        uint32_t out = 0;
        for (uint32_t k = 0; k < 4; k++) {
            const uint64_t byte = w * 4 + k;
            if (byte >= n_bytes) break;
            const uint64_t rec = byte / stride;
            const uint32_t j = (uint32_t)(byte - rec * stride);
            out |= (uint32_t)ngsq_synth_seq_byte(&cfg, first + rec, cfg.read_len, j) << (8 * k);
        }
        if (w * 4 + 3 < n_bytes)
            *reinterpret_cast<uint32_t *>(seq + w * 4) = out;
        else
            for (uint32_t k = 0; w * 4 + k < n_bytes; k++) seq[w * 4 + k] = (uint8_t)(out >> (8 * k));
    }
}

__global__ __launch_bounds__(256) void k_synth_qual_fixed(ngsq_synth_config cfg, uint8_t *qual, uint32_t stride,
                                                          uint64_t first, uint64_t n_bytes) {
    const uint64_t stride_t = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w * 4 < n_bytes; w += stride_t) {
        uint32_t out = 0;
        for (uint32_t k = 0; k < 4; k++) {
            const uint64_t byte = w * 4 + k;
            if (byte >= n_bytes) break;
            const uint64_t rec = byte / stride;
            const uint32_t j = (uint32_t)(byte - rec * stride);
            out |= (uint32_t)ngsq_synth_qual_byte(&cfg, first + rec, cfg.read_len, j) << (8 * k);
        }
        if (w * 4 + 3 < n_bytes)
            *reinterpret_cast<uint32_t *>(qual + w * 4) = out;
        else
            for (uint32_t k = 0; w * 4 + k < n_bytes; k++) qual[w * 4 + k] = (uint8_t)(out >> (8 * k));
    }
}

// MIXED mode: one wave per record writes its seq and qual bytes
__global__ __launch_bounds__(256) void k_synth_seq_qual_var(ngsq_synth_config cfg, Cols c, uint64_t first,
                                                            uint64_t n) {
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t i = wave; i < n; i += waves) {
        const uint32_t l = ngsq_synth_len(&cfg, first + i);
        uint8_t *s = c.seq + (c.seq_off ? c.seq_off[i] : i * (uint64_t)c.seq_stride);
        uint8_t *q = c.qual + (c.qual_off ? c.qual_off[i] : i * (uint64_t)c.qual_stride);
        for (uint32_t j = lane; j < (l + 1) / 2; j += 64) s[j] = ngsq_synth_seq_byte(&cfg, first + i, l, j);
        for (uint32_t j = lane; j < l; j += 64) q[j] = ngsq_synth_qual_byte(&cfg, first + i, l, j);
    }
}

int check_cols(const ngsq_synth_config *cfg, const ngsq_batch *b) {
    if (!cfg || !b || b->struct_size != sizeof(ngsq_batch)) return NGSQ_ERR_INVALID_ARGUMENT;
    if (!b->flag || !b->mapq || !b->ref_id || !b->pos || !b->mate_ref_id || !b->tlen || !b->l_seq || !b->n_cigar ||
        !b->seq || !b->qual || !b->cigar)
        return NGSQ_ERR_INVALID_ARGUMENT;
    if (cfg->mode == NGSQ_SYNTH_FIXED) {
        if (b->seq_off || b->qual_off || b->cigar_off) return NGSQ_ERR_INVALID_ARGUMENT;
        if (b->seq_stride != (cfg->read_len + 1) / 2 || b->qual_stride != cfg->read_len || b->cigar_stride < 1)
            return NGSQ_ERR_INVALID_ARGUMENT;
    } else {
        if (cfg->mode != NGSQ_SYNTH_MIXED || cfg->min_len < 50 || cfg->max_len < cfg->min_len)
            return NGSQ_ERR_INVALID_ARGUMENT;
        if (!b->seq_off || !b->qual_off || !b->cigar_off) return NGSQ_ERR_INVALID_ARGUMENT;
    }
    return NGSQ_OK;
}

} // namespace

extern "C" {

int ngsq_synth_sizes(const ngsq_synth_config *cfg, uint64_t first, uint64_t n, uint64_t *seq_bytes,
                     uint64_t *qual_bytes, uint64_t *cigar_ops) {
    if (!cfg) return NGSQ_ERR_INVALID_ARGUMENT;
    uint64_t sb = 0, qb = 0, co = 0;
    if (cfg->mode == NGSQ_SYNTH_FIXED) {
        sb = n * ((cfg->read_len + 1) / 2);
        qb = n * cfg->read_len;
        co = n;
    } else {
        for (uint64_t i = 0; i < n; i++) {
            ngsq_synth_record r;
            ngsq_synth_record_at(cfg, first + i, &r);
            sb += (r.l_seq + 1) / 2;
            qb += r.l_seq;
            co += r.n_cigar;
        }
    }
    if (seq_bytes) *seq_bytes = sb;
    if (qual_bytes) *qual_bytes = qb;
    if (cigar_ops) *cigar_ops = co;
    return NGSQ_OK;
}

int ngsq_synth_fill_reference(const ngsq_synth_config *cfg, uint32_t ref, uint8_t *codes, uint64_t len, int n_threads) {
    if (!cfg || (!codes && len)) return NGSQ_ERR_INVALID_ARGUMENT;
    const int nt = n_threads > 0 ? n_threads : ngsq::effective_cores();
    std::vector<std::thread> pool;
    const uint64_t per = ((len + (uint64_t)nt - 1) / (uint64_t)nt + 31) & ~31ull;
    for (int t = 0; t < nt; t++)
        pool.emplace_back([=]() {
            const uint64_t lo = per * (uint64_t)t, hi = lo + per < len ? lo + per : len;
            for (uint64_t p = lo; p < hi; p++) codes[p] = (uint8_t)ngsq_synth_ref_code(cfg, ref, p);
        });
    for (auto &th : pool) th.join();
    return NGSQ_OK;
}

int ngsq_synth_fill_host(const ngsq_synth_config *cfg, uint64_t first, uint64_t n, const ngsq_batch *b) {
    int rc = check_cols(cfg, b);
    if (rc) return rc;
    Cols c = cols_of(b);
    if (cfg->mode == NGSQ_SYNTH_MIXED) {
        // offsets first (the caller provides the n+1 arrays, we fill them)
        uint64_t *so = const_cast<uint64_t *>(b->seq_off), *qo = const_cast<uint64_t *>(b->qual_off),
                 *co = const_cast<uint64_t *>(b->cigar_off);
        so[0] = qo[0] = co[0] = 0;
        for (uint64_t i = 0; i < n; i++) {
            ngsq_synth_record r;
            ngsq_synth_record_at(cfg, first + i, &r);
            so[i + 1] = so[i] + (r.l_seq + 1) / 2;
            qo[i + 1] = qo[i] + r.l_seq;
            co[i + 1] = co[i] + r.n_cigar;
        }
    }
    for (uint64_t i = 0; i < n; i++) {
        fill_fixed_fields(*cfg, c, first, i);
        const uint32_t l = c.l_seq[i];
        uint8_t *s = c.seq + (c.seq_off ? c.seq_off[i] : i * (uint64_t)c.seq_stride);
        uint8_t *q = c.qual + (c.qual_off ? c.qual_off[i] : i * (uint64_t)c.qual_stride);
        for (uint32_t j = 0; j < (l + 1) / 2; j++) s[j] = ngsq_synth_seq_byte(cfg, first + i, l, j);
        for (uint32_t j = 0; j < l; j++) q[j] = ngsq_synth_qual_byte(cfg, first + i, l, j);
    }
    return NGSQ_OK;
}

int ngsq_synth_fill_device(ngsq_ctx *ctx, const ngsq_synth_config *cfg, uint64_t first, uint64_t n,
                           const ngsq_batch *b) {
    if (!ctx) return NGSQ_ERR_INVALID_ARGUMENT;
    int rc = check_cols(cfg, b);
    if (rc) return rc;
    if (!n) return NGSQ_OK;
    if (hipSetDevice(ctx->device) != hipSuccess) return NGSQ_ERR_DEVICE;
    Cols c = cols_of(b);
    hipStream_t s = ctx->stream;
    if (cfg->mode == NGSQ_SYNTH_MIXED) {
        // offsets are computed on the host (pure function of the index) and uploaded
        std::vector<uint64_t> so(n + 1), qo(n + 1), co(n + 1);
        so[0] = qo[0] = co[0] = 0;
        for (uint64_t i = 0; i < n; i++) {
            ngsq_synth_record r;
            ngsq_synth_record_at(cfg, first + i, &r);
            so[i + 1] = so[i] + (r.l_seq + 1) / 2;
            qo[i + 1] = qo[i] + r.l_seq;
            co[i + 1] = co[i] + r.n_cigar;
        }
        if (hipMemcpy(const_cast<uint64_t *>(b->seq_off), so.data(), (n + 1) * 8, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(const_cast<uint64_t *>(b->qual_off), qo.data(), (n + 1) * 8, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(const_cast<uint64_t *>(b->cigar_off), co.data(), (n + 1) * 8, hipMemcpyHostToDevice) != hipSuccess)
            return NGSQ_ERR_DEVICE;
    }
    // GENOME mode: the kernels take the configuration by value -- with the tables in device memory
    ngsq_synth_config dcfg = *cfg;
    uint32_t *d_glen = nullptr;
    uint64_t *d_groom = nullptr;
    if (cfg->genome_n) {
        if (!cfg->genome_len || !cfg->genome_room) return NGSQ_ERR_INVALID_ARGUMENT;
        if (hipMalloc((void **)&d_glen, cfg->genome_n * 4ull) != hipSuccess || hipMalloc((void **)&d_groom, (cfg->genome_n + 1ull) * 8) != hipSuccess ||
            hipMemcpy(d_glen, cfg->genome_len, cfg->genome_n * 4ull, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(d_groom, cfg->genome_room, (cfg->genome_n + 1ull) * 8, hipMemcpyHostToDevice) != hipSuccess) {
            (void)hipFree(d_glen);
            (void)hipFree(d_groom);
            return NGSQ_ERR_DEVICE;
        }
        dcfg.genome_len = d_glen;
        dcfg.genome_room = d_groom;
    }
    cfg = &dcfg;
    const uint32_t g = (uint32_t)((n + 255) / 256);
    hipLaunchKernelGGL(k_synth_fields, dim3(g), dim3(256), 0, s, *cfg, c, first, n);
    const uint32_t big = (uint32_t)ctx->li.n_cu * 16;
    if (cfg->mode == NGSQ_SYNTH_FIXED) {
        hipLaunchKernelGGL(k_synth_seq_fixed, dim3(big), dim3(256), 0, s, *cfg, c.seq, c.seq_stride, first,
                           n * c.seq_stride);
        hipLaunchKernelGGL(k_synth_qual_fixed, dim3(big), dim3(256), 0, s, *cfg, c.qual, c.qual_stride, first,
                           n * c.qual_stride);
    } else {
        hipLaunchKernelGGL(k_synth_seq_qual_var, dim3(big), dim3(256), 0, s, *cfg, c, first, n);
    }
    const bool ok = hipGetLastError() == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
    (void)hipFree(d_glen);
    (void)hipFree(d_groom);
    return ok ? NGSQ_OK : NGSQ_ERR_DEVICE;
}

} // extern "C"
