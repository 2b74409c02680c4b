// rows_probe.hip -- timing probe for the fixed-pitch column kernel on a synthetic record stream (measurement aid for
// DESIGN.md section 9, not part of the library): k_rec_rows against variants that stage the records' span in LDS, against
// the branch-free form it became, and against a plain copy / plain stores of the same bytes.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Ings_amd/csrc -Iinclude tools/rows_probe.hip -o /tmp/rows_probe && /tmp/rows_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include "../ngs_amd/csrc/bam_device.hip"
using namespace ngsq;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// V1: a block stages the contiguous source span of its records in LDS (aligned, coalesced), then writes both columns from there
constexpr uint32_t RB = 64;          // records per block
constexpr uint32_t SPAN_MAX = 40960; // bytes of LDS for the span
__global__ __launch_bounds__(256) void k_rows_lds(const uint8_t *__restrict__ raw, const uint64_t *__restrict__ rec_off, const uint64_t *__restrict__ seq_src,
                                                  const uint32_t *__restrict__ l_seq, uint64_t n, uint32_t *__restrict__ dseq, uint32_t *__restrict__ dqual,
                                                  uint32_t ps, uint32_t pq) {
    __shared__ __attribute__((aligned(16))) uint8_t s_span[SPAN_MAX + 32];
    __shared__ uint32_t s_src[RB + 1], s_len[RB + 1];
    const uint64_t i0 = (uint64_t)blockIdx.x * RB;
    const uint32_t nr = (uint32_t)min((uint64_t)RB, n - i0);
    const uint64_t lo = rec_off[i0] & ~15ull;
    const uint64_t hi = i0 + nr < n ? rec_off[i0 + nr] : seq_src[i0 + nr - 1] + (l_seq[i0 + nr - 1] + 1) / 2 + l_seq[i0 + nr - 1];
    const uint32_t span = (uint32_t)(hi - lo);
    if (threadIdx.x < nr) {
        s_src[threadIdx.x] = (uint32_t)(seq_src[i0 + threadIdx.x] - lo);
        s_len[threadIdx.x] = l_seq[i0 + threadIdx.x];
    }
    if (span > SPAN_MAX) return; // (probe: the library would take the direct path here)
    for (uint32_t o = threadIdx.x * 16; o < span; o += 256 * 16) *reinterpret_cast<uint4 *>(s_span + o) = *reinterpret_cast<const uint4 *>(raw + lo + o);
    __syncthreads();
    const uint32_t *sw = reinterpret_cast<const uint32_t *>(s_span);
    auto column = [&](uint32_t *dst, uint32_t pitch, bool qual) {
        const uint64_t b0 = i0 * pitch;           // first byte of the block's rows in the column
        const uint64_t d0 = b0 / 4, d1 = (b0 + (uint64_t)nr * pitch + 3) / 4; // dwords touched (shared edge dwords: both blocks write... see below)
        for (uint64_t d = d0 + threadIdx.x; d < d1; d += 256) {
            uint32_t v = 0;
#pragma unroll
            for (uint32_t b = 0; b < 4; b++) {
                const uint64_t a = d * 4 + b;
                uint32_t byte = qual ? 0xFFu : 0u;
                if (a >= b0 && a < b0 + (uint64_t)nr * pitch) {
                    const uint32_t rel = (uint32_t)(a - b0), r = rel / pitch, k = rel - r * pitch;
                    const uint32_t l = s_len[r], sb = (l + 1) / 2, len = qual ? l : sb;
                    if (k < len) byte = s_span[s_src[r] + (qual ? sb : 0u) + k];
                }
                v |= byte << (8 * b);
            }
            (void)sw;
            dst[d] = v; // probe only: edge dwords shared with the neighbouring block are written by both (race; timing only)
        }
    };
    column(dseq, ps, false);
    column(dqual, pq, true);
}

// V2: as V1 but dword-wise from LDS: two aligned LDS reads + alignbyte where the four bytes belong to one record
__global__ __launch_bounds__(256) void k_rows_lds2(const uint8_t *__restrict__ raw, const uint64_t *__restrict__ rec_off, const uint64_t *__restrict__ seq_src,
                                                   const uint32_t *__restrict__ l_seq, uint64_t n, uint32_t *__restrict__ dseq, uint32_t *__restrict__ dqual,
                                                   uint32_t ps, uint32_t pq) {
    __shared__ __attribute__((aligned(16))) uint8_t s_span[SPAN_MAX + 32];
    __shared__ uint32_t s_src[RB + 1], s_len[RB + 1];
    const uint64_t i0 = (uint64_t)blockIdx.x * RB;
    const uint32_t nr = (uint32_t)min((uint64_t)RB, n - i0);
    const uint64_t lo = rec_off[i0] & ~15ull;
    const uint64_t hi = i0 + nr < n ? rec_off[i0 + nr] : seq_src[i0 + nr - 1] + (l_seq[i0 + nr - 1] + 1) / 2 + l_seq[i0 + nr - 1];
    const uint32_t span = (uint32_t)(hi - lo);
    if (threadIdx.x < nr) {
        s_src[threadIdx.x] = (uint32_t)(seq_src[i0 + threadIdx.x] - lo);
        s_len[threadIdx.x] = l_seq[i0 + threadIdx.x];
    }
    if (span > SPAN_MAX) return;
    for (uint32_t o = threadIdx.x * 16; o < span; o += 256 * 16) *reinterpret_cast<uint4 *>(s_span + o) = *reinterpret_cast<const uint4 *>(raw + lo + o);
    __syncthreads();
    const uint32_t *sw = reinterpret_cast<const uint32_t *>(s_span);
    auto column = [&](uint32_t *dst, uint32_t pitch, bool qual) {
        const uint64_t b0 = i0 * pitch;
        const uint64_t d0 = b0 / 4, d1 = (b0 + (uint64_t)nr * pitch + 3) / 4;
        for (uint64_t d = d0 + threadIdx.x; d < d1; d += 256) {
            const uint64_t a = d * 4;
            uint32_t v;
            const int64_t rel64 = (int64_t)(a - b0);
            const uint32_t rel = (uint32_t)rel64, r = rel / pitch, k = rel - r * pitch;
            bool fast = rel64 >= 0 && r < nr;
            uint32_t l = 0, sb = 0, len = 0, so = 0;
            if (fast) { l = s_len[r]; sb = (l + 1) / 2; len = qual ? l : sb; so = s_src[r] + (qual ? sb : 0u) + k; fast = k + 4 <= len; }
            if (fast) {
                const uint32_t w0 = sw[so >> 2], w1 = sw[(so >> 2) + 1];
                v = __builtin_amdgcn_alignbyte(w1, w0, so & 3u);
            } else {
                v = 0;
#pragma unroll
                for (uint32_t b = 0; b < 4; b++) {
                    const uint64_t ab = a + b;
                    uint32_t byte = qual ? 0xFFu : 0u;
                    if (ab >= b0 && ab < b0 + (uint64_t)nr * pitch) {
                        const uint32_t rl = (uint32_t)(ab - b0), rr = rl / pitch, kk = rl - rr * pitch;
                        const uint32_t l2 = s_len[rr], sb2 = (l2 + 1) / 2;
                        if (kk < (qual ? l2 : sb2)) byte = s_span[s_src[rr] + (qual ? sb2 : 0u) + kk];
                    }
                    v |= byte << (8 * b);
                }
            }
            dst[d] = v;
        }
    };
    column(dseq, ps, false);
    column(dqual, pq, true);
}


__global__ __launch_bounds__(256) void k_copy(const uint4 *__restrict__ src, uint4 *__restrict__ dst, uint64_t n_src, uint64_t n_dst) {
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n_src; i += (uint64_t)gridDim.x * 256) {
        const uint4 v = src[i];
        if (i < n_dst) dst[i] = v; else { acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) dst[0] = acc;
}
__global__ __launch_bounds__(256) void k_store_only(uint32_t *__restrict__ dst, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) dst[i] = (uint32_t)i;
}
__global__ __launch_bounds__(256) void k_store16(uint4 *__restrict__ dst, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) dst[i] = make_uint4((uint32_t)i, 1, 2, 3);
}

// V3: branch-free: the dword's own row and the next row are both fetched, the bytes are chosen with masks (pitch >= 4)
__device__ __forceinline__ uint32_t bmask(uint32_t nbytes) { return nbytes >= 4 ? 0xFFFFFFFFu : (1u << (8 * nbytes)) - 1u; }
template <bool QUAL>
__global__ __launch_bounds__(256) void k_rows_mask(const uint8_t *__restrict__ raw, const uint64_t *__restrict__ seq_src,
                                                   const uint32_t *__restrict__ l_seq, uint64_t n, uint32_t *__restrict__ dst,
                                                   uint32_t pitch, uint64_t n_dwords) {
    constexpr uint32_t FILLW = QUAL ? 0xFFFFFFFFu : 0u;
    constexpr int U = 4;
    for (uint64_t d0 = (uint64_t)blockIdx.x * (256 * U) + threadIdx.x; d0 < n_dwords; d0 += (uint64_t)gridDim.x * (256 * U)) {
        uint32_t k[U], lenA[U], lenB[U];
        const uint8_t *pa[U], *pb[U];
        bool in[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint64_t d = d0 + (uint64_t)u * 256;
            const uint32_t a = (uint32_t)d * 4u;
            const uint32_t r = a / pitch;
            k[u] = a - r * pitch;
            in[u] = d < n_dwords && r < n;
            const uint32_t r0 = in[u] ? r : 0u, r1 = in[u] && r + 1 < n ? r + 1 : r0;
            const uint32_t l0 = l_seq[r0], l1 = l_seq[r1];
            const uint32_t sb0 = (l0 + 1) / 2, sb1 = (l1 + 1) / 2;
            lenA[u] = QUAL ? l0 : sb0;
            lenB[u] = in[u] && r + 1 < n ? (QUAL ? l1 : sb1) : 0u;
            pa[u] = raw + seq_src[r0] + (QUAL ? sb0 : 0u) + k[u];
            pb[u] = raw + seq_src[r1] + (QUAL ? sb1 : 0u);
        }
        uint32_t A[U], B[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            A[u] = ld32(pa[u]);
            B[u] = ld32(pb[u]);
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint64_t d = d0 + (uint64_t)u * 256;
            if (d >= n_dwords) break;
            const uint32_t nA = lenA[u] > k[u] ? lenA[u] - k[u] : 0u;       // bytes of this dword inside row r's data
            const uint32_t t = pitch - k[u];                                  // bytes of this dword inside row r (>= 1)
            const uint32_t mA = in[u] ? bmask(nA < t ? nA : t) : 0u;
            const uint32_t nB = t < 4 ? (lenB[u] < 4 - t ? lenB[u] : 4 - t) : 0u;
            const uint32_t mB = t < 4 ? bmask(nB) << (8 * t) : 0u;
            const uint32_t Bs = t < 4 ? B[u] << (8 * t) : 0u;
            dst[d] = (A[u] & mA) | (Bs & mB) | (FILLW & ~(mA | mB));
        }
    }
}

int main() {
    const uint64_t n_bytes = 520ull << 20;
    const uint32_t rec = 289, l = 150;
    const uint64_t n = n_bytes / rec;
    std::vector<uint8_t> h(n_bytes + 64, 0);
    std::vector<uint64_t> off(n + 1), src(2 * n);
    std::vector<uint32_t> ls(n), nc(n, 1);
    for (uint64_t i = 0; i < n; i++) {
        const uint64_t o = i * rec;
        uint32_t w[9] = {rec - 4, 0, (uint32_t)i, 20u | (60u << 8), 1u | (99u << 16), l, 0, (uint32_t)i + 300, 350};
        memcpy(h.data() + o, w, 36);
        for (uint32_t k = 36; k < rec; k++) h[o + k] = (uint8_t)(37 + (k * 7 + o) % 5);
        off[i] = o; src[i] = o + 36 + 20; src[n + i] = o + 36 + 20 + 4; ls[i] = l;
    }
    off[n] = n * rec;
    uint8_t *raw; CK(hipMalloc(&raw, n_bytes + 64)); CK(hipMemcpy(raw, h.data(), n_bytes + 64, hipMemcpyHostToDevice));
    uint64_t *d_off, *d_src; uint32_t *d_l, *dseq, *dqual;
    CK(hipMalloc(&d_off, (n + 1) * 8)); CK(hipMemcpy(d_off, off.data(), (n + 1) * 8, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_src, 2 * n * 8)); CK(hipMemcpy(d_src, src.data(), 2 * n * 8, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_l, n * 4)); CK(hipMemcpy(d_l, ls.data(), n * 4, hipMemcpyHostToDevice));
    const uint32_t ps = 75, pq = 150;
    CK(hipMalloc(&dseq, n * ps + 256)); CK(hipMalloc(&dqual, n * pq + 256));
    uint32_t *rseq, *rqual; CK(hipMalloc(&rseq, n * ps + 256)); CK(hipMalloc(&rqual, n * pq + 256));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](const char *name, auto fn) {
        float best = 1e9;
        for (int r = 0; r < 5; r++) { hipEventRecord(e0, 0); fn(); hipEventRecord(e1, 0); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
        printf("%-44s %.3f ms\n", name, best);
    };
    const uint64_t ds = (n * ps + 3) / 4, dq = (n * pq + 3) / 4;
    time("k_rec_rows seq + qual (library)", [&] {
        hipLaunchKernelGGL(k_rec_rows<false>, dim3((uint32_t)std::min<uint64_t>((ds + 1023) / 1024, 1u << 16)), dim3(256), 0, 0, raw, d_src + n, d_l, n, rseq, ps, ds);
        hipLaunchKernelGGL(k_rec_rows<true>, dim3((uint32_t)std::min<uint64_t>((dq + 1023) / 1024, 1u << 16)), dim3(256), 0, 0, raw, d_src + n, d_l, n, rqual, pq, dq);
    });
    time("LDS-staged span, byte gathers", [&] { hipLaunchKernelGGL(k_rows_lds, dim3((uint32_t)((n + RB - 1) / RB)), dim3(256), 0, 0, raw, d_off, d_src + n, d_l, n, dseq, dqual, ps, pq); });
    time("LDS-staged span, dword alignbyte", [&] { hipLaunchKernelGGL(k_rows_lds2, dim3((uint32_t)((n + RB - 1) / RB)), dim3(256), 0, 0, raw, d_off, d_src + n, d_l, n, dseq, dqual, ps, pq); });
    time("branch-free masks, seq + qual", [&] {
        hipLaunchKernelGGL(k_rows_mask<false>, dim3((uint32_t)std::min<uint64_t>((ds + 1023) / 1024, 1u << 16)), dim3(256), 0, 0, raw, d_src + n, d_l, n, dseq, ps, ds);
        hipLaunchKernelGGL(k_rows_mask<true>, dim3((uint32_t)std::min<uint64_t>((dq + 1023) / 1024, 1u << 16)), dim3(256), 0, 0, raw, d_src + n, d_l, n, dqual, pq, dq);
    });
    {
        std::vector<uint8_t> a(n * pq), b(n * pq);
        CK(hipMemcpy(a.data(), rqual, n * pq, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), dqual, n * pq, hipMemcpyDeviceToHost));
        size_t bad = 0; for (size_t i = 0; i < a.size(); i++) bad += a[i] != b[i];
        CK(hipMemcpy(a.data(), rseq, n * ps, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), dseq, n * ps, hipMemcpyDeviceToHost));
        size_t bad2 = 0; for (size_t i = 0; i < n * ps; i++) bad2 += a[i] != b[i];
        printf("masks vs library: qual bytes differing %zu, seq bytes differing %zu\n", bad, bad2);
    }
    time("plain copy 520 MB read, 425 MB written (16 B)", [&] { hipLaunchKernelGGL(k_copy, dim3(8192), dim3(256), 0, 0, (const uint4 *)raw, (uint4 *)dqual, n_bytes / 16, (n * pq) / 16); });
    time("store only 425 MB, dword per lane", [&] { hipLaunchKernelGGL(k_store_only, dim3(16384), dim3(256), 0, 0, dqual, (n * pq) / 4); hipLaunchKernelGGL(k_store_only, dim3(16384), dim3(256), 0, 0, dseq, (n * ps) / 4); });
    time("store only 425 MB, 16 B per lane", [&] { hipLaunchKernelGGL(k_store16, dim3(8192), dim3(256), 0, 0, (uint4 *)dqual, (n * pq) / 16); hipLaunchKernelGGL(k_store16, dim3(8192), dim3(256), 0, 0, (uint4 *)dseq, (n * ps) / 16); });
    std::vector<uint8_t> a(n * pq), b(n * pq);
    CK(hipMemcpy(a.data(), rqual, n * pq, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), dqual, n * pq, hipMemcpyDeviceToHost));
    size_t bad = 0; for (size_t i = 0; i < a.size(); i++) bad += a[i] != b[i];
    CK(hipMemcpy(a.data(), rseq, n * ps, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), dseq, n * ps, hipMemcpyDeviceToHost));
    size_t bad2 = 0; for (size_t i = 0; i < n * ps; i++) bad2 += a[i] != b[i];
    printf("qual bytes differing %zu, seq bytes differing %zu\n", bad, bad2);
    return 0;
}
