// cand_probe.hip -- timing probe for the record-index kernel on a synthetic record stream (measurement aid for
// DESIGN.md section 9, not part of the library): k_rec_candidates against its parts (screening only, one lane walking,
// all listed lanes walking, with the piece table) and against a bare pointer chase over the same records.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Ings_amd/csrc -Iinclude tools/cand_probe.hip -o /tmp/cand_probe && /tmp/cand_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include "../ngs_amd/csrc/bam_device.hip"
using namespace ngsq;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_chase(const uint8_t *raw, uint64_t n_bytes, uint32_t n_seg, uint32_t rec, uint64_t *out, int loads) {
    const uint32_t seg = blockIdx.x * blockDim.x / 64 + threadIdx.x / 64;
    if (seg >= n_seg || (threadIdx.x & 63)) return;
    uint64_t o = ((uint64_t)seg * 65536 + rec - 1) / rec * rec, end = min((uint64_t)(seg + 1) * 65536, n_bytes);
    uint32_t acc = 0;
    while (o < end && o + 36 <= n_bytes) {
        uint4 a, b{};
        __builtin_memcpy(&a, raw + o, 16);
        if (loads > 1) __builtin_memcpy(&b, raw + o + 16, 16);
        acc += b.y;
        o += 4 + (uint64_t)a.x;
    }
    out[seg] = o + acc;
}


template <int MODE>
__global__ __launch_bounds__(64) void k_var(const uint8_t *__restrict__ raw, uint64_t n_bytes, uint64_t first, uint32_t n_seg, int32_t n_ref, uint64_t *__restrict__ outv) {
    constexpr uint32_t LIST = 8;
    __shared__ uint64_t s_list[LIST];
    const uint32_t seg = blockIdx.x, lane = threadIdx.x;
    if (seg >= n_seg) return;
    const uint64_t s0 = (uint64_t)seg * REC_SEGMENT, s1 = min(s0 + REC_SEGMENT, n_bytes);
    uint64_t pos = max(s0, first);
    uint32_t list_n = 0;
    for (int it = 0; it < 16 && pos < s1 && list_n < LIST; it++) {
        const uint64_t o = pos + lane;
        bool pass = false;
        if (o < s1 && o + 36 <= n_bytes) {
            uint4 a, b;
            ld2x16(raw + o, a, b);
            const uint32_t bs = a.x, l_read_name = a.w & 0xFFu, n_ops = b.x & 0xFFFFu, l = b.y;
            const uint64_t need = 32ull + l_read_name + 4ull * n_ops + ((uint64_t)l + 1) / 2 + l;
            const int32_t ref = (int32_t)a.y, p = (int32_t)a.z, mref = (int32_t)b.z, mpos = (int32_t)b.w;
            pass = o + 4 + (uint64_t)bs <= n_bytes && bs >= 32 && l_read_name != 0 && need <= bs && ref >= -1 && ref < n_ref &&
                   mref >= -1 && mref < n_ref && p >= -1 && mpos >= -1;
        }
        const uint64_t m = __ballot(pass);
        const uint32_t k = (uint32_t)__popcll(m), room = LIST - list_n;
        const uint32_t rank = (uint32_t)__popcll(m & ((1ull << lane) - 1));
        if (pass && rank < room) s_list[list_n + rank] = o;
        if (k > room) list_n = LIST; else list_n += k;
        pos += 64;
    }
    __syncthreads();
    if (MODE == 0) { if (lane == 0) outv[seg] = list_n; return; }
    __shared__ uint32_t s_rel[REC_PIECES * LIST], s_cnt[REC_PIECES * LIST];
    uint64_t o = 0, landing = 0; uint32_t count = 0; bool ok = false;
    if (lane < (MODE == 1 ? min(list_n, 1u) : list_n)) {
        o = s_list[lane];
        if (MODE >= 3) {
            for (uint32_t j = 0; j < REC_PIECES; j++) s_rel[j * LIST + lane] = SUB_NONE;
            ok = walk<true>(raw, n_bytes, o, s0, s1, n_ref, &landing, &count, [&](uint32_t j, uint32_t rel, uint32_t cnt) { s_rel[j * LIST + lane] = rel; s_cnt[j * LIST + lane] = cnt; });
        } else
            ok = walk<true>(raw, n_bytes, o, s0, s1, n_ref, &landing, &count, [&](uint32_t, uint32_t, uint32_t) {});
    }
    if (MODE >= 3) { __syncthreads(); if (lane == 0) landing += s_rel[5 * LIST] + s_cnt[7 * LIST]; }
    if (lane == 0) outv[seg] = landing + count + ok;
}

template <int MODE>
__global__ __launch_bounds__(64) void k_var2(const uint8_t *__restrict__ raw, uint64_t n_bytes, uint64_t first, uint32_t n_seg, int32_t n_ref, uint64_t *__restrict__ outv) {
    constexpr uint32_t LIST = 8;
    __shared__ uint64_t s_list[LIST];
    __shared__ uint32_t s_rel[REC_PIECES * LIST], s_cnt[REC_PIECES * LIST];
    const uint32_t seg = blockIdx.x, lane = threadIdx.x;
    if (seg >= n_seg) return;
    const uint64_t s0 = (uint64_t)seg * REC_SEGMENT, s1 = min(s0 + REC_SEGMENT, n_bytes);
    uint64_t pos = max(s0, first);
    uint32_t found = 0;
    uint64_t acc = 0;
    while (pos < s1 && found < REC_CANDIDATES) {
        uint32_t list_n = 0;
        for (int it = 0; it < 16 && pos < s1 && list_n < LIST; it++) {
            const uint64_t o = pos + lane;
            bool pass = false;
            if (o < s1 && o + 36 <= n_bytes) {
                uint4 a, b;
                ld2x16(raw + o, a, b);
                const uint32_t bs = a.x, l_read_name = a.w & 0xFFu, n_ops = b.x & 0xFFFFu, l = b.y;
                const uint64_t need = 32ull + l_read_name + 4ull * n_ops + ((uint64_t)l + 1) / 2 + l;
                const int32_t ref = (int32_t)a.y, p = (int32_t)a.z, mref = (int32_t)b.z, mpos = (int32_t)b.w;
                pass = o + 4 + (uint64_t)bs <= n_bytes && bs >= 32 && l_read_name != 0 && need <= bs && ref >= -1 && ref < n_ref &&
                       mref >= -1 && mref < n_ref && p >= -1 && mpos >= -1;
            }
            const uint64_t m = __ballot(pass);
            const uint32_t k = (uint32_t)__popcll(m), room = LIST - list_n;
            const uint32_t rank = (uint32_t)__popcll(m & ((1ull << lane) - 1));
            if (pass && rank < room) s_list[list_n + rank] = o;
            if (MODE >= 1 && k > room) {
                uint64_t mm = m;
                for (uint32_t q = 1; q < room; q++) mm &= mm - 1;
                pos += (uint32_t)__builtin_ctzll(mm) + 1;
                list_n = LIST;
            } else {
                list_n = min(list_n + k, LIST);
                pos += 64;
            }
        }
        if (!list_n) continue;
        __syncthreads();
        uint64_t o = 0, landing = 0; uint32_t count = 0; bool ok = false;
        if (lane < list_n) {
            o = s_list[lane];
            for (uint32_t j = 0; j < REC_PIECES; j++) s_rel[j * LIST + lane] = SUB_NONE;
            ok = walk<true>(raw, n_bytes, o, s0, s1, n_ref, &landing, &count, [&](uint32_t j, uint32_t rel, uint32_t cnt) { s_rel[j * LIST + lane] = rel; s_cnt[j * LIST + lane] = cnt; });
        }
        const uint64_t m = __ballot(ok);
        found += (uint32_t)__popcll(m);
        acc += landing + count;
        __syncthreads();
    }
    if (lane == 0) outv[seg] = acc + s_rel[5 * LIST] + s_cnt[7 * LIST];
}

int main() {
    const uint64_t n_bytes = 520ull << 20;
    const uint32_t rec = 289;
    std::vector<uint8_t> h(n_bytes + 64, 0);
    for (uint64_t o = 0; o + 36 <= n_bytes; o += rec) {
        uint32_t w[9] = {rec - 4, 0, (uint32_t)(o / rec), 20u | (60u << 8), 1u | (99u << 16), 150, 0, (uint32_t)(o / rec) + 300, 350};
        memcpy(h.data() + o, w, 36);
        for (uint32_t k = 36; k < rec && o + k < n_bytes; k++) h[o + k] = (uint8_t)(37 + (k * 7 + o) % 5);
    }
    uint8_t *raw; CK(hipMalloc(&raw, n_bytes + 64)); CK(hipMemcpy(raw, h.data(), n_bytes + 64, hipMemcpyHostToDevice));
    const uint32_t n_seg = (uint32_t)((n_bytes + REC_SEGMENT - 1) / REC_SEGMENT);
    RecCandidate *cand; CK(hipMalloc(&cand, (size_t)n_seg * REC_CANDIDATES * sizeof(RecCandidate)));
    uint64_t *out; CK(hipMalloc(&out, n_seg * 8));
    RecPieces *pieces; CK(hipMalloc(&pieces, (size_t)n_seg * REC_CANDIDATES * sizeof(RecPieces)));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](const char *name, auto fn) {
        float best = 1e9;
        for (int r = 0; r < 5; r++) { hipEventRecord(e0, 0); fn(); hipEventRecord(e1, 0); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
        printf("%-44s %.3f ms\n", name, best);
    };
    time("k_rec_candidates", [&] { launch_rec_candidates(raw, n_bytes, 0, n_seg, 2, cand, pieces, nullptr, 0); });
    time("screen only", [&] { hipLaunchKernelGGL(k_var<0>, dim3(n_seg), dim3(64), 0, 0, raw, n_bytes, 0, n_seg, 2, out); });
    time("screen + one lane walks", [&] { hipLaunchKernelGGL(k_var<1>, dim3(n_seg), dim3(64), 0, 0, raw, n_bytes, 0, n_seg, 2, out); });
    time("screen + all listed lanes walk", [&] { hipLaunchKernelGGL(k_var<2>, dim3(n_seg), dim3(64), 0, 0, raw, n_bytes, 0, n_seg, 2, out); });
    time("screen + all walk + pieces in LDS", [&] { hipLaunchKernelGGL(k_var<3>, dim3(n_seg), dim3(64), 0, 0, raw, n_bytes, 0, n_seg, 2, out); });
    time("while loop, pos += 64", [&] { hipLaunchKernelGGL(k_var2<0>, dim3(n_seg), dim3(64), 0, 0, raw, n_bytes, 0, n_seg, 2, out); });
    time("while loop, pos by the list", [&] { hipLaunchKernelGGL(k_var2<1>, dim3(n_seg), dim3(64), 0, 0, raw, n_bytes, 0, n_seg, 2, out); });
    time("chase, 1 wave per block, 1 load per hop", [&] { hipLaunchKernelGGL(k_chase, dim3(n_seg), dim3(64), 0, 0, raw, n_bytes, n_seg, rec, out, 1); });
    time("chase, 1 wave per block, 2 loads per hop", [&] { hipLaunchKernelGGL(k_chase, dim3(n_seg), dim3(64), 0, 0, raw, n_bytes, n_seg, rec, out, 2); });
    time("chase, 4 waves per block, 2 loads per hop", [&] { hipLaunchKernelGGL(k_chase, dim3((n_seg + 3) / 4), dim3(256), 0, 0, raw, n_bytes, n_seg, rec, out, 2); });
    time("k_rec_candidates (again)", [&] { launch_rec_candidates(raw, n_bytes, 0, n_seg, 2, cand, pieces, nullptr, 0); });
    std::vector<RecCandidate> hc((size_t)n_seg * REC_CANDIDATES);
    CK(hipMemcpy(hc.data(), cand, hc.size() * sizeof(RecCandidate), hipMemcpyDeviceToHost));
    uint64_t cur = 0, total = 0; int miss = 0;
    for (uint32_t s = 0; s < n_seg; s++) { const RecCandidate *c = nullptr; for (uint32_t k = 0; k < REC_CANDIDATES; k++) if (hc[s * REC_CANDIDATES + k].valid && hc[s * REC_CANDIDATES + k].start == cur) c = &hc[s * REC_CANDIDATES + k]; if (!c) { miss++; break; } total += c->count; cur = c->landing; }
    printf("chain: %llu records, misses %d (expected %llu)\n", (unsigned long long)total, miss, (unsigned long long)(n_bytes / rec));
    return 0;
}
