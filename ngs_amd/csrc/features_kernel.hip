// features_kernel.hip -- Genomic Features facet (gfx950).
// reference: src/qc/record_based/features.rs:115-242 (process), :270-355 (interval stores)
//
// The reference keeps two rust_lapper stores per sequence (UTR/CDS features; gene/exon features)
// and, per record, walks `find(start, end + 1)`.  All it takes from the walk is, per feature NAME,
// how many overlapping intervals there are (capped by the few flags it sets).  For half-open
// intervals [s, e) with s <= e and a query [qs, qe) with qs < qe
//     #overlapping = #{ s < qe } - #{ e <= qs }
// (an interval fails `s < qe && e > qs` either by s >= qe or by e <= qs, never both), so two binary
// searches per name on the separately sorted starts and stops give the EXACT count -- no interval
// tree, no max-length scan: 10 searches per record over L2-resident lists.
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace ngsq {

namespace {

__device__ __forceinline__ uint32_t lower_bound(const uint32_t *a, uint32_t lo, uint32_t hi, uint32_t v) {
    // first index in [lo, hi) with a[i] >= v
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (a[mid] < v) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}
__device__ __forceinline__ uint32_t upper_bound(const uint32_t *a, uint32_t lo, uint32_t hi, uint32_t v) {
    // first index in [lo, hi) with a[i] > v
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (a[mid] <= v) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

// intervals of name id k on sequence r overlapping [qs, qe)
__device__ __forceinline__ uint32_t count_overlaps(const FeatureTables &ft, uint32_t k, uint32_t r, uint32_t qs, uint32_t qe) {
    const uint32_t b = ft.idx[k * ft.n_refs + r], e = ft.idx[k * ft.n_refs + r + 1];
    if (b == e) return 0;
    const uint32_t started = lower_bound(ft.starts, b, e, qe) - b; // s < qe
    const uint32_t ended = upper_bound(ft.stops, b, e, qs) - b;    // e <= qs
    return started - ended;
}

__device__ __forceinline__ void tally(unsigned long long *counter, bool pred) {
    const uint64_t m = __ballot(pred);
    if (m && (threadIdx.x & 63) == 0) atomicAdd(counter, (unsigned long long)__popcll(m));
}

} // namespace

__global__ __launch_bounds__(256) void k_features(DeviceState st, DeviceBatch b, FeatureTables ft) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t n_round = (b.n + 63) & ~63ull; // whole waves take part in the ballots
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_round; i += stride) {
        const bool live = i < b.n;
        bool ign_flags = false, ign_nonprimary = false, err_ref = false, err_pos = false, processed = false;
        bool utr5 = false, utr3 = false, cds = false, intergenic = false, exonic = false, intronic = false;
        if (live) {
            const uint32_t flag = b.flag[i];
            const int32_t ref = b.ref_id[i], pos = b.pos[i];
            if (flag & 0x4u) { // features.rs:127-130
                ign_flags = true;
            } else if (ref < 0 || (uint32_t)ref >= ft.n_refs) { // :132-155
                err_ref = true;
            } else if (!ft.primary[ref]) { // :157-165
                ign_nonprimary = true;
            } else if (pos < 0) { // :171-174
                err_pos = true;
            } else {
                // :176-178  start = alignment_start (1-based), end = start + cigar.alignment_span()
                const uint32_t n_ops = b.n_cigar[i];
                const uint64_t c0 = b.cigar_off ? b.cigar_off[i] : i * b.cigar_stride;
                uint32_t span = 0;
                for (uint32_t k = 0; k < n_ops; k++) {
                    const uint32_t op = b.cigar[c0 + k], code = op & 15u;
                    // M, D, N, =, X consume the reference
                    if (code == 0 || code == 2 || code == 3 || code == 7 || code == 8) span += op >> 4;
                }
                const uint32_t qs = (uint32_t)pos + 1u, qe = qs + span + 1u; // find(start, end + 1)
                // :186-214  UTR / CDS store: the if / else-if chain over the overlapping intervals only
                // depends on how many there are of each name (roles may share a name)
                bool c5 = false, c3 = false, cc = false;
                const uint32_t n5 = ft.role_name[NGSQ_ROLE_FIVE_PRIME_UTR], n3 = ft.role_name[NGSQ_ROLE_THREE_PRIME_UTR],
                               nc = ft.role_name[NGSQ_ROLE_CODING_SEQUENCE];
#pragma unroll
                for (uint32_t role = 0; role < 3; role++) {
                    const uint32_t name = ft.role_name[role];
                    bool seen = false; // count each distinct name once
                    for (uint32_t q = 0; q < role; q++) seen |= ft.role_name[q] == name;
                    if (seen) continue;
                    uint32_t cnt = min(count_overlaps(ft, name, (uint32_t)ref, qs, qe), 3u);
                    for (; cnt; cnt--) {
                        if (!c5 && name == n5) c5 = true;
                        else if (!c3 && name == n3) c3 = true;
                        else if (!cc && name == nc) cc = true;
                    }
                }
                utr5 = c5, utr3 = c3, cds = cc;
                // :216-238  gene / exon store.  A name that is also a UTR/CDS name never reaches this
                // store (:322-333), and `name == gene` is tested before `name == exon`.
                const uint32_t ne = ft.role_name[NGSQ_ROLE_EXON], ng = ft.role_name[NGSQ_ROLE_GENE];
                const bool gene_in_store = ng != n5 && ng != n3 && ng != nc;
                const bool exon_in_store = ne != n5 && ne != n3 && ne != nc && ne != ng;
                const bool has_gene = gene_in_store && count_overlaps(ft, ng, (uint32_t)ref, qs, qe) > 0;
                const bool has_exon = exon_in_store && count_overlaps(ft, ne, (uint32_t)ref, qs, qe) > 0;
                if (has_gene) {
                    if (has_exon) exonic = true;
                    else intronic = true;
                } else {
                    intergenic = true;
                }
                processed = true; // :240
            }
        }
        unsigned long long *c = st.counters + C_FEAT;
        tally(c + F_UTR5, utr5);
        tally(c + F_UTR3, utr3);
        tally(c + F_CDS, cds);
        tally(c + F_INTERGENIC, intergenic);
        tally(c + F_EXONIC, exonic);
        tally(c + F_INTRONIC, intronic);
        tally(c + F_PROCESSED, processed);
        tally(c + F_IGN_FLAGS, ign_flags);
        tally(c + F_IGN_NONPRIMARY, ign_nonprimary);
        tally(st.counters + C_FEAT_ERR_REF, err_ref);
        tally(st.counters + C_FEAT_ERR_POS, err_pos);
    }
}

hipError_t launch_features(const LaunchInfo &li, const DeviceState &st, const DeviceBatch &b, const FeatureTables &ft,
                           hipStream_t s) {
    if (!b.n) return hipSuccess;
    uint64_t g = (b.n + 255) / 256;
    if (g > (uint64_t)li.n_cu * 8) g = (uint64_t)li.n_cu * 8;
    hipLaunchKernelGGL(k_features, dim3((uint32_t)g), dim3(256), 0, s, st, b, ft);
    return hipGetLastError();
}

} // namespace ngsq
