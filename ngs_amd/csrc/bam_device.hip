// bam_device.hip -- BAM record parse on the device (second half):
// record boundaries in the inflated byte stream, then the structure-of-arrays columns of
// include/ngsq.h.  Mirrors, record for record, what csrc/bam_reader.cpp stages on the host
// (which stands in for noodles-bam 0.28's record decoder, src/qc/command.rs:305).
//
// Record boundaries are a linked list (each record's block_size points at the next one), so the
// stream is cut into 64 KiB segments that are walked in parallel:
//   1. k_rec_candidates  per segment, the offsets >= its start from which a chain of plausible
//                        records reaches the segment's end (the first few), each with its
//                        landing offset in a later segment and its record count;
//   2. host              follows entry -> landing through the candidate table (one table
//                        lookup per segment).  The true entry of a segment is always a record
//                        start, so a false candidate can never be selected; if the true entry is
//                        missing from the table (crowded out), k_walk_one walks that segment alone;
//   3. k_rec_offsets     one lane per segment writes the offsets of its records and validates them
//                        with the host reader's rules.
#include <hip/hip_runtime.h>

#include <hipcub/hipcub.hpp>

#include "ingest_kernels.h"

namespace ngsq {

namespace {

__device__ __forceinline__ uint32_t ld32(const uint8_t *p) {
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}
__device__ __forceinline__ uint32_t ld16(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }

// the host reader's validity rule (bam_reader.cpp): block_size >= 32, l_read_name != 0 and the
// variable-length fields fit the block
__device__ __forceinline__ bool record_valid(const uint8_t *r /* at block_size */, uint32_t bs) {
    if (bs < 32) return false;
    const uint32_t l_read_name = r[12], n_ops = ld16(r + 16), l = ld32(r + 20);
    const uint64_t need = 32ull + l_read_name + 4ull * n_ops + ((uint64_t)l + 1) / 2 + l;
    return l_read_name != 0 && need <= bs;
}

// stricter test used only to FIND chains quickly (never to reject a record)
__device__ __forceinline__ bool record_plausible(const uint8_t *r, uint32_t bs, int32_t n_ref) {
    if (!record_valid(r, bs)) return false;
    const int32_t ref = (int32_t)ld32(r + 4), pos = (int32_t)ld32(r + 8);
    const int32_t mref = (int32_t)ld32(r + 24), mpos = (int32_t)ld32(r + 28);
    return ref >= -1 && ref < n_ref && mref >= -1 && mref < n_ref && pos >= -1 && mpos >= -1;
}

// Walk the chain from `o` until it reaches `end` (segment end) or the record at the cursor is not
// completely inside [0, n_bytes).  plausible: apply the strict test.  Returns false on an invalid
// record.  *landing = cursor at the stop, *count = records passed.
__device__ bool walk(const uint8_t *raw, uint64_t n_bytes, uint64_t o, uint64_t end, int32_t n_ref, bool strict,
                     uint64_t *landing, uint32_t *count) {
    uint32_t n = 0;
    while (o < end) {
        if (o + 4 > n_bytes) break; // block_size itself is cut
        const uint32_t bs = ld32(raw + o);
        if (o + 4 + (uint64_t)bs > n_bytes) {
            // incomplete record at the end of the buffer: only its fixed part can be tested
            if (bs < 32) return false;
            break;
        }
        if (strict ? !record_plausible(raw + o, bs, n_ref) : !record_valid(raw + o, bs)) return false;
        o += 4 + (uint64_t)bs;
        n += 1;
    }
    *landing = o;
    *count = n;
    return true;
}

} // namespace

// ---- 1. candidates ----------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_rec_candidates(const uint8_t *__restrict__ raw, uint64_t n_bytes, uint64_t first,
                                                       uint32_t n_seg, int32_t n_ref, RecCandidate *__restrict__ cand) {
    const uint32_t seg = blockIdx.x, lane = threadIdx.x;
    if (seg >= n_seg) return;
    const uint64_t s0 = (uint64_t)seg * REC_SEGMENT, s1 = min(s0 + REC_SEGMENT, n_bytes);
    RecCandidate *out = cand + (uint64_t)seg * REC_CANDIDATES;
    uint32_t found = 0;
    // offsets before `first` (the BAM header) are never record starts
    for (uint64_t b = max(s0, first); b < s1 && found < REC_CANDIDATES; b += 64) {
        const uint64_t o = b + lane;
        uint64_t landing = 0;
        uint32_t count = 0;
        bool ok = false;
        if (o < s1 && o + 4 <= n_bytes) {
            const uint32_t bs = ld32(raw + o);
            // cheap screen before the walk: a complete, plausible record.  (Bytes inside a record read as a
            // huge block_size look like "the record cut by the end of the buffer": never a candidate.  The
            // one true cut record of a chunk is then found by k_walk_one.)
            if (o + 4 + (uint64_t)bs <= n_bytes && record_plausible(raw + o, bs, n_ref))
                ok = walk(raw, n_bytes, o, s1, n_ref, true, &landing, &count);
        }
        uint64_t m = __ballot(ok);
        while (m && found < REC_CANDIDATES) {
            const uint32_t l = (uint32_t)__builtin_ctzll(m);
            m &= m - 1;
            if (lane == l) out[found] = RecCandidate{o, landing, count, 1u};
            found += 1;
        }
    }
    for (uint32_t k = found + lane; k < REC_CANDIDATES; k += 64) out[k] = RecCandidate{0, 0, 0, 0u};
}

// ---- 2b. the rare segment whose entry is not in the table ---------------------------------------
__global__ void k_walk_one(const uint8_t *__restrict__ raw, uint64_t n_bytes, uint64_t start, uint64_t end,
                           RecCandidate *__restrict__ out) {
    if (threadIdx.x || blockIdx.x) return;
    uint64_t landing = start;
    uint32_t count = 0;
    const bool ok = walk(raw, n_bytes, start, end, 0, false, &landing, &count);
    *out = RecCandidate{start, landing, count, ok ? 1u : 0u};
}

// ---- 3. offsets -------------------------------------------------------------------------------
// seg_entry[s] = offset of the first record that starts in segment s (or >= its end: none);
// seg_base[s] = index of that record.  bad[0] = smallest index of an invalid record (or ~0).
__global__ __launch_bounds__(256) void k_rec_offsets(const uint8_t *__restrict__ raw, uint64_t n_bytes, uint32_t n_seg,
                                                     const uint64_t *__restrict__ seg_entry,
                                                     const uint64_t *__restrict__ seg_base, uint64_t *__restrict__ rec_off,
                                                     unsigned long long *__restrict__ bad) {
    const uint32_t seg = blockIdx.x * blockDim.x + threadIdx.x;
    if (seg >= n_seg) return;
    const uint64_t s1 = min(((uint64_t)seg + 1) * REC_SEGMENT, n_bytes);
    uint64_t o = seg_entry[seg], i = seg_base[seg];
    while (o < s1) {
        if (o + 4 > n_bytes) break;
        const uint32_t bs = ld32(raw + o);
        if (o + 4 + (uint64_t)bs > n_bytes) {
            if (bs < 32) atomicMin(bad, (unsigned long long)i);
            break;
        }
        if (!record_valid(raw + o, bs)) {
            atomicMin(bad, (unsigned long long)i);
            break;
        }
        rec_off[i++] = o;
        o += 4 + (uint64_t)bs;
    }
}

// ---- 4. fixed-width columns + the numbers the layout decision needs ----------------------------
// stats: [0] max l_seq, [1] max n_cigar, [2] sum l_seq (u64 in [2..3])
__global__ __launch_bounds__(256) void k_rec_fixed(const uint8_t *__restrict__ raw, const uint64_t *__restrict__ rec_off,
                                                   uint64_t n, RecColumns c, unsigned long long *__restrict__ stats) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t l = 0, n_ops = 0;
    if (i < n) {
        const uint8_t *r = raw + rec_off[i] + 4;
        n_ops = ld16(r + 12);
        l = ld32(r + 16);
        c.ref_id[i] = (int32_t)ld32(r);
        c.pos[i] = (int32_t)ld32(r + 4);
        c.mapq[i] = r[9];
        c.n_cigar[i] = (uint16_t)n_ops;
        c.flag[i] = (uint16_t)ld16(r + 14);
        c.l_seq[i] = l;
        c.mate_ref_id[i] = (int32_t)ld32(r + 20);
        c.tlen[i] = (int32_t)ld32(r + 28);
    }
    // block reduce (wave shuffles, then one atomic per wave)
    uint32_t ml = l, mo = n_ops;
    unsigned long long sl = l;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        ml = max(ml, (uint32_t)__shfl_xor((int)ml, o, 64));
        mo = max(mo, (uint32_t)__shfl_xor((int)mo, o, 64));
        sl += __shfl_xor(sl, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMax(&stats[0], (unsigned long long)ml);
        atomicMax(&stats[1], (unsigned long long)mo);
        atomicAdd(&stats[2], sl);
    }
}

// ---- 5. per-record lengths for the offsets layout ----------------------------------------------
// Absent qualities (l_seq bytes of 0xFF, SAM/BAM spec 4.2.3; noodles yields no scores) take no bytes.
__global__ __launch_bounds__(256) void k_rec_lengths(const uint8_t *__restrict__ raw, const uint64_t *__restrict__ rec_off,
                                                     uint64_t n, uint64_t *__restrict__ seq_len,
                                                     uint64_t *__restrict__ qual_len, uint64_t *__restrict__ cig_len) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t *r = raw + rec_off[i] + 4;
    const uint32_t l_read_name = r[8], n_ops = ld16(r + 12), l = ld32(r + 16);
    const uint8_t *ql = r + 32 + l_read_name + 4ull * n_ops + (l + 1) / 2;
    bool miss = l > 0;
    for (uint32_t k = 0; k < l && miss; k++) miss = ql[k] == 0xFF;
    seq_len[i] = (l + 1) / 2;
    qual_len[i] = miss ? 0 : l;
    cig_len[i] = n_ops;
}

// ---- 6. variable-width columns -----------------------------------------------------------------
// 16 lanes per record.  Fixed-pitch rows are padded (zero nibbles for SEQ, 0xFF for QUAL) exactly as
// bam_reader.cpp pads them; the offsets layout copies l_seq / (l_seq+1)/2 / n_ops units.
__global__ __launch_bounds__(256) void k_rec_var(const uint8_t *__restrict__ raw, const uint64_t *__restrict__ rec_off,
                                                 uint64_t n, RecColumns c) {
    const uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const uint32_t t = threadIdx.x & 15u;
    if (i >= n) return;
    const uint8_t *r = raw + rec_off[i] + 4;
    const uint32_t l_read_name = r[8], n_ops = ld16(r + 12), l = ld32(r + 16);
    const uint8_t *cg = r + 32 + l_read_name;
    const uint8_t *sq = cg + 4ull * n_ops;
    const uint8_t *ql = sq + (l + 1) / 2;
    const uint32_t sb = (l + 1) / 2;
    if (c.cigar_off) {
        uint32_t *dst = c.cigar + c.cigar_off[i];
        for (uint32_t k = t; k < n_ops; k += 16) dst[k] = ld32(cg + 4 * k);
    } else if (t == 0) {
        c.cigar[i] = n_ops ? ld32(cg) : 0u;
    }
    if (c.seq_off) {
        uint8_t *sd = c.seq + c.seq_off[i];
        for (uint32_t k = t; k < sb; k += 16) sd[k] = sq[k];
        const uint64_t q0 = c.qual_off[i], q1 = c.qual_off[i + 1];
        if (q1 > q0) {
            uint8_t *qd = c.qual + q0;
            for (uint32_t k = t; k < l; k += 16) qd[k] = ql[k];
        }
    } else {
        uint8_t *sd = c.seq + (uint64_t)c.seq_pitch * i, *qd = c.qual + (uint64_t)c.qual_pitch * i;
        for (uint32_t k = t; k < c.seq_pitch; k += 16) sd[k] = k < sb ? sq[k] : (uint8_t)0;
        for (uint32_t k = t; k < c.qual_pitch; k += 16) qd[k] = k < l ? ql[k] : (uint8_t)0xFF;
    }
}

// slack behind the packed columns that the facet kernels' vector loads may touch
__global__ void k_fill_slack(uint8_t *seq_end, uint8_t *qual_end) {
    const uint32_t t = threadIdx.x;
    if (t < 64) {
        seq_end[t] = 0;
        qual_end[t] = 0xFF;
    }
}

__global__ void k_count_below_u64(const uint64_t *__restrict__ a, uint64_t n, uint64_t value, unsigned long long *out) {
    if (threadIdx.x || blockIdx.x) return;
    uint64_t lo = 0, hi = n;
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (a[mid] < value) lo = mid + 1;
        else hi = mid;
    }
    *out = lo;
}

// ---- launchers ----------------------------------------------------------------------------------
hipError_t launch_rec_candidates(const uint8_t *raw, uint64_t n_bytes, uint64_t first, uint32_t n_seg, int32_t n_ref,
                                 RecCandidate *cand, hipStream_t s) {
    if (!n_seg) return hipSuccess;
    hipLaunchKernelGGL(k_rec_candidates, dim3(n_seg), dim3(64), 0, s, raw, n_bytes, first, n_seg, n_ref, cand);
    return hipGetLastError();
}
hipError_t launch_walk_one(const uint8_t *raw, uint64_t n_bytes, uint64_t start, uint64_t end, RecCandidate *out,
                           hipStream_t s) {
    hipLaunchKernelGGL(k_walk_one, dim3(1), dim3(64), 0, s, raw, n_bytes, start, end, out);
    return hipGetLastError();
}
hipError_t launch_rec_offsets(const uint8_t *raw, uint64_t n_bytes, uint32_t n_seg, const uint64_t *seg_entry,
                              const uint64_t *seg_base, uint64_t *rec_off, unsigned long long *bad, hipStream_t s) {
    if (!n_seg) return hipSuccess;
    hipLaunchKernelGGL(k_rec_offsets, dim3((n_seg + 255) / 256), dim3(256), 0, s, raw, n_bytes, n_seg, seg_entry, seg_base,
                       rec_off, bad);
    return hipGetLastError();
}
hipError_t launch_count_below_u64(const uint64_t *a, uint64_t n, uint64_t value, unsigned long long *out, hipStream_t s) {
    hipLaunchKernelGGL(k_count_below_u64, dim3(1), dim3(64), 0, s, a, n, value, out);
    return hipGetLastError();
}
hipError_t launch_rec_fixed(const uint8_t *raw, const uint64_t *rec_off, uint64_t n, const RecColumns &c,
                            unsigned long long *stats, hipStream_t s) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(k_rec_fixed, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, raw, rec_off, n, c, stats);
    return hipGetLastError();
}
hipError_t launch_rec_lengths(const uint8_t *raw, const uint64_t *rec_off, uint64_t n, uint64_t *seq_len,
                              uint64_t *qual_len, uint64_t *cig_len, hipStream_t s) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(k_rec_lengths, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, raw, rec_off, n, seq_len,
                       qual_len, cig_len);
    return hipGetLastError();
}
hipError_t launch_exclusive_scan_u64(uint64_t *data, uint64_t n_plus_1, void *tmp, size_t *tmp_bytes, hipStream_t s) {
    return hipcub::DeviceScan::ExclusiveSum(tmp, *tmp_bytes, data, data, (int)n_plus_1, s);
}
hipError_t launch_rec_var(const uint8_t *raw, const uint64_t *rec_off, uint64_t n, const RecColumns &c, uint64_t seq_bytes,
                          uint64_t qual_bytes, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_rec_var, dim3((uint32_t)((n * 16 + 255) / 256)), dim3(256), 0, s, raw, rec_off, n, c);
    hipLaunchKernelGGL(k_fill_slack, dim3(1), dim3(64), 0, s, c.seq + seq_bytes, c.qual + qual_bytes);
    return hipGetLastError();
}

} // namespace ngsq
