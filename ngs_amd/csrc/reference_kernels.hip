// reference_kernels.hip -- FASTA text -> one 4-bit BAM base code per base, on the device (include/ngsq_reference.h).
//
// The host uploads the TEXT of the wanted records (their sequence lines, line terminators included) and a table of where each
// lies; three launches do what noodles-fasta's reader and noodles-sam's Base::try_from do per byte in the reference
// (edits.rs:185-205 record.sequence(), edits.rs:257-261 Base::try_from):
//   k_fasta_count   bases per 4 KiB tile of text (a byte is dropped iff it is '\n', or a '\r' in front of a '\n' / of the end)
//   k_fasta_scan    per record: exclusive prefix of its tiles' counts -> where each tile's bases go; the record's length
//   k_fasta_emit    the bases of every tile as codes, compacted through LDS and stored with coalesced dwords; bytes that are
//                   no base letter are listed (record, 1-based position) -- only a read that covers one fails
// All HBM-bound byte work: 3.1 GB of text in, 3.1 GB of codes out, a few milliseconds per genome.
#include <hip/hip_runtime.h>

#include "reference_kernels.h"

namespace ngsq {
namespace {

constexpr uint32_t TILE = FASTA_TILE; // bytes of text per tile
constexpr uint32_t THREADS = 256;     // 16 bytes per thread

// Base::try_from(u8) (noodles-sam 0.25 record/sequence/base.rs: the char, upper-cased, is one of "=ACMGRSVTWYHKDBN")
__host__ __device__ inline uint32_t base_code_of(uint32_t c) {
    if (c >= 'a' && c <= 'z') c -= 32u;
    switch (c) {
    case '=': return 0;
    case 'A': return 1;
    case 'C': return 2;
    case 'M': return 3;
    case 'G': return 4;
    case 'R': return 5;
    case 'S': return 6;
    case 'V': return 7;
    case 'T': return 8;
    case 'W': return 9;
    case 'Y': return 10;
    case 'H': return 11;
    case 'K': return 12;
    case 'D': return 13;
    case 'B': return 14;
    case 'N': return 15;
    default: return 0xFFu;
    }
}

// which record a tile belongs to: the last s with tile_first[s] <= tile
__device__ __forceinline__ uint32_t seq_of_tile(const uint64_t *__restrict__ tile_first, uint32_t n_seq, uint64_t tile) {
    uint32_t lo = 0, hi = n_seq; // tile_first[lo] <= tile < tile_first[hi]
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (tile_first[mid] <= tile) lo = mid;
        else hi = mid;
    }
    return lo;
}

// the 16 bytes of text this thread owns and the 16-bit mask of the ones that are bases (or invalid bytes: anything kept)
struct Own {
    uint4 v;
    uint32_t keep; // bit j: byte j is part of the sequence
    uint32_t n;    // bytes of the tile this thread has (0..16)
};
__device__ __forceinline__ Own load_own(const uint8_t *__restrict__ text, const FastaSeqDev &sq, uint64_t tile_in_seq, uint32_t t) {
    Own o{};
    const uint64_t off = tile_in_seq * TILE + 16ull * t; // in the record's text
    if (off >= sq.text_len) return o;
    const uint64_t left = sq.text_len - off;
    o.n = left < 16 ? (uint32_t)left : 16u;
    const uint8_t *p = text + sq.text_off + off; // (text_off is 256-aligned, the buffer is readable 64 bytes past its end)
    o.v = *reinterpret_cast<const uint4 *>(p);
    const uint32_t next = left > 16 ? p[16] : 0x100u; // the byte behind this thread's 16; 0x100 = the end of the record's text
    const uint32_t w[4] = {o.v.x, o.v.y, o.v.z, o.v.w};
#pragma unroll
    for (uint32_t j = 0; j < 16; j++) {
        if (j >= o.n) break;
        const uint32_t c = (w[j >> 2] >> (8u * (j & 3u))) & 0xFFu;
        uint32_t nx;
        if (j + 1 < o.n) nx = (w[(j + 1) >> 2] >> (8u * ((j + 1) & 3u))) & 0xFFu;
        else nx = (j + 1 == 16) ? next : 0x100u;
        // a line terminator is "\n" or "\r\n" (noodles-fasta strips both); a '\r' anywhere else stays a byte of the sequence
        const bool drop = c == '\n' || (c == '\r' && (nx == '\n' || nx == 0x100u));
        if (!drop) o.keep |= 1u << j;
    }
    return o;
}

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, uint32_t lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t u = (uint32_t)__shfl_up((int)v, o, 64);
        if (lane >= (uint32_t)o) v += u;
    }
    return v;
}

__global__ __launch_bounds__(THREADS) void k_fasta_count(const uint8_t *__restrict__ text, const FastaSeqDev *__restrict__ seqs, uint32_t n_seq,
                                                         const uint64_t *__restrict__ tile_first, uint64_t n_tiles, uint32_t *__restrict__ counts) {
    __shared__ uint32_t s_w[THREADS / 64];
    const uint32_t t = threadIdx.x, lane = t & 63;
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint32_t s = seq_of_tile(tile_first, n_seq, tile);
        const FastaSeqDev sq = seqs[s];
        const Own o = load_own(text, sq, tile - tile_first[s], t);
        uint32_t c = (uint32_t)__popc(o.keep);
#pragma unroll
        for (int k = 32; k > 0; k >>= 1) c += (uint32_t)__shfl_xor((int)c, k, 64);
        if (lane == 0) s_w[t >> 6] = c;
        __syncthreads();
        if (t == 0) counts[tile] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
        __syncthreads();
    }
}

// one block per record: where each of its tiles' bases go
__global__ __launch_bounds__(1024) void k_fasta_scan(const uint32_t *__restrict__ counts, const uint64_t *__restrict__ tile_first, uint32_t n_seq,
                                                      uint64_t *__restrict__ tile_base, unsigned long long *__restrict__ seq_len) {
    __shared__ uint32_t s_w[16];
    __shared__ unsigned long long s_carry;
    const uint32_t s = blockIdx.x, t = threadIdx.x, lane = t & 63, w = t >> 6;
    if (s >= n_seq) return;
    const uint64_t t0 = tile_first[s], n = tile_first[s + 1] - t0;
    if (t == 0) s_carry = 0;
    __syncthreads();
    for (uint64_t base = 0; base < n; base += 1024) {
        const uint64_t i = base + t;
        const uint32_t v = i < n ? counts[t0 + i] : 0u;
        const uint32_t inc = wave_incl_scan(v, lane);
        if (lane == 63) s_w[w] = inc;
        __syncthreads();
        uint32_t before = 0;
        for (uint32_t k = 0; k < w; k++) before += s_w[k];
        const unsigned long long carry = s_carry;
        if (i < n) tile_base[t0 + i] = carry + before + (inc - v);
        __syncthreads();
        if (t == 1023) s_carry = carry + before + inc;
        __syncthreads();
    }
    if (t == 0) seq_len[s] = s_carry;
}

__global__ __launch_bounds__(THREADS) void k_fasta_emit(const uint8_t *__restrict__ text, const FastaSeqDev *__restrict__ seqs, uint32_t n_seq,
                                                        const uint64_t *__restrict__ tile_first, uint64_t n_tiles,
                                                        const uint64_t *__restrict__ tile_base, uint8_t *__restrict__ codes,
                                                        unsigned long long *__restrict__ n_bad, unsigned long long *__restrict__ bad_list,
                                                        uint32_t bad_cap) {
    __shared__ uint32_t s_w[THREADS / 64];
    __shared__ __attribute__((aligned(16))) uint8_t s_out[TILE + 16];
    const uint32_t t = threadIdx.x, lane = t & 63, w = t >> 6;
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint32_t s = seq_of_tile(tile_first, n_seq, tile);
        const FastaSeqDev sq = seqs[s];
        const Own o = load_own(text, sq, tile - tile_first[s], t);
        const uint32_t c = (uint32_t)__popc(o.keep);
        const uint32_t inc = wave_incl_scan(c, lane);
        if (lane == 63) s_w[w] = inc;
        __syncthreads();
        uint32_t before = 0, total = 0;
#pragma unroll
        for (uint32_t k = 0; k < THREADS / 64; k++) {
            if (k < w) before += s_w[k];
            total += s_w[k];
        }
        const uint64_t first = tile_base[tile]; // 0-based index in the record of the tile's first base
        uint32_t at = before + inc - c;
        const uint32_t ww[4] = {o.v.x, o.v.y, o.v.z, o.v.w};
#pragma unroll
        for (uint32_t j = 0; j < 16; j++) {
            if (!((o.keep >> j) & 1u)) continue;
            const uint32_t ch = (ww[j >> 2] >> (8u * (j & 3u))) & 0xFFu;
            uint32_t code = base_code_of(ch);
            if (code > 15u) { // no base letter: the position is remembered (1-based), the slot holds N
                const unsigned long long k = atomicAdd(n_bad, 1ull);
                if (k < bad_cap) bad_list[k] = (unsigned long long)s << 40 | (first + at + 1);
                code = 15u;
            }
            s_out[at++] = (uint8_t)code;
        }
        __syncthreads();
        // the tile's codes -> codes[sq.text_off + first ...): bytes up to the first dword boundary, dwords, the tail's bytes
        uint8_t *const dst = codes + sq.text_off + first;
        const uint32_t head = min(total, (uint32_t)((4u - ((uint32_t)(uintptr_t)dst & 3u)) & 3u));
        if (t < head) dst[t] = s_out[t];
        const uint32_t n_dw = (total - head) >> 2;
        for (uint32_t k = t; k < n_dw; k += THREADS) {
            const uint32_t q = head + 4u * k;
            const uint32_t v = (uint32_t)s_out[q] | (uint32_t)s_out[q + 1] << 8 | (uint32_t)s_out[q + 2] << 16 | (uint32_t)s_out[q + 3] << 24;
            *reinterpret_cast<uint32_t *>(dst + q) = v;
        }
        const uint32_t tail0 = head + 4u * n_dw;
        if (t < total - tail0) dst[tail0 + t] = s_out[tail0 + t];
        __syncthreads();
    }
}

} // namespace

int fasta_base_code(uint8_t byte) {
    const uint32_t c = base_code_of(byte);
    return c > 15u ? -1 : (int)c;
}

hipError_t launch_fasta_convert(const LaunchInfo &li, const uint8_t *text, const FastaSeqDev *seqs, uint32_t n_seq, const uint64_t *tile_first,
                                uint64_t n_tiles, uint32_t *counts, uint64_t *tile_base, unsigned long long *seq_len, uint8_t *codes,
                                unsigned long long *n_bad, unsigned long long *bad_list, uint32_t bad_cap, hipStream_t s) {
    if (!n_seq || !n_tiles) return hipSuccess;
    const uint32_t grid = (uint32_t)std::min<uint64_t>(n_tiles, (uint64_t)li.n_cu * 16);
    hipLaunchKernelGGL(k_fasta_count, dim3(grid), dim3(THREADS), 0, s, text, seqs, n_seq, tile_first, n_tiles, counts);
    hipLaunchKernelGGL(k_fasta_scan, dim3(n_seq), dim3(1024), 0, s, counts, tile_first, n_seq, tile_base, seq_len);
    hipLaunchKernelGGL(k_fasta_emit, dim3(grid), dim3(THREADS), 0, s, text, seqs, n_seq, tile_first, n_tiles, tile_base, codes, n_bad, bad_list, bad_cap);
    return hipGetLastError();
}

} // namespace ngsq
