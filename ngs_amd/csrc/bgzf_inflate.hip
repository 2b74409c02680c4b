// bgzf_inflate.hip -- DEFLATE (RFC 1951) decoder for BGZF blocks on gfx950.
//
// Replaces, for the device ingest path, the inflate the reference gets from noodles-bgzf 0.20
// (flate2 1.0.24 / miniz_oxide 0.5.4) under `reader.records(&header)`, src/qc/command.rs:305 and
// src/utils/formats/bam.rs:32-56.  Written from RFC 1951 and the SAM/BAM specification 4.1.
//
// One WAVEFRONT per BGZF block (blocks are independent gzip members of <= 64 KiB):
//  * the bit stream is decoded by the whole wave in lock step -- every lane holds the same
//    decoder state, forced into SGPRs with readfirstlane/readlane, so the serial part runs
//    on the scalar unit and the vector unit is used where DEFLATE is parallel:
//      - the compressed bytes are fetched 256 B at a time, one dword per lane (coalesced,
//        one buffer ahead), and handed to the bit reader with v_readlane;
//      - Huffman tables are built by the 64 lanes (ballot-ranked canonical sort, parallel fill);
//      - LZ77 matches are copied by the lanes, 64 bytes per step;
//      - the finished block leaves LDS as coalesced dword stores.
//  * the whole output of the block (<= 64 KiB) lives in LDS, so a match never reads HBM.
//  * CRC32 of the block (gzip trailer) is verified on request: 64 slices in parallel, then
//    combined with GF(2) polynomial multiplication.
#include <hip/hip_runtime.h>

#include "ingest_kernels.h"

namespace ngsq {

namespace {

constexpr uint32_t LB = 10, DB = 8; // bits of the primary lookup tables
constexpr uint32_t OUT_CAP = 65536;

// table entry: value << 16 | extra_bits << 8 | kind << 5 | code_bits
constexpr uint32_t K_LIT = 0, K_EOB = 1, K_BASE = 2, K_ESC = 3, K_INVALID = 7;
__device__ __forceinline__ constexpr uint32_t mk_entry(uint32_t value, uint32_t extra, uint32_t kind, uint32_t bits) {
    return value << 16 | extra << 8 | kind << 5 | bits;
}

__constant__ uint16_t c_len_base[31] = {3,  4,  5,  6,  7,  8,  9,  10, 11,  13,  15,  17,  19,  23, 27, 31,
                                        35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258, 0,  0};
__constant__ uint8_t c_len_extra[31] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2,
                                        3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0, 0, 0};
__constant__ uint16_t c_dist_base[32] = {1,   2,   3,   4,   5,   7,    9,    13,   17,   25,   33,   49,   65,    97,    129, 193,
                                         257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577, 0,   0};
__constant__ uint8_t c_dist_extra[32] = {0, 0, 0, 0, 1, 1, 2, 2,  3,  3,  4,  4,  5,  5,  6, 6,
                                         7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13, 0, 0};
__constant__ uint8_t c_cl_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

struct Lds {
    uint8_t out[OUT_CAP];
    uint32_t lit_tab[1u << LB];
    uint32_t dist_tab[1u << DB]; // also the code-length-code table while a dynamic header is read
    uint32_t crc_tab[256];
    uint32_t cnt[2][16];   // codes per length: [0] literal/length, [1] distance (or code-length code)
    uint32_t start[2][16]; // first index in syms of each length
    uint32_t fcode[2][16]; // first canonical code of each length
    uint16_t syms[2][288]; // symbols in canonical order
    uint8_t lens[320];
};

__device__ __forceinline__ uint32_t uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }

// ---- bit reader: uniform state, input staged one dword per lane ------------------------------
struct BitReader {
    const uint32_t *src; // dword-aligned start
    uint32_t cur, nxt;   // per lane: dword (base + lane), dword (base + 64 + lane)
    uint32_t base;       // dword index of lane 0 of cur
    uint32_t idx;        // next dword to hand to the bit buffer
    uint64_t buf;
    uint32_t cnt;

    __device__ void init(const uint8_t *p) {
        const uintptr_t a = reinterpret_cast<uintptr_t>(p);
        src = reinterpret_cast<const uint32_t *>(a & ~(uintptr_t)3);
        seek((uint32_t)(a & 3));
    }
    // continue at byte `b` (counted from the dword-aligned origin `src`)
    __device__ void seek(uint32_t b) {
        base = idx = b >> 2;
        cur = src[base + (threadIdx.x & 63)];
        nxt = src[base + 64 + (threadIdx.x & 63)];
        buf = 0;
        cnt = 0;
        refill();
        drop((b & 3) * 8);
        refill();
    }
    // make at least 33 bits available
    __device__ __forceinline__ void refill() {
        if (cnt <= 32) {
            if (idx - base >= 64) {
                base += 64;
                cur = nxt;
                nxt = src[base + 64 + (threadIdx.x & 63)];
            }
            const uint32_t w = __builtin_amdgcn_readlane(cur, idx - base);
            buf |= (uint64_t)w << cnt;
            cnt += 32;
            idx += 1;
        }
    }
    __device__ __forceinline__ uint32_t peek(uint32_t n) const { return (uint32_t)buf & ((1u << n) - 1u); }
    __device__ __forceinline__ void drop(uint32_t n) {
        buf >>= n;
        cnt -= n;
    }
    __device__ __forceinline__ uint32_t take(uint32_t n) {
        const uint32_t v = peek(n);
        drop(n);
        return v;
    }
    // bits consumed since init, counted from the dword-aligned start
    __device__ uint64_t consumed_bits() const { return (uint64_t)idx * 32 - cnt; }
};

// ---- canonical Huffman tables ----------------------------------------------------------------
// lens[0..n) -> primary table of TB bits + (cnt, syms) for codes longer than TB.  which: 0 = lit/len, 1 = dist/cl.
// kind_of: 0 literal/length alphabet, 1 distance alphabet, 2 code-length alphabet.
// Returns false for an over-subscribed set.
__device__ bool build_table(Lds &L, const uint8_t *lens, uint32_t n, uint32_t which, uint32_t TB, uint32_t *tab,
                            uint32_t alphabet, uint32_t lane) {
    if (lane < 16) L.cnt[which][lane] = 0;
    for (uint32_t i = lane; i < (1u << TB); i += 64) tab[i] = mk_entry(0, 0, K_INVALID, 0);
    __syncthreads();
    for (uint32_t i = lane; i < n; i += 64) atomicAdd(&L.cnt[which][lens[i]], 1u);
    __syncthreads();
    // offsets and first codes (uniform, registers)
    uint32_t off[16], code = 0, index = 0;
    int32_t left = 1;
    bool ok = true;
#pragma unroll
    for (uint32_t l = 1; l < 16; l++) {
        const uint32_t c = uni(L.cnt[which][l]);
        left = (left << 1) - (int32_t)c;
        if (left < 0) ok = false;
        code <<= 1;
        off[l] = index;
        if (lane == 0) {
            L.start[which][l] = index;
            L.fcode[which][l] = code;
        }
        code += c;
        index += c;
    }
    if (!ok) return false;
    // canonical order: by length, then by symbol.  Rank inside a 64-symbol chunk by ballot.
    for (uint32_t b = 0; b < n; b += 64) {
        const uint32_t s = b + lane;
        const uint32_t l = s < n ? lens[s] : 0u;
#pragma unroll
        for (uint32_t k = 1; k < 16; k++) {
            const uint64_t m = __ballot(l == k);
            if (l == k) L.syms[which][off[k] + __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)s;
            off[k] += __popcll(m);
        }
    }
    __syncthreads();
    // fill: every code of length <= TB owns 2^(TB-len) slots; longer codes mark their prefix slot
    for (uint32_t i = lane; i < index; i += 64) {
        const uint32_t s = L.syms[which][i], l = lens[s];
        const uint32_t c = L.fcode[which][l] + (i - L.start[which][l]);
        const uint32_t rev = __brev(c) >> (32 - l);
        uint32_t e;
        if (alphabet == 0) {
            if (s < 256) e = mk_entry(s, 0, K_LIT, l);
            else if (s == 256) e = mk_entry(0, 0, K_EOB, l);
            else if (s < 286) e = mk_entry(c_len_base[s - 257], c_len_extra[s - 257], K_BASE, l);
            else e = mk_entry(0, 0, K_INVALID, l);
        } else if (alphabet == 1) {
            e = s < 30 ? mk_entry(c_dist_base[s], c_dist_extra[s], K_BASE, l) : mk_entry(0, 0, K_INVALID, l);
        } else {
            e = mk_entry(s, 0, K_LIT, l);
        }
        if (l <= TB) {
            for (uint32_t k = rev; k < (1u << TB); k += 1u << l) tab[k] = e;
        } else {
            tab[rev & ((1u << TB) - 1u)] = mk_entry(0, 0, K_ESC, 0);
        }
    }
    __syncthreads();
    return true;
}

// decode one symbol whose code is longer than the primary table (bit-serial canonical decode);
// returns the symbol or 0xFFFF, and the code length in *bits
__device__ uint32_t slow_symbol(const Lds &L, uint32_t which, uint64_t buf, uint32_t *bits) {
    uint32_t code = 0, first = 0, index = 0;
    for (uint32_t l = 1; l < 16; l++) {
        code |= (uint32_t)(buf >> (l - 1)) & 1u;
        const uint32_t c = uni(L.cnt[which][l]);
        if (code - first < c) {
            *bits = l;
            return uni(L.syms[which][index + (code - first)]);
        }
        index += c;
        first += c;
        first <<= 1;
        code <<= 1;
    }
    *bits = 15;
    return 0xFFFFu;
}

// entry of the next symbol of table `tab`; resolves long codes to a full entry
__device__ __forceinline__ uint32_t next_entry(const Lds &L, const uint32_t *tab, uint32_t TB, uint32_t which,
                                               uint32_t alphabet, const BitReader &br) {
    uint32_t e = uni(tab[br.peek(TB)]);
    if (__builtin_expect(((e >> 5) & 7u) == K_ESC, 0)) {
        uint32_t bits;
        const uint32_t s = slow_symbol(L, which, br.buf, &bits);
        if (s == 0xFFFFu) return mk_entry(0, 0, K_INVALID, 15);
        if (alphabet == 0) {
            if (s < 256) e = mk_entry(s, 0, K_LIT, bits);
            else if (s == 256) e = mk_entry(0, 0, K_EOB, bits);
            else if (s < 286) e = mk_entry(c_len_base[s - 257], c_len_extra[s - 257], K_BASE, bits);
            else e = mk_entry(0, 0, K_INVALID, bits);
        } else {
            e = s < 30 ? mk_entry(c_dist_base[s], c_dist_extra[s], K_BASE, bits) : mk_entry(0, 0, K_INVALID, bits);
        }
    }
    return e;
}

// ---- CRC32 (gzip): GF(2) helpers in the reflected representation -----------------------------
constexpr uint32_t CRC_POLY = 0xEDB88320u;
__device__ uint32_t crc_mul(uint32_t a, uint32_t b) { // a * b mod P
    uint32_t p = 0;
    for (uint32_t m = 1u << 31; m; m >>= 1) {
        if (a & m) p ^= b;
        b = (b & 1u) ? (b >> 1) ^ CRC_POLY : b >> 1;
    }
    return p;
}
__device__ uint32_t crc_xpow8(uint32_t n_bytes) { // x^(8 n) mod P
    uint32_t r = 1u << 31, sq = 0x00800000u;      // x^0, x^8
    for (uint32_t e = n_bytes; e; e >>= 1) {
        if (e & 1u) r = crc_mul(r, sq);
        sq = crc_mul(sq, sq);
    }
    return r;
}

} // namespace

__global__ __launch_bounds__(64) void k_bgzf_inflate(const uint8_t *__restrict__ comp,
                                                     const BgzfBlock *__restrict__ blocks, uint32_t n_blocks,
                                                     uint8_t *__restrict__ out, uint32_t *__restrict__ status,
                                                     uint32_t check_crc) {
    extern __shared__ __align__(16) uint8_t s_raw[];
    Lds &L = *reinterpret_cast<Lds *>(s_raw);
    const uint32_t lane = threadIdx.x;
    const uint32_t bi = blockIdx.x;
    if (bi >= n_blocks) return;
    const BgzfBlock blk = blocks[bi];
    const uint32_t isize = uni(blk.isize), in_len = uni(blk.in_len);
    if (isize == 0 && in_len == 0) {
        if (lane == 0) status[bi] = INF_OK;
        return;
    }
    if (check_crc) {
        // byte-wise CRC table
        for (uint32_t i = lane; i < 256; i += 64) {
            uint32_t c = i;
#pragma unroll
            for (int k = 0; k < 8; k++) c = (c & 1u) ? (c >> 1) ^ CRC_POLY : c >> 1;
            L.crc_tab[i] = c;
        }
    }
    BitReader br;
    br.init(comp + blk.in_off);
    const uint64_t bit_limit = (uint64_t)((reinterpret_cast<uintptr_t>(comp + blk.in_off) & 3u) + in_len) * 8u;

    uint32_t pos = 0, err = INF_OK;
    bool last = false;
    while (!last && err == INF_OK) {
        br.refill();
        last = br.take(1);
        const uint32_t type = br.take(2);
        if (type == 0) {
            // stored: skip to the byte boundary, LEN, NLEN, then LEN raw bytes
            br.drop(br.cnt & 7u);
            br.refill();
            const uint32_t len = br.take(16);
            br.refill();
            const uint32_t nlen = br.take(16);
            if ((len ^ nlen) != 0xFFFFu) {
                err = INF_BAD_STORED_LEN;
                break;
            }
            if (pos + len > isize) {
                err = INF_OUTPUT_OVERRUN;
                break;
            }
            // the bit buffer holds whole bytes now: copy the source bytes directly, then restart behind them
            const uint32_t byte0 = (uint32_t)(br.consumed_bits() >> 3);
            const uint8_t *sp = reinterpret_cast<const uint8_t *>(br.src) + byte0;
            for (uint32_t i = lane; i < len; i += 64) L.out[pos + i] = sp[i];
            pos += len;
            br.seek(byte0 + len);
            continue;
        }
        if (type == 3) {
            err = INF_BAD_BLOCK_TYPE;
            break;
        }
        uint32_t hlit, hdist;
        if (type == 1) {
            // fixed codes (RFC 1951 3.2.6)
            for (uint32_t i = lane; i < 320; i += 64)
                L.lens[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : i < 288 ? 8 : 5;
            hlit = 288;
            hdist = 32;
            __syncthreads();
        } else {
            br.refill();
            hlit = br.take(5) + 257;
            hdist = br.take(5) + 1;
            const uint32_t hclen = br.take(4) + 4;
            if (hlit > 286 || hdist > 30) {
                err = INF_BAD_CODE_LENGTHS;
                break;
            }
            if (lane < 19) L.lens[lane] = 0;
            __syncthreads();
            for (uint32_t i = 0; i < hclen; i++) {
                br.refill();
                const uint32_t v = br.take(3);
                if (lane == 0) L.lens[c_cl_order[i]] = (uint8_t)v;
            }
            __syncthreads();
            if (!build_table(L, L.lens, 19, 1, 7, L.dist_tab, 2, lane)) {
                err = INF_BAD_CODE_LENGTHS;
                break;
            }
            // the code lengths of both alphabets, run-length coded with the code-length code; decoded
            // in place (build_table is done with the code-length-code lengths, whose codes fit the table)
            uint8_t *cl = L.lens;
            const uint32_t total = hlit + hdist;
            uint32_t i = 0, prev = 0;
            while (i < total) {
                br.refill();
                uint32_t e = uni(L.dist_tab[br.peek(7)]);
                const uint32_t kind = (e >> 5) & 7u;
                if (kind != K_LIT) { // the code-length code has at most 7 bits: no long codes
                    err = INF_BAD_CODE_LENGTHS;
                    break;
                }
                br.drop(e & 31u);
                const uint32_t s = e >> 16;
                if (s < 16) {
                    if (lane == 0) cl[i] = (uint8_t)s;
                    prev = s;
                    i += 1;
                    continue;
                }
                uint32_t rep, val;
                if (s == 16) {
                    if (i == 0) {
                        err = INF_BAD_CODE_LENGTHS;
                        break;
                    }
                    rep = 3 + br.take(2);
                    val = prev;
                } else if (s == 17) {
                    rep = 3 + br.take(3);
                    val = 0;
                    prev = 0;
                } else {
                    rep = 11 + br.take(7);
                    val = 0;
                    prev = 0;
                }
                if (i + rep > total) {
                    err = INF_BAD_CODE_LENGTHS;
                    break;
                }
                for (uint32_t k = lane; k < rep; k += 64) cl[i + k] = (uint8_t)val;
                i += rep;
            }
            if (err != INF_OK) break;
            __syncthreads();
            if (uni(L.lens[256]) == 0) { // no end-of-block code
                err = INF_BAD_CODE_LENGTHS;
                break;
            }
        }
        if (!build_table(L, L.lens, hlit, 0, LB, L.lit_tab, 0, lane) ||
            !build_table(L, L.lens + hlit, hdist, 1, DB, L.dist_tab, 1, lane)) {
            err = INF_BAD_CODE_LENGTHS;
            break;
        }
        // ---- the symbol loop
        for (;;) {
            br.refill();
            const uint32_t e = next_entry(L, L.lit_tab, LB, 0, 0, br);
            const uint32_t kind = (e >> 5) & 7u;
            br.drop(e & 31u);
            if (kind == K_LIT) {
                if (pos >= isize) {
                    err = INF_OUTPUT_OVERRUN;
                    break;
                }
                if (lane == 0) L.out[pos] = (uint8_t)(e >> 16);
                pos += 1;
                continue;
            }
            if (kind == K_EOB) break;
            if (kind != K_BASE) {
                err = INF_BAD_SYMBOL;
                break;
            }
            const uint32_t len = (e >> 16) + br.take((e >> 8) & 15u);
            br.refill();
            const uint32_t d = next_entry(L, L.dist_tab, DB, 1, 1, br);
            if (((d >> 5) & 7u) != K_BASE) {
                err = INF_BAD_SYMBOL;
                break;
            }
            br.drop(d & 31u);
            const uint32_t dist = (d >> 16) + br.take((d >> 8) & 15u);
            if (dist > pos) {
                err = INF_BAD_DISTANCE;
                break;
            }
            if (pos + len > isize) {
                err = INF_OUTPUT_OVERRUN;
                break;
            }
            // the source run [pos - dist, pos) is final: byte i of the match is its byte i mod dist
            const uint32_t from = pos - dist;
            if (dist >= len) {
                for (uint32_t i = lane; i < len; i += 64) L.out[pos + i] = L.out[from + i];
            } else {
                for (uint32_t i = lane; i < len; i += 64) L.out[pos + i] = L.out[from + i % dist];
            }
            pos += len;
        }
    }
    if (err == INF_OK && pos != isize) err = INF_SIZE_MISMATCH;
    if (err == INF_OK && br.consumed_bits() > bit_limit) err = INF_INPUT_OVERRUN;
    __syncthreads();

    if (err == INF_OK && check_crc) {
        // 64 slices of S bytes (S/4 odd: the lanes read distinct LDS banks), then combine
        const uint32_t S = (((isize + 63) / 64 + 3) / 4 | 1u) * 4;
        const uint32_t lo = min(lane * S, isize), hi = min(lo + S, isize);
        uint32_t c = 0xFFFFFFFFu;
        for (uint32_t i = lo; i < hi; i++) c = L.crc_tab[(c ^ L.out[i]) & 0xFFu] ^ (c >> 8);
        c = ~c; // CRC of the slice (of the empty string: 0)
        const uint32_t xs = crc_xpow8(S);
        uint32_t acc = 0;
        for (uint32_t k = 0; k < 64; k++) {
            const uint32_t ck = __builtin_amdgcn_readlane(c, k);
            const uint32_t lk = min(k * S, isize), hk = min(lk + S, isize);
            if (hk == lk) break;
            // crc(A || B) = crc(A) * x^(8 |B|) + crc(B)
            acc = crc_mul(hk - lk == S ? xs : crc_xpow8(hk - lk), acc) ^ ck;
        }
        if (acc != uni(blk.crc)) err = INF_CRC_MISMATCH;
    }
    if (lane == 0) status[bi] = err;
    if (err != INF_OK) return;

    // ---- LDS -> HBM, aligned dword stores
    uint8_t *dst = out + blk.out_off;
    const uint32_t head = min((uint32_t)((4u - (reinterpret_cast<uintptr_t>(dst) & 3u)) & 3u), isize);
    if (lane < head) dst[lane] = L.out[lane];
    const uint32_t body = (isize - head) / 4;
    uint32_t *dw = reinterpret_cast<uint32_t *>(dst + head);
    const uint32_t *lw = reinterpret_cast<const uint32_t *>(L.out);
    for (uint32_t j = lane; j < body; j += 64) {
        const uint32_t k = head + 4 * j; // LDS byte offset of this dword
        const uint32_t w0 = lw[k >> 2], w1 = lw[(k >> 2) + 1 < OUT_CAP / 4 ? (k >> 2) + 1 : k >> 2];
        dw[j] = __builtin_amdgcn_alignbyte(w1, w0, k & 3u);
    }
    const uint32_t tail0 = head + 4 * body;
    if (tail0 + lane < isize) dst[tail0 + lane] = L.out[tail0 + lane];
}

hipError_t launch_bgzf_inflate(const uint8_t *comp, const BgzfBlock *blocks, uint32_t n_blocks, uint8_t *out,
                               uint32_t *status, bool check_crc, hipStream_t s) {
    if (!n_blocks) return hipSuccess;
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_bgzf_inflate),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(Lds));
        if (e != hipSuccess) return e;
        attr = true;
    }
    hipLaunchKernelGGL(k_bgzf_inflate, dim3(n_blocks), dim3(64), sizeof(Lds), s, comp, blocks, n_blocks, out, status,
                       check_crc ? 1u : 0u);
    return hipGetLastError();
}

} // namespace ngsq
