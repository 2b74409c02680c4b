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
//      - (the compressed bytes themselves are read with scalar loads, one dword ahead);
//      - Huffman tables are built by the 64 lanes (ballot-ranked canonical sort, parallel fill);
//      - LZ77 matches are copied by the lanes, 64 bytes per step;
//      - the finished block leaves LDS as coalesced dword stores.
//  * the last 32 KiB of output (DEFLATE's window) live in an LDS ring, so a match never reads
//    HBM; finished 16 KiB pieces leave the ring as coalesced dword stores.  39 KiB of LDS per
//    wave: four waves per CU, one per SIMD -- the decoder is bound by instruction issue (one
//    instruction per wave every four cycles), not by memory.
//  * literals are collected in a register, one byte per lane (v_writelane), and reach the ring
//    64 at a time; table entries carry two literals when both codes fit the lookup index.
//  * CRC32 of the block (gzip trailer) is verified on request: 64 slices per piece in
//    parallel, combined with GF(2) polynomial multiplication.
#include <hip/hip_runtime.h>

#include "ingest_kernels.h"

namespace ngsq {

namespace {

constexpr uint32_t LB = 10, DB = 8; // bits of the primary lookup tables
constexpr uint32_t RING = 32768, RMASK = RING - 1; // the DEFLATE window
constexpr uint32_t PIECE = 16384;                  // bytes that leave the ring together

// table entry: value << 16 | extra_bits << 8 | kind << 5 | code_bits
// K_LIT2: two literals (bits 16..23, then 24..31), code_bits = both codes.  Literal <=> (kind & 3) == 0.
constexpr uint32_t K_LIT = 0, K_EOB = 1, K_BASE = 2, K_ESC = 3, K_LIT2 = 4, K_INVALID = 7;
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
    uint8_t ring[RING];
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
// v[lane] = value (both uniform): one v_writelane_b32, no EXEC change, no memory
__device__ __forceinline__ void write_lane(uint32_t &v, uint32_t value, uint32_t lane) {
    // two SGPR sources would break the constant-bus rule: the lane select goes through M0
    asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(v) : "s"(value), "s"(lane) : "m0"); // NOLINT
}

// ---- bit reader: every field is uniform (SGPRs) ----------------------------------------------
// The compressed bytes are read with SCALAR loads, one dword ahead of the bit buffer: the load for
// the next refill is issued by this one and waited for together with the next table lookup.
// constant address space: the compressed buffer is never written while the kernel runs, and this
// is what lets the compiler use s_load_dword for it
typedef const __attribute__((address_space(4))) uint32_t const_u32;

struct BitReader {
    const_u32 *src; // dword-aligned origin (read-only, uniform address: s_load_dword)
    uint32_t idx;        // dword that `ahead` holds
    uint32_t ahead;      // src[idx], already loaded
    uint64_t buf;
    uint32_t cnt;

    __device__ void init(const uint8_t *p) {
        const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(p) & 3u);
        src = (const_u32 *)(reinterpret_cast<uintptr_t>(p - mis));
        seek(mis);
    }
    // continue at byte `b` (counted from the dword-aligned origin `src`)
    __device__ void seek(uint32_t b) {
        idx = b >> 2;
        ahead = src[idx];
        buf = 0;
        cnt = 0;
        refill();
        drop((b & 3) * 8);
        refill();
    }
    // make at least 33 bits available
    __device__ __forceinline__ void refill() {
        if (cnt <= 32) {
            buf |= (uint64_t)ahead << cnt;
            cnt += 32;
            idx += 1;
            ahead = src[idx];
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
    // bits consumed, counted from the dword-aligned origin (`ahead` is not in the buffer yet)
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

// After build_table(lit): pair up literals.  Index i starts with a literal of L1 bits; if the code
// that follows is decided by the remaining LB - L1 bits and is a literal too, the entry takes both.
__device__ void pair_literals(Lds &L, uint32_t lane) {
    uint32_t ne[(1u << LB) / 64];
#pragma unroll
    for (uint32_t k = 0; k < (1u << LB) / 64; k++) {
        const uint32_t i = k * 64 + lane;
        uint32_t e = L.lit_tab[i];
        const uint32_t l1 = e & 31u;
        if (((e >> 5) & 7u) == K_LIT && l1 < LB) {
            const uint32_t e2 = L.lit_tab[i >> l1];
            const uint32_t l2 = e2 & 31u;
            if (((e2 >> 5) & 7u) == K_LIT && l1 + l2 <= LB)
                e = ((e2 >> 16) & 0xFFu) << 24 | ((e >> 16) & 0xFFu) << 16 | K_LIT2 << 5 | (l1 + l2);
        }
        ne[k] = e;
    }
    __syncthreads();
#pragma unroll
    for (uint32_t k = 0; k < (1u << LB) / 64; k++) L.lit_tab[k * 64 + lane] = ne[k];
    __syncthreads();
}

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
    uint32_t xs_full = 0; // x^(8 * slice) for full pieces
    constexpr uint32_t SLICE = 260; // PIECE / 64 rounded up to 4 * odd: the lanes' slices start in distinct banks
    if (check_crc) {
        // byte-wise CRC table
        for (uint32_t i = lane; i < 256; i += 64) {
            uint32_t c = i;
#pragma unroll
            for (int k = 0; k < 8; k++) c = (c & 1u) ? (c >> 1) ^ CRC_POLY : c >> 1;
            L.crc_tab[i] = c;
        }
        xs_full = crc_xpow8(SLICE);
    }
    BitReader br;
    br.init(comp + blk.in_off);
    const uint64_t bit_limit = (uint64_t)((reinterpret_cast<uintptr_t>(comp + blk.in_off) & 3u) + in_len) * 8u;
    uint8_t *const gdst = out + blk.out_off;

    // output state: bytes [0, spos) are in the ring, [spos, pos) are staged in `lit` (lane = position & 63),
    // [0, flushed) have left for HBM
    uint32_t pos = 0, spos = 0, flushed = 0, err = INF_OK, crc = 0;
    uint32_t lit = 0;

    auto flush_stage = [&]() {
        const uint32_t p = (spos & ~63u) + lane;
        if (p >= spos && p < pos) L.ring[p & RMASK] = (uint8_t)lit;
        spos = pos;
    };
    // ring bytes [flushed, flushed + n) -> HBM (and into the running CRC)
    auto flush_piece = [&](uint32_t n) {
        if (check_crc) {
            const uint32_t lo = min(lane * SLICE, n), hi = min(lo + SLICE, n);
            uint32_t c = 0xFFFFFFFFu;
            for (uint32_t i = lo; i < hi; i++) c = L.crc_tab[(c ^ L.ring[(flushed + i) & RMASK]) & 0xFFu] ^ (c >> 8);
            c = ~c; // CRC of the slice (of the empty string: 0)
            for (uint32_t k = 0; k < 64; k++) {
                const uint32_t lk = min(k * SLICE, n), hk = min(lk + SLICE, n);
                if (hk == lk) break;
                // crc(A || B) = crc(A) * x^(8 |B|) + crc(B)
                crc = crc_mul(hk - lk == SLICE ? xs_full : crc_xpow8(hk - lk), crc) ^ __builtin_amdgcn_readlane(c, k);
            }
        }
        uint8_t *dst = gdst + flushed;
        const uint32_t head = min((uint32_t)((4u - (reinterpret_cast<uintptr_t>(dst) & 3u)) & 3u), n);
        if (lane < head) dst[lane] = L.ring[(flushed + lane) & RMASK];
        const uint32_t body = (n - head) / 4;
        uint32_t *dw = reinterpret_cast<uint32_t *>(dst + head);
        const uint32_t *lw = reinterpret_cast<const uint32_t *>(L.ring);
        for (uint32_t j = lane; j < body; j += 64) {
            const uint32_t k = flushed + head + 4 * j; // stream offset of this dword
            const uint32_t w0 = lw[(k & RMASK) >> 2], w1 = lw[((k + 4) & RMASK) >> 2];
            dw[j] = __builtin_amdgcn_alignbyte(w1, w0, k & 3u);
        }
        const uint32_t tail0 = head + 4 * body;
        if (tail0 + lane < n) dst[tail0 + lane] = L.ring[(flushed + tail0 + lane) & RMASK];
        flushed += n;
    };

    bool last = false;
    while (!last && err == INF_OK) {
        br.refill();
        last = br.take(1);
        const uint32_t type = br.take(2);
        if (br.consumed_bits() > bit_limit) {
            err = INF_INPUT_OVERRUN;
            break;
        }
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
            flush_stage();
            // the bit buffer holds whole bytes now: copy the source bytes directly, then restart behind them
            const uint32_t byte0 = (uint32_t)(br.consumed_bits() >> 3);
            const uint8_t *sp = comp + blk.in_off - (reinterpret_cast<uintptr_t>(comp + blk.in_off) & 3u) + byte0;
            for (uint32_t done = 0; done < len;) {
                const uint32_t n = min(len - done, PIECE - (pos - flushed));
                for (uint32_t i = lane; i < n; i += 64) L.ring[(pos + i) & RMASK] = sp[done + i];
                pos += n;
                done += n;
                if (pos - flushed >= PIECE) flush_piece(PIECE);
            }
            spos = pos;
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
        pair_literals(L, lane);
        // ---- the symbol loop
        for (;;) {
            br.refill();
            const uint32_t e = next_entry(L, L.lit_tab, LB, 0, 0, br);
            br.drop(e & 31u);
            if ((e & (3u << 5)) == 0) {
                // one or two literals: into the staging register, lane = output position & 63
                write_lane(lit, (e >> 16) & 0xFFu, pos & 63u);
                pos += 1;
                if (e & (4u << 5)) {
                    if ((pos & 63u) == 0) flush_stage();
                    write_lane(lit, e >> 24, pos & 63u);
                    pos += 1;
                }
                if ((pos & 63u) == 0) {
                    flush_stage();
                    if (pos > isize) { // also bounds the work on a corrupt stream
                        err = INF_OUTPUT_OVERRUN;
                        break;
                    }
                    if (pos - flushed >= PIECE) flush_piece(PIECE);
                }
                continue;
            }
            const uint32_t kind = (e >> 5) & 7u;
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
            flush_stage();
            // the source run [pos - dist, pos) is final: byte i of the match is its byte i mod dist
            const uint32_t from = pos - dist;
            if (dist >= len) {
                for (uint32_t i = lane; i < len; i += 64) L.ring[(pos + i) & RMASK] = L.ring[(from + i) & RMASK];
            } else {
                for (uint32_t i = lane; i < len; i += 64) L.ring[(pos + i) & RMASK] = L.ring[(from + i % dist) & RMASK];
            }
            pos += len;
            spos = pos;
            if (pos - flushed >= PIECE) flush_piece(PIECE);
        }
    }
    flush_stage();
    if (err == INF_OK && pos != isize) err = pos > isize ? INF_OUTPUT_OVERRUN : INF_SIZE_MISMATCH;
    if (err == INF_OK && br.consumed_bits() > bit_limit) err = INF_INPUT_OVERRUN;
    __syncthreads();
    if (err == INF_OK) {
        while (flushed < pos) flush_piece(min(pos - flushed, PIECE));
        if (check_crc && crc != uni(blk.crc)) err = INF_CRC_MISMATCH;
    }
    if (lane == 0) status[bi] = err;
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
