// bgzf_inflate.hip -- DEFLATE (RFC 1951) decoder for BGZF blocks on gfx950.
//
// Replaces, for the device ingest path, the inflate the reference gets from noodles-bgzf 0.20
// (flate2 1.0.24 / miniz_oxide 0.5.4) under `reader.records(&header)`, src/qc/command.rs:305 and
// src/utils/formats/bam.rs:32-56.  Written from RFC 1951 and the SAM/BAM specification 4.1.
//
// One WAVEFRONT per BGZF block (blocks are independent gzip members of <= 64 KiB).  A wave running
// alone on its SIMD issues one instruction every ~5 cycles, so the design minimises instructions
// per symbol and puts all 64 lanes to work on the one bit stream:
//  * SYMBOL WINDOW.  Lane j looks up the Huffman code that WOULD start at bit j of the next 64
//    bits (two LDS gathers: literal/length and distance table, every lane at once).  The real
//    symbol boundaries are then a chain 0 -> L[0] -> L[0]+L[L[0]] ... followed with one
//    v_readlane + add per symbol on the scalar unit; the lanes on the chain write their literals
//    to the output ring in one store.  A length code ends the chain: its extra bits, the distance
//    code and its extra bits are already sitting in the lanes at those bit offsets.
//  * the bit stream lives in SGPRs (four window dwords + four prefetched by scalar loads).
//  * Huffman tables are built by the 64 lanes (ballot-ranked canonical sort, parallel fill).
//  * LZ77 matches are copied by the lanes, 64 bytes per step.  The last 8 KiB of output live in an
//    LDS ring; finished 2 KiB pieces leave it as coalesced dword stores, and a match that reaches
//    further back than the ring reads its source from HBM (the bytes this wave wrote earlier).
//    16 KiB of LDS per wave: ten decoders per CU.
//  * CRC32 of the block (gzip trailer) is verified on request by a second, wide kernel
//    (k_bgzf_crc: 64 slices per block, combined with GF(2) polynomial multiplication).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#include "ingest_kernels.h"

namespace ngsq {

// Build with -DNGSQ_INFLATE_PROFILE to accumulate s_memtime deltas per decoder phase (measurement aid
// for DESIGN.md; the counters are read back by launch_bgzf_inflate and printed to stderr).
#ifdef NGSQ_INFLATE_PROFILE
__device__ unsigned long long g_inflate_prof[16];
#define PROF_DECL unsigned long long prof_t = __builtin_readcyclecounter(), prof_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, prof_cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define PROF(k)                                                     \
    do {                                                            \
        const unsigned long long now_ = __builtin_readcyclecounter(); \
        prof_acc[k] += now_ - prof_t;                               \
        prof_t = now_;                                              \
    } while (0)
#define PROF_COUNT(k, n) prof_cnt[k] += (n)
#define PROF_FLUSH                                                                             \
    do {                                                                                       \
        if (lane == 0)                                                                         \
            for (int k_ = 0; k_ < 8; k_++) {                                                   \
                atomicAdd(&g_inflate_prof[k_], prof_acc[k_]);                                  \
                atomicAdd(&g_inflate_prof[8 + k_], prof_cnt[k_]);                              \
            }                                                                                  \
    } while (0)
#else
#define PROF_DECL
#define PROF(k)
#define PROF_COUNT(k, n)
#define PROF_FLUSH
#endif

namespace {

constexpr uint32_t LB = 10, DB = 8; // bits of the primary lookup tables
// The most recent RING bytes of output live in LDS; older ones (already written to HBM) are read back
// from there when a match reaches that far.  A decoder wave spends most of its time waiting (LDS round
// trips of the dependent window steps, the far reads), so what counts is how many decoders a CU holds, and
// that is set by the LDS per decoder in 1280-byte granules: measured on the synthetic BAM (1.09 GB out,
// 35 % of the matches reach beyond 2 KiB, 19 % beyond 4 KiB, 10 % beyond 8 KiB), same box, kernel time:
// ring 8192 (12 decoders per CU) 24.6 ms, 4096 (17-19) 17.3 ms, 2048 (25) 14.7 ms, 1024 with a 7-bit distance
// table (28, the register limit) 15.7 ms.  (-DNGSQ_INFLATE_RING=... to rebuild with another size, tools/ring_sweep.sh)
#ifndef NGSQ_INFLATE_RING
#define NGSQ_INFLATE_RING 2048
#endif
constexpr uint32_t RING = NGSQ_INFLATE_RING, RMASK = RING - 1;
constexpr uint32_t PIECE = RING / 4;  // bytes that leave the ring together

// Table entries are 16 bits (LDS per decoder is what limits the decoders per CU):
//   literal/length table:  bit 15 = 0: [14:12] kind (LIT / EOB / ESC = code longer than the table / INVALID),
//                                      [11:4] the literal byte, [3:0] code bits
//                          bit 15 = 1: a length code: [14:12] extra bits (0..5), [11:4] base length - 3, [3:0] code bits
//   distance table:        [11:10] kind (BASE / ESC / INVALID), [9:8] m, [7:4] extra bits (0..13), [3:0] code bits;
//                          base distance = 1 + (m << extra bits)  (codes 0, 1: m = 0, 1; code c >= 2: m = 2 + (c & 1))
// The code-length code of a dynamic header uses the literal format (symbol in the byte field).
constexpr uint32_t LK_LIT = 0, LK_EOB = 1, LK_ESC = 2, LK_INVALID = 3;
constexpr uint32_t DK_BASE = 0, DK_ESC = 1, DK_INVALID = 2;
__device__ __forceinline__ constexpr uint32_t lit_entry(uint32_t byte, uint32_t kind, uint32_t bits) {
    return kind << 12 | byte << 4 | bits;
}
__device__ __forceinline__ constexpr uint32_t len_entry(uint32_t base, uint32_t extra, uint32_t bits) {
    return 0x8000u | extra << 12 | (base - 3u) << 4 | bits;
}
__device__ __forceinline__ constexpr uint32_t dist_entry(uint32_t code, uint32_t bits) {
    return code < 2 ? (DK_BASE << 10 | code << 8 | bits) : (DK_BASE << 10 | (2u + (code & 1u)) << 8 | ((code >> 1) - 1u) << 4 | bits);
}
__device__ __forceinline__ constexpr uint32_t dist_special(uint32_t kind, uint32_t bits) { return kind << 10 | bits; }
// fields
__device__ __forceinline__ uint32_t e_bits(uint32_t e) { return e & 15u; }
__device__ __forceinline__ bool e_is_len(uint32_t e) { return (e & 0x8000u) != 0; }
__device__ __forceinline__ uint32_t e_kind(uint32_t e) { return (e >> 12) & 7u; } // of a non-length entry
__device__ __forceinline__ bool e_is_lit(uint32_t e) { return (e & 0xF000u) == 0; }
__device__ __forceinline__ uint32_t e_byte(uint32_t e) { return (e >> 4) & 255u; }
__device__ __forceinline__ uint32_t e_len_extra(uint32_t e) { return e_is_len(e) ? (e >> 12) & 7u : 0u; }
__device__ __forceinline__ uint32_t d_kind(uint32_t d) { return (d >> 10) & 3u; }
__device__ __forceinline__ uint32_t d_extra(uint32_t d) { return (d >> 4) & 15u; }
__device__ __forceinline__ uint32_t d_base(uint32_t d) { return 1u + (((d >> 8) & 3u) << d_extra(d)); }

__constant__ uint16_t c_len_base[31] = {3,  4,  5,  6,  7,  8,  9,  10, 11,  13,  15,  17,  19,  23, 27, 31,
                                        35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258, 0,  0};
__constant__ uint8_t c_len_extra[31] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2,
                                        3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0, 0, 0};
__constant__ uint8_t c_cl_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// Decoded symbols wait in a queue until 64 bytes of output can be produced at once: a 64-bit window holds ~4.6
// symbols = ~17 bytes of output on BAM data, and emitting them window by window ran the 64-lane emit at a quarter of
// its width (a third of all instructions of the kernel).  Entry: [15:0] output offset of the symbol's first byte
// (ISIZE <= 65536), [16] literal, [31:17] the literal byte or distance - 1.
constexpr uint32_t QCAP = 64;
__device__ __forceinline__ uint32_t q_lit(uint32_t at, uint32_t byte) { return at | 0x10000u | byte << 17; }
__device__ __forceinline__ uint32_t q_match(uint32_t at, uint32_t dist) { return at | (dist - 1u) << 17; }

struct Lds {
    uint8_t ring[RING];
    uint16_t lit_tab[1u << LB];
    uint16_t dist_tab[1u << DB]; // also the code-length-code table while a dynamic header is read
    uint32_t in_ring[128]; // two 256-byte chunks of the compressed stream (chunk c in slot c & 1)
    uint32_t cnt[2][16];   // codes per length: [0] literal/length, [1] distance (or code-length code)
    uint16_t start[2][16]; // first index in syms of each length
    uint16_t fcode[2][16]; // first canonical code of each length
    uint16_t syms0[288];   // literal/length symbols in canonical order
    uint16_t syms1[32];    // distance (or code-length) symbols in canonical order
    union {
        uint8_t lens[320]; // code lengths while a block's tables are built
        struct {
            uint8_t mark[64]; // emit: the queued symbol that starts at each byte of a 64-byte output chunk
            uint32_t q[QCAP]; // decoded symbols not yet written out (a circular queue, drained before the tables change)
        } e;
    };
    __device__ __forceinline__ uint16_t *syms(uint32_t which) { return which ? syms1 : syms0; }
    __device__ __forceinline__ const uint16_t *syms(uint32_t which) const { return which ? syms1 : syms0; }
};

__device__ __forceinline__ uint32_t uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
// The serial core of the decoder, seven scalar instructions per symbol: starting at bit s, mark the
// symbol start in `lits` (literals and whole matches alike) and step to the next one (step[s] bits further) until a lane says stop
// (bit 7 of its step) or the window ends (s >= 64).  Hand-scheduled: the compiler's structurised
// control flow needs about twice as many instructions for this loop.
__device__ __forceinline__ void chain_literals(uint32_t step, uint32_t &s, uint64_t &lits) {
    uint32_t st;
    asm volatile("1:\n\t"
                 "v_readlane_b32 %[st], %[step], %[s]\n\t"
                 "s_bitcmp1_b32 %[st], 7\n\t"
                 "s_cbranch_scc1 2f\n\t"
                 "s_bitset1_b64 %[lits], %[s]\n\t"
                 "s_add_u32 %[s], %[s], %[st]\n\t"
                 "s_cmp_lt_u32 %[s], 64\n\t"
                 "s_cbranch_scc1 1b\n"
                 "2:"
                 : [s] "+s"(s), [lits] "+s"(lits), [st] "=&s"(st)
                 : [step] "v"(step)
                 : "scc");
}
// inclusive prefix sum over the 64 lanes: four row_shr steps inside the rows of 16, then the two
// row broadcasts (DPP, no LDS)
__device__ __forceinline__ uint32_t wave_inclusive_sum(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false); // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false); // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false); // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false); // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false); // row_bcast:15 -> rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false); // row_bcast:31 -> rows 2, 3
    return v;
}
// inclusive prefix maximum over the 64 lanes (same DPP steps; lanes without a source keep their value)
__device__ __forceinline__ uint32_t wave_inclusive_max(uint32_t v) {
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false));
    return v;
}
__device__ __forceinline__ uint64_t uni64(uint64_t v) {
    return (uint64_t)uni((uint32_t)(v >> 32)) << 32 | uni((uint32_t)v);
}
// ---- the compressed stream ---------------------------------------------------------------------
// Two 256-byte chunks of it sit in LDS, a third is in flight in a register (one dword per lane);
// the only decoder state is the bit position.  Lane j reads the 32 bits that start j bits further on
// with one ds_read2_b32 (all lanes hit the same two or three words: broadcast reads).
struct InStream {
    const uint32_t *src; // dword-aligned origin
    uint32_t *ring;      // Lds::in_ring
    uint32_t bitpos;     // uniform: bits consumed, counted from the origin
    uint32_t chunk;      // uniform: chunks `chunk` and `chunk + 1` are in the ring
    uint32_t pre;        // per lane: dword 64 * (chunk + 2) + lane

    __device__ __forceinline__ void init(const uint8_t *p, uint32_t *lds_ring) {
        const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(p) & 3u);
        src = reinterpret_cast<const uint32_t *>(p - mis);
        ring = lds_ring;
        seek(mis);
    }
    // continue at byte `b` (counted from the origin)
    __device__ __forceinline__ void seek(uint32_t b) {
        const uint32_t lane = threadIdx.x & 63u;
        bitpos = b * 8u;
        chunk = b >> 8;
        ring[(chunk & 1u) * 64u + lane] = src[chunk * 64u + lane];
        ring[((chunk + 1u) & 1u) * 64u + lane] = src[(chunk + 1u) * 64u + lane];
        pre = src[(chunk + 2u) * 64u + lane];
    }
    // bring the ring up to the bit position (call before reading; at most one chunk per call in the
    // symbol loops, which consume less than 256 bytes between calls)
    __device__ __forceinline__ void sync() {
        while ((bitpos >> 11) != chunk) {
            chunk += 1;
            ring[((chunk + 1u) & 1u) * 64u + (threadIdx.x & 63u)] = pre;
            pre = src[(chunk + 2u) * 64u + (threadIdx.x & 63u)];
        }
    }
    // 32 bits starting `lane` bits ahead of the bit position (lane < 64), per lane
    __device__ __forceinline__ uint32_t lane_bits32(uint32_t lane) const {
        const uint32_t t = bitpos + lane, dw = t >> 5;
        return __builtin_amdgcn_alignbit(ring[(dw + 1u) & 127u], ring[dw & 127u], t & 31u);
    }
    // the next 32 bits, uniform (one LDS round trip: the serial header code uses HeadBits instead)
    __device__ __forceinline__ uint32_t bits32() const { return uni(lane_bits32(0)); }
    __device__ __forceinline__ void consume(uint32_t n) { bitpos += n; }
    __device__ __forceinline__ uint64_t consumed_bits() const { return bitpos; }
};

// uniform bit buffer over an InStream for the serial parts (block headers, code lengths): one LDS
// round trip per ~33 bits instead of one per field
struct HeadBits {
    InStream &in;
    uint64_t buf = 0;
    uint32_t cnt = 0;
    __device__ __forceinline__ explicit HeadBits(InStream &i) : in(i) {}
    __device__ __forceinline__ void fill() {
        in.sync();
        const uint32_t dw = in.bitpos >> 5;
        const uint32_t lo = uni(in.ring[dw & 127u]), hi = uni(in.ring[(dw + 1u) & 127u]);
        buf = (((uint64_t)hi << 32) | lo) >> (in.bitpos & 31u);
        cnt = 64u - (in.bitpos & 31u);
    }
    __device__ __forceinline__ uint32_t peek(uint32_t n) { // n <= 32
        if (cnt < n) fill();
        return (uint32_t)buf & (uint32_t)((1ull << n) - 1ull);
    }
    __device__ __forceinline__ void drop(uint32_t n) {
        buf >>= n;
        cnt -= n;
        in.bitpos += n;
    }
    __device__ __forceinline__ uint32_t take(uint32_t n) {
        const uint32_t v = peek(n);
        drop(n);
        return v;
    }
};

// ---- canonical Huffman tables ----------------------------------------------------------------
// lens[0..n) -> primary table of TB bits + (cnt, syms) for codes longer than TB.  which: 0 = lit/len, 1 = dist/cl.
// kind_of: 0 literal/length alphabet, 1 distance alphabet, 2 code-length alphabet.
// Returns false for an over-subscribed set.
__device__ bool build_table(Lds &L, const uint8_t *lens, uint32_t n, uint32_t which, uint32_t TB, uint16_t *tab,
                            uint32_t alphabet, uint32_t lane) {
    if (lane < 16) L.cnt[which][lane] = 0;
    for (uint32_t i = lane; i < (1u << TB); i += 64)
        tab[i] = (uint16_t)(alphabet == 1 ? dist_special(DK_INVALID, 0) : lit_entry(0, LK_INVALID, 0));
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
            L.start[which][l] = (uint16_t)index;
            L.fcode[which][l] = (uint16_t)code;
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
            if (l == k) L.syms(which)[off[k] + __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)s;
            off[k] += __popcll(m);
        }
    }
    __syncthreads();
    // fill: every code of length <= TB owns 2^(TB-len) slots; longer codes mark their prefix slot
    for (uint32_t i = lane; i < index; i += 64) {
        const uint32_t s = L.syms(which)[i], l = lens[s];
        const uint32_t c = L.fcode[which][l] + (i - L.start[which][l]);
        const uint32_t rev = __brev(c) >> (32 - l);
        uint32_t e;
        if (alphabet == 0) {
            if (s < 256) e = lit_entry(s, LK_LIT, l);
            else if (s == 256) e = lit_entry(0, LK_EOB, l);
            else if (s < 286) e = len_entry(c_len_base[s - 257], c_len_extra[s - 257], l);
            else e = lit_entry(0, LK_INVALID, l);
        } else if (alphabet == 1) {
            e = s < 30 ? dist_entry(s, l) : dist_special(DK_INVALID, l);
        } else {
            e = lit_entry(s, LK_LIT, l);
        }
        if (l <= TB) {
            for (uint32_t k = rev; k < (1u << TB); k += 1u << l) tab[k] = (uint16_t)e;
        } else {
            tab[rev & ((1u << TB) - 1u)] = (uint16_t)(alphabet == 1 ? dist_special(DK_ESC, 0) : lit_entry(0, LK_ESC, 0));
        }
    }
    __syncthreads();
    return true;
}

// Decode one symbol whose code is longer than TB bits.  x = the stream bits at the symbol.  For each
// length l the first l bits, read as a number MSB first, are a code of that length iff they fall into
// [fcode[l], fcode[l] + cnt[l]) (canonical Huffman codes).  Returns the symbol or 0xFFFF; *bits = l.
__device__ uint32_t slow_symbol(const Lds &L, uint32_t which, uint32_t TB, uint32_t x, uint32_t *bits) {
    const uint32_t rev = __brev(x);
    for (uint32_t l = TB + 1; l < 16; l++) {
        const uint32_t code = rev >> (32 - l);
        const uint32_t f = uni(L.fcode[which][l]), c = uni(L.cnt[which][l]);
        if (code - f < c) {
            *bits = l;
            return uni(L.syms(which)[uni(L.start[which][l]) + (code - f)]);
        }
    }
    *bits = 15;
    return 0xFFFFu;
}

// full entry of a symbol whose primary entry says "long code"
// (out of line: rare, and bulky enough to slow the window loop down when inlined into it)
__device__ __noinline__ uint32_t resolve_long(const Lds &L, uint32_t alphabet, uint32_t x) {
    uint32_t bits;
    const uint32_t s = slow_symbol(L, alphabet, alphabet == 0 ? LB : DB, x, &bits);
    if (s == 0xFFFFu) return alphabet == 0 ? lit_entry(0, LK_INVALID, 15) : dist_special(DK_INVALID, 15);
    if (alphabet == 0) {
        if (s < 256) return lit_entry(s, LK_LIT, bits);
        if (s == 256) return lit_entry(0, LK_EOB, bits);
        if (s < 286) return len_entry(c_len_base[s - 257], c_len_extra[s - 257], bits);
        return lit_entry(0, LK_INVALID, bits);
    }
    return s < 30 ? dist_entry(s, bits) : dist_special(DK_INVALID, bits);
}

// ---- CRC32 (gzip): GF(2) helpers in the reflected representation -----------------------------
constexpr uint32_t CRC_POLY = 0xEDB88320u;
// Slicing tables: t[k][b] = register after byte b followed by k zero bytes.  Sixteen bytes are then folded with sixteen
// INDEPENDENT lookups (one LDS round trip) where the byte-wise table needs sixteen dependent ones: k_bgzf_crc was bound
// by exactly that latency (0.98 ms per 520 MB; 16 KiB of tables per workgroup instead of 1 KiB).
constexpr uint32_t CRC_SLICES = 16;
struct CrcTable {
    uint32_t t[CRC_SLICES][256];
    constexpr CrcTable() : t() {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i;
            for (int k = 0; k < 8; k++) c = (c & 1u) ? (c >> 1) ^ CRC_POLY : c >> 1;
            t[0][i] = c;
        }
        for (uint32_t k = 1; k < CRC_SLICES; k++)
            for (uint32_t i = 0; i < 256; i++) t[k][i] = (t[k - 1][i] >> 8) ^ t[0][t[k - 1][i] & 0xFFu];
    }
};
// (copied to LDS by k_bgzf_crc)
__constant__ CrcTable c_crc;
__device__ __forceinline__ uint32_t ld32u(const uint8_t *p) {
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}
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

// one BGZF block, by one wavefront
__device__ __forceinline__ void inflate_block(Lds &L, const uint8_t *__restrict__ comp, const BgzfBlock *__restrict__ blocks,
                                              uint32_t bi, uint8_t *__restrict__ out, uint32_t *__restrict__ status, uint32_t lane) {
    // the descriptor, forced uniform: everything derived from it (the whole bit stream state) stays in SGPRs
    BgzfBlock blk = blocks[bi];
    blk.in_off = uni64(blk.in_off);
    blk.out_off = uni64(blk.out_off);
    const uint32_t isize = uni(blk.isize), in_len = uni(blk.in_len);
    if (isize == 0 && in_len == 0) {
        if (lane == 0) status[bi] = INF_OK;
        return;
    }
    InStream br;
    br.init(comp + blk.in_off, L.in_ring);
    const uint32_t in_mis = (uint32_t)(reinterpret_cast<uintptr_t>(comp + blk.in_off) & 3u);
    const uint64_t bit_limit = (uint64_t)(in_mis + in_len) * 8u;
    uint8_t *const gdst = out + blk.out_off;

    // output state: the last RING bytes of [0, pos) are in the ring, [0, flushed) have left for HBM
    uint32_t pos = 0, flushed = 0, err = INF_OK;

    // ring bytes [flushed, flushed + n) -> HBM
    auto flush_piece = [&](uint32_t n) {
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
        // Far matches read these bytes back, much later (a byte leaves the ring 3 pieces after it was flushed) and
        // from this same wave: its stores and its L1-bypassing loads (sc1: served by this XCD's L2) take the same
        // path in order, so no fence is needed -- an agent-scope release here cost a write-back of the XCD's L2
        // (buffer_wbl2) and a drain of the wave's stores per KiB of output.
#ifdef NGSQ_INFLATE_FENCE
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
#endif
    };

    PROF_DECL;
    // ---- the symbol queue and the emit
    // [0, pos) has been written (ring / HBM); the queued symbols cover [pos, dpos) without gaps, in order
    uint32_t dpos = 0, qhead = 0, qn = 0;
    // Write the next n <= 64 bytes: every output byte is produced by one lane, no loop over the symbols (a serial copy
    // per match cost ~35 scalar instructions each, and the scalar unit -- one per CU -- is what bounds this kernel):
    //   (1) the symbol that owns each byte: the queued symbols mark the byte they start at, a max-scan spreads the marks
    //       (byte 0 may belong to the queue's first symbol, begun in an earlier chunk);
    //   (2) the owner's literal byte or distance is read from the queue; a match byte's source is T - distance;
    //   (3) a source inside this chunk is a pointer to another lane: pointer jumping (p = p[p], at most six
    //       rounds, none for the usual match that reaches behind the chunk) until every pointer ends at a byte
    //       whose value is known -- a literal, a byte already in the ring, or one that has left the ring and is
    //       read back from HBM (all such bytes of a chunk in one round trip);
    //   (4) the values travel back along the pointers and the chunk is stored.
    // A source byte S is still in the ring iff S >= base - RING (base = first byte of the chunk: everything in
    // front of it has been written); older bytes have been flushed, because base - flushed < PIECE at every
    // chunk start and PIECE + 64 <= RING.
    auto emit_chunk = [&](uint32_t n) {
        const uint32_t base = pos;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); // queue entries are read by other lanes than wrote them
        const uint32_t qe = L.e.q[(qhead + lane) & (QCAP - 1u)];
        const uint32_t st = qe & 0xFFFFu;
        const bool queued = lane < qn;
        // (the marks are read by OTHER lanes than wrote them: the wavefront-scope fence makes the compiler reload
        // instead of forwarding this lane's own zero; LDS operations of one wave execute in order, nothing else is
        // needed.  A `volatile` pointer did that too, but it lost the LDS address space: flat_store_byte /
        // flat_load_ubyte with a full s_waitcnt after each, three round trips per 64 bytes of output.)
        uint8_t *const mark = L.e.mark;
        mark[lane] = 0;
        if (queued && st - base < 64u) mark[st - base] = (uint8_t)(lane + 1u);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        uint32_t own = mark[lane];
        if (lane == 0 && own == 0) own = 1u;
        own = wave_inclusive_max(own);
        const uint32_t oe = L.e.q[(qhead + own - 1u) & (QCAP - 1u)];
        const bool active = lane < n;
        const bool lit = (oe & 0x10000u) != 0;
        const uint32_t T = base + lane, src = T - (oe >> 17) - 1u;
        const bool inchunk = active && !lit && (int32_t)(src - base) >= 0;
        uint32_t val = (oe >> 17) & 255u;
        if (active && !lit && !inchunk) {
            if (__builtin_expect((int32_t)(src - base + RING) < 0, 0))
                val = __hip_atomic_load(gdst + src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else
                val = L.ring[src & RMASK];
        }
        if (__ballot(inchunk)) {
            uint32_t p = inchunk ? src - base : lane;
            for (;;) {
                const uint32_t q = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(p << 2), (int)p);
                if (!__ballot(q != p)) break;
                p = q;
            }
            val = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(p << 2), (int)val);
        }
        if (active) L.ring[T & RMASK] = (uint8_t)val;
        pos += n;
        while (pos - flushed >= PIECE) flush_piece(PIECE);
        // symbols that are written out completely leave the queue: all that start in front of the new position, but
        // the last of them if it reaches beyond it (a symbol ends where the next one starts)
        const uint32_t k = (uint32_t)__popcll(__ballot(queued && st < pos));
        const uint32_t nxt = k < qn ? (uint32_t)__builtin_amdgcn_readlane((int)st, (int)k) : dpos;
        const uint32_t drop = k - (nxt > pos ? 1u : 0u);
        qhead = (qhead + drop) & (QCAP - 1u);
        qn -= drop;
        PROF_COUNT(2, 1);
    };
    // whole 64-byte chunks while there are any; all = everything decoded so far (before the tables or the ring change hands)
    auto drain = [&](bool all) {
        while (dpos - pos >= 64u) emit_chunk(64u);
        if (all && dpos != pos) emit_chunk(dpos - pos);
    };

    bool last = false;
    while (!last && err == INF_OK) {
        PROF(0); // other
        HeadBits hb(br);
        last = hb.take(1);
        const uint32_t type = hb.take(2);
        if (br.consumed_bits() > bit_limit) {
            err = INF_INPUT_OVERRUN;
            break;
        }
        if (type == 0) {
            // stored: skip to the byte boundary, LEN, NLEN, then LEN raw bytes
            hb.take((8u - (br.bitpos & 7u)) & 7u);
            const uint32_t len = hb.take(16);
            const uint32_t nlen = hb.take(16);
            if ((len ^ nlen) != 0xFFFFu) {
                err = INF_BAD_STORED_LEN;
                break;
            }
            if (pos + len > isize) {
                err = INF_OUTPUT_OVERRUN;
                break;
            }
            // the stream is at a byte boundary now: copy the source bytes directly, then restart behind them
            const uint32_t byte0 = (uint32_t)(br.consumed_bits() >> 3);
            const uint8_t *sp = comp + blk.in_off - in_mis + byte0;
            for (uint32_t done = 0; done < len;) {
                const uint32_t n = min(len - done, PIECE - (pos - flushed));
                for (uint32_t i = lane; i < n; i += 64) L.ring[(pos + i) & RMASK] = sp[done + i];
                pos += n;
                done += n;
                if (pos - flushed >= PIECE) flush_piece(PIECE);
            }
            dpos = pos;
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
            hlit = hb.take(5) + 257;
            hdist = hb.take(5) + 1;
            const uint32_t hclen = hb.take(4) + 4;
            if (hlit > 286 || hdist > 30) {
                err = INF_BAD_CODE_LENGTHS;
                break;
            }
            if (lane < 19) L.lens[lane] = 0;
            __syncthreads();
            for (uint32_t i = 0; i < hclen; i++) {
                const uint32_t v = hb.take(3);
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
                const uint32_t x = hb.peek(14); // a code (<= 7 bits) and its repeat count (<= 7 bits)
                const uint32_t e = uni((uint32_t)L.dist_tab[x & 127u]);
                if (!e_is_lit(e)) { // the code-length code has at most 7 bits: no long codes
                    err = INF_BAD_CODE_LENGTHS;
                    break;
                }
                const uint32_t nb = e_bits(e), s = e_byte(e);
                if (s < 16) {
                    hb.drop(nb);
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
                    rep = 3 + ((x >> nb) & 3u);
                    hb.drop(nb + 2);
                    val = prev;
                } else if (s == 17) {
                    rep = 3 + ((x >> nb) & 7u);
                    hb.drop(nb + 3);
                    val = 0;
                    prev = 0;
                } else {
                    rep = 11 + ((x >> nb) & 127u);
                    hb.drop(nb + 7);
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
        PROF(1); // block header + tables
        // ---- the symbol windows
        bool end_of_block = false;
        while (!end_of_block && err == INF_OK) {
            PROF(6); // tail of the previous window (loop)
            drain(false); // (also makes room in the queue: fewer than 64 bytes pending = fewer than 64 symbols)
            PROF(5); // emit
            // Lane j decodes what would start j bits from here: a literal, or a whole match -- length code,
            // its extra bits, and (from the lane at that bit offset) the distance code and its extra bits.
            br.sync();
            const uint32_t x = br.lane_bits32(lane);
            uint32_t E = L.lit_tab[x & ((1u << LB) - 1u)];
            const uint32_t nb = e_bits(E), lex = e_len_extra(E);
            const uint32_t len = e_byte(E) + 3u + ((x >> nb) & ((1u << lex) - 1u));
            const uint32_t s2 = lane + nb + lex; // bit offset of the distance code
            const uint32_t xd = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(s2 << 2), (int)x);
            const uint32_t D = L.dist_tab[xd & ((1u << DB) - 1u)];
            const uint32_t db = e_bits(D), dex = d_extra(D);
            const uint32_t dist = d_base(D) + ((xd >> db) & ((1u << dex) - 1u));
            bool is_lit = e_is_lit(E);
            const bool is_match = e_is_len(E) && s2 < 64 && d_kind(D) == DK_BASE;
            // bits to the next symbol, or a stop mark: end of block, long codes, a distance code beyond lane 63
            uint32_t step = is_lit ? nb : is_match ? nb + lex + db + dex : 0x80u;
            PROF(2); // window bits + gathers
            PROF_COUNT(0, 1);
            // the chain of real symbol starts: readlane + add per symbol
            uint64_t syms = 0;
            uint32_t s = 0;
            for (;;) {
                chain_literals(step, s, syms);
                if (s >= 64) break;
                // a literal with a long code is patched into its lane and the chain goes on
                uint32_t e = __builtin_amdgcn_readlane(E, s);
                if (e_is_len(e) || e_kind(e) != LK_ESC) break;
                e = uni(resolve_long(L, 0, __builtin_amdgcn_readlane(x, s)));
                PROF_COUNT(5, 1);
                if (!e_is_lit(e)) break;
                if (lane == s) {
                    E = e;
                    step = e_bits(e);
                    is_lit = true;
                }
            }
            // the queue takes QCAP - qn more symbols: a window with more (short codes, a nearly full queue) ends early, the
            // next one starts at the first symbol left out
            bool trimmed = false;
            if (__builtin_expect((uint32_t)__popcll(syms) > QCAP - qn, 0)) {
                uint64_t m = syms;
                for (uint32_t k = QCAP - qn; k; k--) m &= m - 1; // drop the symbols that fit
                s = (uint32_t)__builtin_ctzll(m);
                syms &= (1ull << s) - 1ull;
                trimmed = true;
            }
            PROF(3); // chain
            // where each symbol on the chain writes: prefix sum of the output lengths
            const bool on_chain = (syms >> lane) & 1ull;
            const uint32_t olen = !on_chain ? 0u : is_lit ? 1u : len;
            const uint32_t incl = wave_inclusive_sum(olen);
            const uint32_t at = dpos + incl - olen;
            const uint32_t total = __builtin_amdgcn_readlane(incl, 63);
            PROF_COUNT(1, __popcll(__ballot(on_chain && is_lit)));
            // a corrupt stream stops here, before anything of this window is queued: a distance beyond the
            // start of the output would read in front of the block's buffer, and output beyond ISIZE would be flushed
            // past its end
            if (__ballot(on_chain && !is_lit && dist > at)) {
                err = INF_BAD_DISTANCE;
                break;
            }
            if (dpos + total > isize) {
                err = INF_OUTPUT_OVERRUN;
                break;
            }
            if (on_chain) {
                const uint32_t rank = (uint32_t)__popcll(syms & ((1ull << lane) - 1ull));
                L.e.q[(qhead + qn + rank) & (QCAP - 1u)] = is_lit ? q_lit(at, e_byte(E)) : q_match(at, dist);
            }
            qn += (uint32_t)__popcll(syms);
            dpos += total;
            br.consume(s);
            PROF(4); // queue
            // A chain that stopped at a length code whose distance code lies beyond lane 63 just ends the
            // window there: the next window starts at that length code and sees all of the match.
            const bool resume = s > 0 && s < 64 && e_is_len(__builtin_amdgcn_readlane(E, s & 63u));
            if (s < 64 && !resume && !trimmed) {
                // The chain stopped at a symbol the lanes could not finish: end of block, a long code (or a
                // distance code with one), or an invalid code.  One symbol the plain way, through the queue like the others.
                PROF_COUNT(6, 1);
                drain(false);
                br.sync();
                const uint32_t x0 = br.bits32();
                uint32_t e = uni((uint32_t)L.lit_tab[x0 & ((1u << LB) - 1u)]);
                if (!e_is_len(e) && e_kind(e) == LK_ESC) e = uni(resolve_long(L, 0, x0));
                const uint32_t eb = e_bits(e);
                if (e_is_lit(e)) {
                    if (dpos + 1 > isize) err = INF_OUTPUT_OVERRUN;
                    else {
                        if (lane == 0) L.e.q[(qhead + qn) & (QCAP - 1u)] = q_lit(dpos, e_byte(e));
                        qn += 1;
                        dpos += 1;
                        br.consume(eb);
                    }
                } else if (!e_is_len(e) && e_kind(e) == LK_EOB) {
                    br.consume(eb);
                    end_of_block = true;
                } else if (e_is_len(e)) {
                    const uint32_t ex = e_len_extra(e);
                    const uint32_t l = e_byte(e) + 3u + ((x0 >> eb) & ((1u << ex) - 1u));
                    br.consume(eb + ex);
                    br.sync();
                    const uint32_t x1 = br.bits32();
                    uint32_t d = uni((uint32_t)L.dist_tab[x1 & ((1u << DB) - 1u)]);
                    if (d_kind(d) == DK_ESC) d = uni(resolve_long(L, 1, x1));
                    const uint32_t b2 = e_bits(d), ex2 = d_extra(d);
                    const uint32_t dd0 = d_base(d) + ((x1 >> b2) & ((1u << ex2) - 1u));
                    br.consume(b2 + ex2);
                    if (d_kind(d) != DK_BASE) err = INF_BAD_SYMBOL;
                    else if (dd0 > dpos) err = INF_BAD_DISTANCE;
                    else if (dpos + l > isize) err = INF_OUTPUT_OVERRUN;
                    else {
                        if (lane == 0) L.e.q[(qhead + qn) & (QCAP - 1u)] = q_match(dpos, dd0);
                        qn += 1;
                        dpos += l;
                    }
                } else {
                    err = INF_BAD_SYMBOL;
                }
            }
        }
        // the tables (and the code-length scratch the queue shares its memory with) are about to change
        if (err == INF_OK) drain(true);
    }
    PROF(0);
    if (err == INF_OK && pos != isize) err = pos > isize ? INF_OUTPUT_OVERRUN : INF_SIZE_MISMATCH;
    if (err == INF_OK && br.consumed_bits() > bit_limit) err = INF_INPUT_OVERRUN;
    __syncthreads();
    if (err == INF_OK) {
        while (flushed < pos) flush_piece(min(pos - flushed, PIECE));
    }
    PROF(7); // final flush + CRC
    PROF_FLUSH;
    if (lane == 0) status[bi] = err;
}

// The decoders are PERSISTENT: the grid is what the device holds at once (per_cu wavefronts on every CU, LDS-limited)
// and each wavefront takes block after block from a counter.  One launch per chunk either way, but a grid of 8-16 k
// one-block workgroups kept the dispatcher busy placing them for the whole 7 ms, and while a dispatch still has
// workgroups waiting for a slot the workgroups of OTHER queues are not placed at all -- the record index and the
// column kernels of the previous chunk (another stream) ran only when the inflate had finished.
// A grid that is resident from the start leaves the dispatcher free, and the kernels of the other stream take the
// wave slots and registers the decoders leave.  The counter also evens out the tail: no last partial round of blocks.
__global__ __launch_bounds__(64) void k_bgzf_inflate(const uint8_t *__restrict__ comp,
                                                     const BgzfBlock *__restrict__ blocks, uint32_t n_blocks,
                                                     uint8_t *__restrict__ out, uint32_t *__restrict__ status,
                                                     uint32_t *__restrict__ next_block) {
    extern __shared__ __align__(16) uint8_t s_raw[];
    Lds &L = *reinterpret_cast<Lds *>(s_raw);
    const uint32_t lane = threadIdx.x;
    for (;;) {
        uint32_t bi = 0;
        if (lane == 0) bi = atomicAdd(next_block, 1u);
        bi = uni(bi);
        if (bi >= n_blocks) return;
        inflate_block(L, comp, blocks, bi, out, status, lane);
        __syncthreads(); // the next block's first LDS writes come after this block's last reads
    }
}

// CRC32 of every inflated block against its gzip trailer: its own kernel (one wave per block, no LDS
// ring, so many waves per CU) instead of a tax on the four decoders of a CU.  The block is cut into 64
// equal slices, right-aligned (the CRC register is linear in the message once the initial value is
// accounted for, and leading zero bytes leave a zero register at zero): the lane that holds byte 0
// starts from 0xFFFFFFFF, the others from 0, and the slices are combined pairwise in six steps,
// register(A || B) = register(A) * x^(8 |B|) + register(B), with |B| the same for every pair of a step.
constexpr uint32_t CRC_WAVES = 8; // BGZF blocks per workgroup: the tables are loaded once for all of them
__global__ __launch_bounds__(64 * CRC_WAVES) void k_bgzf_crc(const uint8_t *__restrict__ out, const BgzfBlock *__restrict__ blocks,
                                                             uint32_t n_blocks, uint32_t *__restrict__ status) {
    NGSQ_FOREGROUND_WAVE();
    __shared__ uint32_t s_tab[CRC_SLICES * 256];
    for (uint32_t k = threadIdx.x; k < CRC_SLICES * 256; k += 64 * CRC_WAVES) s_tab[k] = c_crc.t[k >> 8][k & 0xFFu];
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t bi = blockIdx.x * CRC_WAVES + (threadIdx.x >> 6);
    if (bi >= n_blocks) return;
    const uint32_t isize = uni(blocks[bi].isize);
    if (uni(status[bi]) != INF_OK) return;
    const uint8_t *p = out + uni64(blocks[bi].out_off);
    const uint32_t S = (isize + 63u) / 64u, pad = 64u * S - isize;
    // real bytes of this lane's slice
    const uint32_t v0 = lane * S, v1 = v0 + S;
    const uint32_t a = v0 > pad ? v0 - pad : 0u, b = v1 > pad ? v1 - pad : 0u;
    uint32_t c = (a == 0 && b > 0) ? 0xFFFFFFFFu : 0u;
    // 64 bytes per step into registers: every lane walks its own slice, so with narrower loads a cache line
    // would be fetched again for each of them (the 64 lanes' lines of a step do not fit the vector cache)
    uint32_t i = a;
    for (; i + 64 <= b; i += 64) {
        uint32_t w[16];
        __builtin_memcpy(w, p + i, 64);
#pragma unroll
        for (int k = 0; k < 16; k += 4) {
            const uint32_t x0 = c ^ w[k], x1 = w[k + 1], x2 = w[k + 2], x3 = w[k + 3];
            c = s_tab[15 * 256 + (x0 & 0xFFu)] ^ s_tab[14 * 256 + ((x0 >> 8) & 0xFFu)] ^ s_tab[13 * 256 + ((x0 >> 16) & 0xFFu)] ^
                s_tab[12 * 256 + (x0 >> 24)] ^ s_tab[11 * 256 + (x1 & 0xFFu)] ^ s_tab[10 * 256 + ((x1 >> 8) & 0xFFu)] ^
                s_tab[9 * 256 + ((x1 >> 16) & 0xFFu)] ^ s_tab[8 * 256 + (x1 >> 24)] ^ s_tab[7 * 256 + (x2 & 0xFFu)] ^
                s_tab[6 * 256 + ((x2 >> 8) & 0xFFu)] ^ s_tab[5 * 256 + ((x2 >> 16) & 0xFFu)] ^ s_tab[4 * 256 + (x2 >> 24)] ^
                s_tab[3 * 256 + (x3 & 0xFFu)] ^ s_tab[2 * 256 + ((x3 >> 8) & 0xFFu)] ^ s_tab[1 * 256 + ((x3 >> 16) & 0xFFu)] ^
                s_tab[x3 >> 24];
        }
    }
    for (; i + 4 <= b; i += 4) {
        const uint32_t x = c ^ ld32u(p + i);
        c = s_tab[3 * 256 + (x & 0xFFu)] ^ s_tab[2 * 256 + ((x >> 8) & 0xFFu)] ^ s_tab[1 * 256 + ((x >> 16) & 0xFFu)] ^ s_tab[x >> 24];
    }
    for (; i < b; i++) c = s_tab[(c ^ p[i]) & 0xFFu] ^ (c >> 8);
    // pairwise combination: after step j the lanes whose low j+1 bits are ones hold 2^(j+1) slices
    uint32_t mult = crc_xpow8(S); // x^(8 |B|) of this step (uniform)
    for (uint32_t j = 0; j < 6; j++) {
        const uint32_t left = (uint32_t)__shfl_up((int)c, 1u << j, 64);
        if ((lane & ((2u << j) - 1u)) == (2u << j) - 1u) c = crc_mul(left, mult) ^ c;
        mult = crc_mul(mult, mult);
    }
    const uint32_t crc = ~__builtin_amdgcn_readlane(c, 63);
    if (isize && lane == 0 && crc != blocks[bi].crc) status[bi] = INF_CRC_MISMATCH;
    if (!isize && lane == 0 && blocks[bi].crc != 0) status[bi] = INF_CRC_MISMATCH;
}

hipError_t launch_bgzf_crc(const BgzfBlock *blocks, uint32_t n_blocks, const uint8_t *out, uint32_t *status, hipStream_t s) {
    if (!n_blocks) return hipSuccess;
    hipLaunchKernelGGL(k_bgzf_crc, dim3((n_blocks + CRC_WAVES - 1) / CRC_WAVES), dim3(64 * CRC_WAVES), 0, s, out, blocks, n_blocks, status);
    return hipGetLastError();
}

hipError_t launch_bgzf_inflate(const uint8_t *comp, const BgzfBlock *blocks, uint32_t n_blocks, uint8_t *out,
                               uint32_t *status, uint32_t *counter, bool check_crc, hipStream_t s) {
    if (!n_blocks) return hipSuccess;
    static bool attr = false;
    static uint32_t resident = 0;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_bgzf_inflate),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(Lds));
        if (e != hipSuccess) return e;
        int dev = 0, n_cu = 256;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
        // decoders per CU: what the LDS holds (160 KiB / 6400-byte allocations = 25), less one so that every SIMD keeps
        // registers and a wave slot for the other stream's kernels (NGSQ_INFLATE_PER_CU: measurement aid)
        uint32_t per_cu = 24;
        if (const char *v = getenv("NGSQ_INFLATE_PER_CU")) per_cu = (uint32_t)atoi(v); // 0: a workgroup per block, as before
        resident = per_cu ? (uint32_t)n_cu * per_cu : 0xFFFFFFFFu;
        attr = true;
    }
    hipError_t e = hipMemsetAsync(counter, 0, sizeof(uint32_t), s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_bgzf_inflate, dim3(n_blocks < resident ? n_blocks : resident), dim3(64), sizeof(Lds), s, comp, blocks, n_blocks, out,
                       status, counter);
    if (check_crc) (void)launch_bgzf_crc(blocks, n_blocks, out, status, s);
#ifdef NGSQ_INFLATE_PROFILE
    {
        unsigned long long h[16];
        (void)hipStreamSynchronize(s);
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_inflate_prof), sizeof h);
        static const char *names[8] = {"other", "header+tables", "window+gathers", "chain", "queue", "emit",
                                       "window tail/piece flush", "final flush+crc"};
        unsigned long long tot = 0;
        for (int k = 0; k < 8; k++) tot += h[k];
        for (int k = 0; k < 8; k++)
            fprintf(stderr, "[inflate-prof] %-24s %6.2f %%\n", names[k], tot ? 100.0 * (double)h[k] / (double)tot : 0.0);
        fprintf(stderr, "[inflate-prof] windows %llu, literals %llu, matches %llu (bytes %llu; distance > 4 KiB %llu, > 16 KiB %llu), "
                        "long codes %llu, symbols taken the plain way %llu\n",
                h[8], h[9], h[10], h[11], h[12], h[15], h[13], h[14]);
        unsigned long long z[16] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_inflate_prof), z, sizeof z);
    }
#endif
    return hipGetLastError();
}

} // namespace ngsq
