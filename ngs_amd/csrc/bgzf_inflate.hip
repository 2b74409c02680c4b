// bgzf_inflate.hip -- DEFLATE (RFC 1951) decoder for BGZF blocks on gfx950.
//
// Replaces, for the device ingest path, the inflate the reference gets from noodles-bgzf 0.20
// (flate2 1.0.24 / miniz_oxide 0.5.4) under `reader.records(&header)`, src/qc/command.rs:305 and
// src/utils/formats/bam.rs:32-56.  Written from RFC 1951 and the SAM/BAM specification 4.1.
//
// One WAVEFRONT per BGZF block (blocks are independent gzip members of <= 64 KiB); a CU holds 24 of them and is bound by
// the instructions it can issue (one scalar and one vector instruction per cycle), so the design minimises instructions
// per symbol and puts all 64 lanes to work on the one bit stream, in two phases:
//  * SYMBOL WINDOWS find where the symbols start.  Lane j looks up the Huffman codes that WOULD start at bits j and
//    j + 64 of the next 128 bits (literal/length table, then the distance table with the stream bits behind the length
//    code and its extra bits, every lane at once; the low bits of a table entry are the number of bits the symbol takes).
//    The real symbol starts are the chain 0 -> step[0] -> step[0] + step[step[0]] ..., followed with one v_readlane +
//    add per symbol on the scalar unit, literals and whole matches alike; the lanes on the chain append their bit offset
//    to a list.
//  * BATCHES of up to 64 listed symbols are decoded one per lane -- every lane a real symbol -- and one prefix sum of
//    their output lengths gives the offsets.  The decoded symbols wait in a register, one per lane, and leave 64 bytes
//    of output at a time: every output byte is produced by one lane (the owner of each byte by a max-scan, match
//    sources inside the same 64 bytes by pointer jumping).
//  * Huffman tables are built by the 64 lanes (ballot-ranked canonical codes, parallel fill); literal/length codes longer
//    than the 10-bit first level have second-level tables, so no code is ever resolved serially.
//  * The compressed stream is staged through a small LDS ring; the only reader state is the bit position.  The last 2 KiB
//    of output live in an LDS ring; finished 512-byte pieces leave it as coalesced dword stores, and a match that reaches
//    further back than the ring reads its source from HBM (the bytes this wave wrote earlier).
//    6.4 KB of LDS and 80 registers per decoder: 24 decoders per CU.
//  * CRC32 of the block (gzip trailer) is verified on request by a second, wide kernel
//    (k_bgzf_crc: 64 slices per block, combined with GF(2) polynomial multiplication).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <vector>

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

// Table entries are 16 bits (LDS per decoder is what limits the decoders per CU).  The low seven bits of every entry are
// what the symbol windows need -- how far the symbol reaches, or 64 = "the chain stops here" (bit 6) -- so that a window's
// lanes get from two table reads to their step with an AND and an ADD each:
//   literal/length table:  literal          [15] 0  [14:7] the byte         [6:0] code bits
//                          length code      [15] 1  [14:7] base length - 3  [6:0] code bits + extra bits
//                                           (the extra bits follow from the base: 0 below 11 and for 258, else log2(base - 3) - 2)
//                          end of block, invalid code
//                                           [15] 0  [14:11] code bits  [8:7] which (LK_EOB / LK_INVALID)  [6:0] 64
//                          LK_ESC, first level only: a prefix of codes longer than LB bits
//                                           [15] 0  [14:7] first entry of its second-level table in Lds::sub / 2
//                                                   [6:5] 11  [4:2] index bits of that table
//   distance table:        [14:13] kind (BASE / ESC / INVALID)  [12:11] m  [10:7] extra bits (0..13)
//                          [6:0] code bits + extra bits, or 64;
//                          base distance = 1 + (m << extra bits)  (codes 0, 1: m = 0, 1; code c >= 2: m = 2 + (c & 1))
// The code-length code of a dynamic header uses the literal format (symbol in the byte field).
constexpr uint32_t LK_LIT = 0, LK_EOB = 1, LK_INVALID = 2;
constexpr uint32_t SUB_CAP = 340;
constexpr uint32_t DK_BASE = 0, DK_ESC = 1, DK_INVALID = 2;
constexpr uint32_t STOP = 0x40u; // bit 6 of a step
__device__ __forceinline__ constexpr uint32_t lit_entry(uint32_t byte, uint32_t kind, uint32_t bits) {
    return kind == LK_LIT ? byte << 7 | bits : bits << 11 | kind << 7 | STOP;
}
__device__ __forceinline__ constexpr uint32_t len_entry(uint32_t base, uint32_t extra, uint32_t bits) {
    return 0x8000u | (base - 3u) << 7 | (bits + extra);
}
__device__ __forceinline__ constexpr uint32_t esc_entry(uint32_t first, uint32_t index_bits) { return (first >> 1) << 7 | 0x60u | index_bits << 2; }
__device__ __forceinline__ constexpr uint32_t dist_entry(uint32_t code, uint32_t bits) {
    return code < 2 ? (DK_BASE << 13 | code << 11 | bits) : (DK_BASE << 13 | (2u + (code & 1u)) << 11 | ((code >> 1) - 1u) << 7 | (bits + (code >> 1) - 1u));
}
__device__ __forceinline__ constexpr uint32_t dist_special(uint32_t kind) { return kind << 13 | STOP; }
// fields
__device__ __forceinline__ bool e_is_len(uint32_t e) { return (e & 0x8000u) != 0; }
__device__ __forceinline__ bool e_is_lit(uint32_t e) { return (e & (0x8000u | STOP)) == 0; }
__device__ __forceinline__ bool e_is_esc(uint32_t e) { return (e & 0x8060u) == 0x60u; }
__device__ __forceinline__ uint32_t e_stop_kind(uint32_t e) { return (e >> 7) & 3u; } // of an entry that is neither literal nor length
__device__ __forceinline__ uint32_t e_step(uint32_t e) { return e & 127u; }
__device__ __forceinline__ uint32_t e_byte(uint32_t e) { return (e >> 7) & 255u; }
__device__ __forceinline__ uint32_t e_len_extra(uint32_t e) { // of a length entry
    const uint32_t b3 = e_byte(e);
    return b3 < 8u || b3 == 255u ? 0u : 29u - (uint32_t)__clz((int)b3);
}
// code bits of any final entry
__device__ __forceinline__ uint32_t e_bits(uint32_t e) {
    return e_is_len(e) ? e_step(e) - e_len_extra(e) : (e & STOP) ? (e >> 11) & 15u : e_step(e);
}
__device__ __forceinline__ uint32_t d_kind(uint32_t d) { return (d >> 13) & 3u; }
__device__ __forceinline__ uint32_t d_step(uint32_t d) { return d & 127u; }
__device__ __forceinline__ uint32_t d_extra(uint32_t d) { return (d >> 7) & 15u; }
__device__ __forceinline__ uint32_t d_base(uint32_t d) { return 1u + (((d >> 11) & 3u) << d_extra(d)); }

__constant__ uint16_t c_len_base[31] = {3,  4,  5,  6,  7,  8,  9,  10, 11,  13,  15,  17,  19,  23, 27, 31,
                                        35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258, 0,  0};
__constant__ uint8_t c_len_extra[31] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2,
                                        3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0, 0, 0};
__constant__ uint8_t c_cl_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// Decoded symbols wait (one per lane, in a register) until 64 bytes of output can be produced at once: emitting them window by
// window ran the 64-lane emit at a quarter of its width.  Entry: [15:0] output offset of the symbol's first byte
// (ISIZE <= 65536), [16] literal, [31:17] the literal byte or distance - 1.  QCAP = symbols per batch = lanes.
constexpr uint32_t QCAP = 64;
constexpr uint32_t WH = 2;                       // bit positions per lane and window: lane j looks at bits j, j + 64, ...
// A batch's windows start at most this many bits behind the bit position: the stream ring holds 4096 bits from the start of
// the chunk the bit position is in (2047 at worst), a window and its second phase read 64 WH + 20 + 32 bits from their start.
constexpr uint32_t REL_LIMIT = 4096u - 2047u - 64u * WH - 20u - 32u - 64u;
__device__ __forceinline__ uint32_t q_lit(uint32_t at, uint32_t byte) { return at | 0x10000u | byte << 17; }
__device__ __forceinline__ uint32_t q_match(uint32_t at, uint32_t dist) { return at | (dist - 1u) << 17; }

struct Lds {
    // two 256-byte chunks of the compressed stream (chunk c in slot c & 1); [128..130] repeat [0..2].  First member: its
    // dwords are read in pairs at computed addresses, and from LDS address 0 the pair's place goes into the instruction.
    uint32_t in_ring[128 + 3];
    uint8_t ring[RING];
    uint16_t lit_tab[1u << LB];
    uint16_t dist_tab[1u << DB]; // also the code-length-code table while a dynamic header is read
    uint32_t cnt[2][16];   // codes per length: [0] literal/length, [1] distance (or code-length code)
    uint16_t start1[16];   // distance / code-length alphabet: first index in syms1 of each length,
    uint16_t fcode1[16];   //   first canonical code of each length,
    uint16_t syms1[32];    //   symbols in canonical order (a code longer than the table is looked up there)
    // literal/length codes longer than LB bits: second-level tables, one per LB-bit prefix that has such codes, indexed by
    // the following bits (as many as the prefix's longest code needs).  A complete code over 286 symbols of at most 15 bits
    // needs at most 308 entries behind a 10-bit first level (zlib's `enough 286 10 15` = 1332 for both levels).
    uint16_t sub[SUB_CAP];
    union {
        uint8_t lens[320]; // code lengths while a block's tables are built
        struct {
            uint8_t mark[64]; // emit: the queued symbol that starts at each byte of a 64-byte output chunk
            uint32_t q[QCAP]; // a batch's symbol starts: bit offsets from the bit position
        } e;
    };
};
static_assert(sizeof(Lds) <= 6400, "five 1280-byte LDS granules per decoder: 25 decoders per CU");

__device__ __forceinline__ uint32_t uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
// The serial core of the decoder, FOUR instructions per symbol (three scalar, one v_readlane): starting at bit s < 64 of a
// 64-bit part of the window, mark the symbol start in `lits` (literals and whole matches alike) and step to the next one
// (step[s] bits further) until the part ends.  The position is kept biased by 2^32 - 64: v_readlane and s_bitset1_b64 take
// its low six bits, and the add's carry says "past bit 63" -- no compare.  A lane whose symbol cannot be stepped over (end
// of block, long distance code, invalid code) has bit 6 set in its step: that also carries, and the loop needs no test for
// it either; the last step read tells afterwards which it was (then the mark and the step are taken back).  Returns true
// if the chain stopped at such a symbol; s = its position, or (not stopped) the position in the NEXT part where the chain
// goes on.  Hand-written: the compiler's structurised control flow needs about four times as many instructions.
__device__ __forceinline__ bool chain_literals(uint32_t step, uint32_t &s, uint64_t &lits) {
    uint32_t st, stopped, sb = s - 64u;
    asm volatile("1:\n\t"
                 "v_readlane_b32 %[st], %[step], %[sb]\n\t"
                 "s_bitset1_b64 %[lits], %[sb]\n\t"
                 "s_add_u32 %[sb], %[sb], %[st]\n\t"
                 "s_cbranch_scc0 1b\n\t"
                 "s_bitcmp1_b32 %[st], 6\n\t"
                 "s_cselect_b32 %[stopped], 1, 0\n\t"
                 "s_cbranch_scc0 2f\n\t"
                 "s_sub_u32 %[sb], %[sb], %[st]\n\t" // (biased again: the stop symbol's position in this part)
                 "s_bitset0_b64 %[lits], %[sb]\n\t"
                 "s_and_b32 %[sb], %[sb], 63\n"
                 "2:"
                 : [sb] "+s"(sb), [lits] "+s"(lits), [st] "=&s"(st), [stopped] "=&s"(stopped)
                 : [step] "v"(step)
                 : "scc");
    s = sb;
    return stopped != 0;
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
        if (lane < 3) ring[128 + lane] = src[(chunk + (chunk & 1u)) * 64u + lane]; // dwords 0..2 again, behind dword 127:
        pre = src[(chunk + 2u) * 64u + lane];                                       // reads of consecutive dwords never wrap
    }
    // bring the ring up to the bit position (call before reading; at most one chunk per call in the
    // symbol loops, which consume less than 256 bytes between calls)
    __device__ __forceinline__ void sync() {
        while ((bitpos >> 11) != chunk) {
            chunk += 1;
            ring[((chunk + 1u) & 1u) * 64u + (threadIdx.x & 63u)] = pre;
            if ((chunk & 1u) && (threadIdx.x & 63u) < 3) ring[128 + (threadIdx.x & 63u)] = pre;
            pre = src[(chunk + 2u) * 64u + (threadIdx.x & 63u)];
        }
    }
    // 32 bits starting p bits ahead of the bit position, per lane (p + 32 bits stay inside the two chunks: p < 2000)
    __device__ __forceinline__ uint32_t lane_bits32(uint32_t p) const {
        const uint32_t t = bitpos + p, dw = (t >> 5) & 127u;
        return __builtin_amdgcn_alignbit(ring[dw + 1u], ring[dw], t & 31u);
    }
    // the same for p and p + 64: one address, one shift
    __device__ __forceinline__ void lane_bits32x2(uint32_t p, uint32_t &a, uint32_t &b) const {
        const uint32_t t = bitpos + p, dw = (t >> 5) & 127u;
        a = __builtin_amdgcn_alignbit(ring[dw + 1u], ring[dw], t & 31u);
        b = __builtin_amdgcn_alignbit(ring[dw + 3u], ring[dw + 2u], t & 31u);
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
// Distance and code-length alphabets: lens[0..n) -> primary table of TB bits + (cnt, syms1) for codes longer than TB.
// alphabet: 1 distance, 2 code-length.  Returns false for an over-subscribed set.
__device__ bool build_table(Lds &L, const uint8_t *lens, uint32_t n, uint32_t TB, uint16_t *tab, uint32_t alphabet, uint32_t lane) {
    if (lane < 16) L.cnt[1][lane] = 0;
    for (uint32_t i = lane; i < (1u << TB); i += 64)
        tab[i] = (uint16_t)(alphabet == 1 ? dist_special(DK_INVALID) : lit_entry(0, LK_INVALID, 0));
    __syncthreads();
    for (uint32_t i = lane; i < n; i += 64) atomicAdd(&L.cnt[1][lens[i]], 1u);
    __syncthreads();
    // offsets and first codes (uniform, registers)
    uint32_t off[16], code = 0, index = 0;
    int32_t left = 1;
    bool ok = true;
#pragma unroll
    for (uint32_t l = 1; l < 16; l++) {
        const uint32_t c = uni(L.cnt[1][l]);
        left = (left << 1) - (int32_t)c;
        if (left < 0) ok = false;
        code <<= 1;
        off[l] = index;
        if (lane == 0) {
            L.start1[l] = (uint16_t)index;
            L.fcode1[l] = (uint16_t)code;
        }
        code += c;
        index += c;
    }
    if (!ok) return false;
    // canonical order: by length, then by symbol.  Rank by ballot (n <= 32: one round).
    {
        const uint32_t s = lane;
        const uint32_t l = s < n ? lens[s] : 0u;
#pragma unroll
        for (uint32_t k = 1; k < 16; k++) {
            const uint64_t m = __ballot(l == k);
            if (l == k) L.syms1[off[k] + __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)s;
        }
    }
    __syncthreads();
    // fill: every code of length <= TB owns 2^(TB-len) slots; longer codes mark their prefix slot
    for (uint32_t i = lane; i < index; i += 64) {
        const uint32_t s = L.syms1[i], l = lens[s];
        const uint32_t c = L.fcode1[l] + (i - L.start1[l]);
        const uint32_t rev = __brev(c) >> (32 - l);
        const uint32_t e = alphabet == 1 ? (s < 30 ? dist_entry(s, l) : dist_special(DK_INVALID)) : lit_entry(s, LK_LIT, l);
        if (l <= TB) {
            for (uint32_t k = rev; k < (1u << TB); k += 1u << l) tab[k] = (uint16_t)e;
        } else {
            tab[rev & ((1u << TB) - 1u)] = (uint16_t)(alphabet == 1 ? dist_special(DK_ESC) : lit_entry(0, LK_INVALID, 0)); // (code-length codes fit)
        }
    }
    __syncthreads();
    return true;
}

__device__ __forceinline__ uint32_t litlen_entry(uint32_t s, uint32_t l) {
    return s < 256 ? lit_entry(s, LK_LIT, l) : s == 256 ? lit_entry(0, LK_EOB, l) : s < 286 ? len_entry(c_len_base[s - 257], c_len_extra[s - 257], l) : lit_entry(0, LK_INVALID, l);
}
// Literal/length alphabet: lens[0..n), n <= 288 -> Lds::lit_tab (LB bits) + Lds::sub.  Every symbol is one lane's (five rounds
// of 64): its canonical code is the first code of its length + its rank among the symbols of that length (ballots), so no
// sorted symbol list is written.  Codes of at most LB bits fill their slots of the first level; a longer one notes its
// length in the slot of its first LB bits (pass A), the slots so marked get a second-level table sized by their longest
// code (pass B: 16 slots per lane, a prefix sum over the lanes), and the long codes fill those (pass C).
// Returns false for an over-subscribed set, or (an incomplete set: zlib and miniz reject every such set with more than
// one code) when the second-level tables do not fit.
__device__ __noinline__ bool build_litlen_table(Lds &L, const uint8_t *lens, uint32_t n, uint32_t lane) {
    constexpr uint32_t INV2 = lit_entry(0, LK_INVALID, 0) * 0x10001u;
    // a first-level slot while the lengths of its long codes are collected: LK_INVALID | MARK, [4:0]: bit l - LB - 1
    constexpr uint32_t MARK = 0x80u, MARKED_MASK = 0x81C0u, MARKED = 0x1C0u;
    static_assert((lit_entry(0, LK_INVALID, 0) | MARK) == MARKED, "marking is an OR on top of the initial entry");
    uint32_t *const tab32 = reinterpret_cast<uint32_t *>(L.lit_tab);
    uint32_t *const sub32 = reinterpret_cast<uint32_t *>(L.sub);
    if (lane < 16) L.cnt[0][lane] = 0;
    for (uint32_t i = lane; i < (1u << LB) / 2; i += 64) tab32[i] = INV2;
    for (uint32_t i = lane; i < SUB_CAP / 2; i += 64) sub32[i] = INV2;
    __syncthreads();
    for (uint32_t i = lane; i < n; i += 64) atomicAdd(&L.cnt[0][lens[i]], 1u);
    __syncthreads();
    uint32_t first[16], code = 0; // first canonical code of each length + the codes of that length given out so far
    int32_t left = 1;
    bool ok = true;
#pragma unroll
    for (uint32_t l = 1; l < 16; l++) {
        const uint32_t c = uni(L.cnt[0][l]);
        left = (left << 1) - (int32_t)c;
        if (left < 0) ok = false;
        code <<= 1;
        first[l] = code;
        code += c;
    }
    if (!ok) return false;
    // pass A
    uint32_t mine[5]; // per round: the symbol's code, bit-reversed (as it comes in the stream) | length << 16
#pragma unroll
    for (uint32_t b = 0; b < 5; b++) {
        const uint32_t s = b * 64 + lane;
        const uint32_t l = s < n ? lens[s] : 0u;
        uint32_t c = 0;
#pragma unroll
        for (uint32_t k = 1; k < 16; k++) {
            const uint64_t m = __ballot(l == k);
            if (l == k) c = first[k] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            first[k] += (uint32_t)__popcll(m);
        }
        const uint32_t rev = l ? __brev(c) >> (32 - l) : 0u;
        mine[b] = rev | l << 16;
        if (l && l <= LB) {
            const uint32_t e = litlen_entry(s, l);
            for (uint32_t k = rev; k < (1u << LB); k += 1u << l) L.lit_tab[k] = (uint16_t)e;
        } else if (l > LB) {
            const uint32_t slot = rev & ((1u << LB) - 1u);
            atomicOr(&tab32[slot >> 1], (MARK | 1u << (l - LB - 1u)) << (16u * (slot & 1u)));
        }
    }
    __syncthreads();
    // pass B  (loops kept rolled: the table is built once per DEFLATE block, and the registers are the window loop's)
    {
        uint32_t need = 0;
#pragma unroll 1
        for (uint32_t k = 0; k < 16; k++) {
            const uint32_t e = L.lit_tab[lane * 16 + k];
            if ((e & MARKED_MASK) == MARKED) need += 1u << (32 - __clz((int)(e & 31u)));
        }
        const uint32_t incl = wave_inclusive_sum(need);
        if ((uint32_t)__builtin_amdgcn_readlane(incl, 63) > SUB_CAP) return false;
        uint32_t at = incl - need;
#pragma unroll 1
        for (uint32_t k = 0; k < 16; k++) {
            const uint32_t e = L.lit_tab[lane * 16 + k];
            if ((e & MARKED_MASK) == MARKED) {
                const uint32_t bits = 32 - __clz((int)(e & 31u));
                L.lit_tab[lane * 16 + k] = (uint16_t)esc_entry(at, bits); // (every table has at least two entries: `at` is even)
                at += 1u << bits;
            }
        }
    }
    __syncthreads();
    // pass C
#pragma unroll
    for (uint32_t b = 0; b < 5; b++) {
        const uint32_t rev = mine[b] & 0xFFFFu, l = mine[b] >> 16;
        if (l > LB) {
            const uint32_t pe = L.lit_tab[rev & ((1u << LB) - 1u)];
            const uint32_t at = e_byte(pe) * 2u, bits = (pe >> 2) & 7u;
            const uint32_t e = litlen_entry(b * 64 + lane, l);
            for (uint32_t k = rev >> LB; k < (1u << bits); k += 1u << (l - LB)) L.sub[at + k] = (uint16_t)e;
        }
    }
    __syncthreads();
    return true;
}
// the table entry of the literal/length code at the head of the stream bits x (per lane)
__device__ __forceinline__ uint32_t litlen_lookup(const Lds &L, uint32_t x) {
    const uint32_t E = L.lit_tab[x & ((1u << LB) - 1u)];
    // (read by every lane, whatever its E: no branch around the second read, see the window loop)
    uint32_t E2 = L.sub[min(e_byte(E) * 2u + ((x >> LB) & ((1u << ((E >> 2) & 7u)) - 1u)), SUB_CAP - 1u)];
    asm volatile("" : "+v"(E2));
    return e_is_esc(E) ? E2 : E;
}

// Decode one distance symbol whose code is longer than DB bits.  x = the stream bits at the symbol.  For each
// length l the first l bits, read as a number MSB first, are a code of that length iff they fall into
// [fcode[l], fcode[l] + cnt[l]) (canonical Huffman codes).  Returns the symbol or 0xFFFF; *bits = l.
__device__ uint32_t slow_symbol(const Lds &L, uint32_t x, uint32_t *bits) {
    const uint32_t rev = __brev(x);
    for (uint32_t l = DB + 1; l < 16; l++) {
        const uint32_t code = rev >> (32 - l);
        const uint32_t f = uni(L.fcode1[l]), c = uni(L.cnt[1][l]);
        if (code - f < c) {
            *bits = l;
            return uni(L.syms1[uni(L.start1[l]) + (code - f)]);
        }
    }
    *bits = 15;
    return 0xFFFFu;
}

// full entry of a distance symbol whose primary entry says "long code"
// (out of line: rare, and bulky enough to slow the window loop down when inlined into it)
__device__ __noinline__ uint32_t resolve_long_dist(const Lds &L, uint32_t x) {
    uint32_t bits;
    const uint32_t s = slow_symbol(L, x, &bits);
    if (s == 0xFFFFu) return dist_special(DK_INVALID);
    return s < 30 ? dist_entry(s, bits) : dist_special(DK_INVALID);
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
__host__ __device__ inline uint32_t crc_mul(uint32_t a, uint32_t b) { // a * b mod P
    uint32_t p = 0;
    for (uint32_t m = 1u << 31; m; m >>= 1) {
        if (a & m) p ^= b;
        b = (b & 1u) ? (b >> 1) ^ CRC_POLY : b >> 1;
    }
    return p;
}
__host__ __device__ inline uint32_t crc_xpow8(uint32_t n_bytes) { // x^(8 n) mod P
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
    // ---- the decoded symbols and the emit
    // [0, pos) has been written (ring / HBM); the symbols in `qe` (lanes [0, nq), in stream order) cover [pos, dpos) without
    // gaps.  Entry: q_lit / q_match above.
    uint32_t dpos = 0, nq = 0;
    uint32_t qe = 0;
    // Write the next n <= 64 bytes: every output byte is produced by one lane, no loop over the symbols (a serial copy
    // per match cost ~35 scalar instructions each, and the scalar unit -- one per CU -- is what bounds this kernel):
    //   (1) the symbol that owns each byte: the symbols mark the byte they start at, a max-scan spreads the marks
    //       (byte 0 may belong to a symbol begun in an earlier chunk: the last one that starts at or before it);
    //   (2) the owner's literal byte or distance comes from the owner's lane; a match byte's source is T - distance;
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
        const uint32_t st = qe & 0xFFFFu;
        const bool queued = lane < nq;
        // (the marks are read by OTHER lanes than wrote them: the wavefront-scope fence makes the compiler reload
        // instead of forwarding this lane's own zero; LDS operations of one wave execute in order, nothing else is
        // needed.  A `volatile` pointer did that too, but it lost the LDS address space: flat_store_byte /
        // flat_load_ubyte with a full s_waitcnt after each, three round trips per 64 bytes of output.)
        uint8_t *const mark = L.e.mark;
        mark[lane] = 0;
        if (queued && st - base < 64u) mark[st - base] = (uint8_t)(lane + 1u);
        const uint32_t before = (uint32_t)__popcll(__ballot(queued && st <= base)); // (sorted by offset: an index)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        uint32_t own = mark[lane];
        if (lane == 0 && own == 0) own = before;
        own = wave_inclusive_max(own);
        const uint32_t oe = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((own - 1u) << 2), (int)qe);
        const bool active = lane < n;
        const bool lit = (oe & 0x10000u) != 0;
        const uint32_t T = base + lane, src = T - (oe >> 17) - 1u;
        const bool inchunk = active && !lit && (int32_t)(src - base) >= 0;
        uint32_t val = (oe >> 17) & 255u;
#ifdef NGSQ_INFLATE_PROFILE
        {   // (round 4: nine emits in ten of an aligner-style BAM hold bytes whose source has left the ring; 46 % of the output bytes)
            const uint64_t farm = __ballot(active && !lit && !inchunk && (int32_t)(src - base + RING) < 0);
            PROF_COUNT(3, farm != 0);
            PROF_COUNT(4, __popcll(farm));
            PROF_COUNT(5, __popcll(__ballot(inchunk)) != 0);
        }
#endif
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
        PROF_COUNT(2, 1);
    };
    // whole 64-byte chunks while there are any; all = everything decoded so far (before a stored block writes the ring, at
    // the end).  What is not written out completely moves to the first lanes: the symbols from the one that holds byte
    // `pos` on -- fewer than 64 bytes, so fewer than 64 symbols.
    auto drain = [&](bool all) {
        if (dpos - pos < 64u && !(all && dpos != pos)) return;
        while (dpos - pos >= 64u) emit_chunk(64u);
        if (all && dpos != pos) emit_chunk(dpos - pos);
        if (dpos == pos) {
            nq = 0;
        } else {
            const uint32_t k0 = (uint32_t)__popcll(__ballot(lane < nq && (qe & 0xFFFFu) <= pos)) - 1u;
            qe = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((lane + k0) << 2), (int)qe);
            nq -= k0;
        }
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
            drain(true); // the bytes go straight into the ring: behind everything decoded so far
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
            if (!build_table(L, L.lens, 19, 7, L.dist_tab, 2, lane)) {
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
        if (!build_litlen_table(L, L.lens, hlit, lane) || !build_table(L, L.lens + hlit, hdist, DB, L.dist_tab, 1, lane)) {
            err = INF_BAD_CODE_LENGTHS;
            break;
        }
        PROF(1); // block header + tables
        // ---- the symbols of the block, in batches of up to 64
        // A window of 128 bits holds ~12 symbols, and decoding them where they were found -- at one lane in ten -- made the
        // decode, the prefix sum of the output lengths and the append a third of all instructions.  So it takes two phases:
        //   (1) per WINDOW, all the lanes need is how far each symbol reaches: lane j looks up the codes that WOULD start
        //       j and j + 64 bits from here (literal/length table, then the distance table with the stream bits behind
        //       the length code and its extra bits) and adds up their bits; the real symbol starts are then the chain
        //       0 -> step[0] -> ... (one v_readlane + add per symbol on the scalar unit, literals and whole matches alike),
        //       and the lanes on the chain append their bit offset to a list;
        //   (2) per BATCH (the list is full, or the stream ring ends, or a symbol stops the chain), lane i decodes the i-th
        //       symbol of the list completely -- every lane a real symbol -- one prefix sum gives the output offsets, and
        //       the entries join the symbols left over from the last batch in `qe`.
        bool end_of_block = false;
        while (!end_of_block && err == INF_OK) {
            PROF(5);
            br.sync();
            uint32_t np = nq, rel = 0; // symbols listed (the first nq are the ones in qe), window bits behind the bit position
            bool stopped = false;
            while (np < QCAP && rel <= REL_LIMIT && !stopped) {
                uint32_t step[WH], xw[WH];
                static_assert(WH == 2, "the stream bits of a window are read for two positions per lane");
                br.lane_bits32x2(rel + lane, xw[0], xw[1]);
#pragma unroll
                for (uint32_t h = 0; h < WH; h++) {
                    const uint32_t E = litlen_lookup(L, xw[h]);
                    // the distance code (if E is a length code); looked up by every lane, no branch: the scalar unit is the
                    // busier one, and a branch around these few instructions costs it nine
                    const uint32_t xd = br.lane_bits32(rel + 64u * h + lane + e_step(E));
                    uint32_t D = L.dist_tab[xd & ((1u << DB) - 1u)];
                    asm volatile("" : "+v"(D));
                    // bits to the next symbol, or a stop mark (bit 6: end of block, a distance with a long code, an invalid code)
                    step[h] = e_step(E) + (e_is_len(E) ? d_step(D) : 0u);
                }
                PROF(2); // window bits + gathers
                PROF_COUNT(0, 1);
                // the chain of real symbol starts: readlane + add per symbol, position after position
                uint64_t syms[WH];
                uint32_t s = 0; // bits of the window consumed
#pragma unroll
                for (uint32_t h = 0; h < WH; h++) {
                    syms[h] = 0;
                    if (stopped) continue;
                    uint32_t sl = s - 64u * h; // (a step is at most 48 bits: the chain enters every 64-bit part of the window)
                    stopped = chain_literals(step[h], sl, syms[h]);
                    s = 64u * h + (stopped ? sl : sl + 64u);
                }
                // the list takes QCAP - np more symbols: a window with more ends early, the next batch starts at the first
                // symbol left out
                uint32_t n_syms = 0;
#pragma unroll
                for (uint32_t h = 0; h < WH; h++) n_syms += (uint32_t)__popcll(syms[h]);
                np = uni(np); // (short of scalar registers the compiler keeps it in a vector one, and the branch below would be a vector one)
                if (__builtin_expect(n_syms > QCAP - np, 0)) {
                    uint32_t keep = QCAP - np;
                    bool full = false;
                    stopped = false;
                    n_syms = keep;
#pragma unroll
                    for (uint32_t h = 0; h < WH; h++) {
                        const uint32_t c = (uint32_t)__popcll(syms[h]);
                        if (full) {
                            syms[h] = 0;
                        } else if (keep >= c) {
                            keep -= c;
                        } else {
                            uint64_t m = syms[h];
                            for (uint32_t k = keep; k; k--) m &= m - 1; // drop the symbols that fit
                            const uint32_t cut = (uint32_t)__builtin_ctzll(m);
                            syms[h] &= (1ull << cut) - 1ull;
                            s = 64u * h + cut;
                            full = true;
                        }
                    }
                }
                PROF(3); // chain
                {   // (the chain's lanes straight from the scalar mask, their rank among them by v_mbcnt: no vector compare)
                    uint32_t slot = np;
#pragma unroll
                    for (uint32_t h = 0; h < WH; h++) {
                        if (__builtin_amdgcn_inverse_ballot_w64(syms[h]))
                            L.e.q[slot + __builtin_amdgcn_mbcnt_hi((uint32_t)(syms[h] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)syms[h], 0u))] = rel + 64u * h + lane;
                        slot += (uint32_t)__popcll(syms[h]);
                    }
                }
                np += n_syms;
                rel += s;
                PROF(4); // list append
            }
            // phase 2
            if (np > nq) {
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); // list entries are read by other lanes than wrote them
                const bool fresh = lane >= nq && lane < np;
                const uint32_t v = fresh ? L.e.q[lane] : 0u;
                const uint32_t at_bit = v;
                const uint32_t x = br.lane_bits32(at_bit);
                const uint32_t E = litlen_lookup(L, x);
                const uint32_t lex = e_len_extra(E), nb = e_step(E) - lex; // (of a length code; the literal's byte is all it needs)
                const uint32_t len = e_byte(E) + 3u + ((x >> nb) & ((1u << lex) - 1u));
                const uint32_t xd = br.lane_bits32(at_bit + e_step(E));
                const uint32_t D = L.dist_tab[xd & ((1u << DB) - 1u)];
                const uint32_t dex = d_extra(D), db = d_step(D) - dex;
                const uint32_t dist = d_base(D) + ((xd >> db) & ((1u << dex) - 1u));
                const bool is_lit = e_is_lit(E);
                // where each symbol writes: prefix sum of the output lengths
                const uint32_t olen = !fresh ? 0u : is_lit ? 1u : len;
                const uint32_t incl = wave_inclusive_sum(olen);
                const uint32_t at = dpos + incl - olen;
                const uint32_t total = __builtin_amdgcn_readlane(incl, 63);
                PROF_COUNT(1, __popcll(__ballot(fresh && is_lit)));
                // a corrupt stream stops here, before anything of this batch is written: a distance beyond the
                // start of the output would read in front of the block's buffer, and output beyond ISIZE would be flushed
                // past its end
                if (__ballot(fresh && !is_lit && dist > at)) {
                    err = INF_BAD_DISTANCE;
                    break;
                }
                if (dpos + total > isize) {
                    err = INF_OUTPUT_OVERRUN;
                    break;
                }
                if (fresh) qe = is_lit ? q_lit(at, e_byte(E)) : q_match(at, dist);
                nq = np;
                dpos += total;
            }
            br.consume(rel);
            PROF(6); // phase 2
            drain(false);
            if (stopped) {
                // The chain stopped at a symbol the lanes could not finish: end of block, a long code (or a
                // distance code with one), or an invalid code.  One symbol the plain way, into the next free lane of qe
                // (fewer than 64 bytes are waiting there: fewer than 64 symbols).
                PROF_COUNT(6, 1);
                br.sync();
                const uint32_t x0 = br.bits32();
                const uint32_t e = uni(litlen_lookup(L, x0));
                const uint32_t eb = e_bits(e);
                if (e_is_lit(e)) {
                    if (dpos + 1 > isize) err = INF_OUTPUT_OVERRUN;
                    else {
                        if (lane == nq) qe = q_lit(dpos, e_byte(e));
                        nq += 1;
                        dpos += 1;
                        br.consume(eb);
                    }
                } else if (!e_is_len(e) && e_stop_kind(e) == LK_EOB) {
                    br.consume(eb);
                    end_of_block = true;
                } else if (e_is_len(e)) {
                    const uint32_t ex = e_len_extra(e);
                    const uint32_t l = e_byte(e) + 3u + ((x0 >> eb) & ((1u << ex) - 1u));
                    br.consume(eb + ex);
                    br.sync();
                    const uint32_t x1 = br.bits32();
                    uint32_t d = uni((uint32_t)L.dist_tab[x1 & ((1u << DB) - 1u)]);
                    if (d_kind(d) == DK_ESC) d = uni(resolve_long_dist(L, x1));
                    const uint32_t ex2 = d_extra(d), b2 = d_step(d) - ex2;
                    const uint32_t dd0 = d_base(d) + ((x1 >> b2) & ((1u << ex2) - 1u));
                    br.consume(b2 + ex2);
                    if (d_kind(d) != DK_BASE) err = INF_BAD_SYMBOL;
                    else if (dd0 > dpos) err = INF_BAD_DISTANCE;
                    else if (dpos + l > isize) err = INF_OUTPUT_OVERRUN;
                    else {
                        if (lane == nq) qe = q_match(dpos, dd0);
                        nq += 1;
                        dpos += l;
                    }
                } else {
                    err = INF_BAD_SYMBOL;
                }
            }
        }
    }
    if (err == INF_OK) drain(true);
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
// (registers: at most 80, for six decoders per SIMD -- left alone the compiler takes 92, mostly for scalars that no longer fit the
// scalar file: five per SIMD, 20 per CU, 9 % slower)
#ifndef NGSQ_INFLATE_WAVES
#define NGSQ_INFLATE_WAVES 6
#endif
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(NGSQ_INFLATE_WAVES, 8))) void k_bgzf_inflate(const uint8_t *__restrict__ comp,
                                                     const BgzfBlock *__restrict__ blocks, uint32_t n_blocks,
                                                     uint8_t *__restrict__ out, uint32_t *__restrict__ status,
                                                     uint32_t *__restrict__ next_block, uint32_t base) {
    __shared__ Lds L; // (static: with a dynamic allocation every LDS address is computed with an add of the base, zero)
    const uint32_t lane = threadIdx.x;
    for (;;) {
        uint32_t bi = 0;
        if (lane == 0) bi = atomicAdd(next_block, 1u) - base; // (the counter is never reset: `base` is what the launch found in it)
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
constexpr uint32_t CRC_MAX_SLICE = 1024; // bytes of one of the 64 slices of a block (ISIZE <= 65536)
__global__ __launch_bounds__(64 * CRC_WAVES) void k_bgzf_crc(const uint8_t *__restrict__ out, const BgzfBlock *__restrict__ blocks,
                                                             uint32_t n_blocks, uint32_t *__restrict__ status,
                                                             const uint32_t *__restrict__ pow_tab, uint32_t *__restrict__ status_host, int skip) {
    NGSQ_FOREGROUND_WAVE();
    __shared__ uint32_t s_tab[CRC_SLICES * 256];
    for (uint32_t k = threadIdx.x; k < CRC_SLICES * 256; k += 64 * CRC_WAVES) s_tab[k] = c_crc.t[k >> 8][k & 0xFFu];
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t bi = blockIdx.x * CRC_WAVES + (threadIdx.x >> 6);
    if (bi >= n_blocks) return;
    const uint32_t isize = uni(blocks[bi].isize);
    {
        const uint32_t st0 = uni(status[bi]);
        if (st0 != INF_OK) {
            if (status_host && lane == 0) status_host[bi] = st0;
            return;
        }
    }
    if (skip) { // (NGSQ_CRC_SKIP=1, measurement aid: the verdicts travel, the checksum is not computed)
        if (status_host && lane == 0) status_host[bi] = INF_OK;
        return;
    }
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
    // pairwise combination: after step j the lanes whose low j+1 bits are ones hold 2^(j+1) slices.  The multiplier of step
    // j, x^(8 S 2^j), comes from a table (a slice has at most 1024 bytes): computed here -- a dozen GF(2) multiplications of 32
    // scalar steps each, per block -- it was half of this kernel's scalar instructions, on the unit the decoders are short of.
    const uint32_t *const pw = pow_tab + uni(S) * 6u;
    for (uint32_t j = 0; j < 6; j++) {
        const uint32_t left = (uint32_t)__shfl_up((int)c, 1u << j, 64);
        if ((lane & ((2u << j) - 1u)) == (2u << j) - 1u) c = crc_mul(left, pw[j]) ^ c;
    }
    const uint32_t crc = ~__builtin_amdgcn_readlane(c, 63);
    const bool mismatch = isize ? crc != blocks[bi].crc : blocks[bi].crc != 0;
    if (lane == 0) {
        if (mismatch) status[bi] = INF_CRC_MISMATCH;
        if (status_host) status_host[bi] = mismatch ? (uint32_t)INF_CRC_MISMATCH : (uint32_t)INF_OK;
    }
}

hipError_t launch_bgzf_crc(const BgzfBlock *blocks, uint32_t n_blocks, const uint8_t *out, uint32_t *status, uint32_t *status_host, hipStream_t s) {
    if (!n_blocks) return hipSuccess;
    // x^(8 S 2^j) mod P for every slice length S and combination step j: 25 KB, computed once per device and process
    static std::mutex mu;
    static uint32_t *tab[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    uint32_t *pow_tab = nullptr;
    {
        std::lock_guard<std::mutex> g(mu);
        if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
        if (!tab[dev]) {
            std::vector<uint32_t> h((size_t)(CRC_MAX_SLICE + 1) * 6);
            for (uint32_t S = 0; S <= CRC_MAX_SLICE; S++) {
                uint32_t m = crc_xpow8(S);
                for (uint32_t j = 0; j < 6; j++) {
                    h[(size_t)S * 6 + j] = m;
                    m = crc_mul(m, m);
                }
            }
            hipError_t e = hipMalloc((void **)&tab[dev], h.size() * sizeof(uint32_t));
            if (e != hipSuccess) return e;
            e = hipMemcpy(tab[dev], h.data(), h.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
            if (e != hipSuccess) return e;
        }
        pow_tab = tab[dev];
    }
    static const int skip = getenv("NGSQ_CRC_SKIP") && atoi(getenv("NGSQ_CRC_SKIP")) ? 1 : 0; // what the file path would gain if the CRC cost nothing
    hipLaunchKernelGGL(k_bgzf_crc, dim3((n_blocks + CRC_WAVES - 1) / CRC_WAVES), dim3(64 * CRC_WAVES), 0, s, out, blocks, n_blocks, status, pow_tab, status_host, skip);
    return hipGetLastError();
}

hipError_t launch_bgzf_inflate(const uint8_t *comp, const BgzfBlock *blocks, uint32_t n_blocks, uint8_t *out,
                               uint32_t *status, uint32_t *counter, bool check_crc, hipStream_t s, uint32_t *counter_base) {
    if (!n_blocks) return hipSuccess;
    static bool attr = false;
    static uint32_t resident = 0;
    if (!attr) {
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
    const uint32_t grid = n_blocks < resident ? n_blocks : resident;
    uint32_t base = 0;
    if (counter_base) { // every decoder draws one ticket past the blocks when it leaves
        base = *counter_base;
    } else {
        hipError_t e = hipMemsetAsync(counter, 0, sizeof(uint32_t), s);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k_bgzf_inflate, dim3(grid), dim3(64), 0, s, comp, blocks, n_blocks, out, status, counter, base);
    {   // the host's copy of the never-reset counter moves only with a launch that was accepted: a refused launch leaves the
        // device word where it was, and every later launch on this stream still computes its block indices from the right base
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        if (counter_base) *counter_base = base + n_blocks + grid;
    }
    if (check_crc) (void)launch_bgzf_crc(blocks, n_blocks, out, status, nullptr, s);
#ifdef NGSQ_INFLATE_PROFILE
    {
        unsigned long long h[16];
        (void)hipStreamSynchronize(s);
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_inflate_prof), sizeof h);
        static const char *names[8] = {"other", "header+tables", "window: table reads", "window: chain", "window: list append", "emit",
                                       "batch decode", "final flush"};
        unsigned long long tot = 0;
        for (int k = 0; k < 8; k++) tot += h[k];
        for (int k = 0; k < 8; k++)
            fprintf(stderr, "[inflate-prof] %-24s %6.2f %%\n", names[k], tot ? 100.0 * (double)h[k] / (double)tot : 0.0);
        fprintf(stderr, "[inflate-prof] windows %llu, literals %llu, emits %llu, symbols taken the plain way %llu\n", h[8], h[9], h[10], h[14]);
        fprintf(stderr, "[inflate-prof] emits with bytes read back from HBM %llu (%llu bytes), emits with a source inside their own 64 bytes %llu\n", h[11], h[12], h[13]);
        unsigned long long z[16] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_inflate_prof), z, sizeof z);
    }
#endif
    return hipGetLastError();
}

} // namespace ngsq
