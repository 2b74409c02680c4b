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

#include <algorithm>

#include "ingest_kernels.h"

namespace ngsq {

namespace {

__device__ __forceinline__ uint32_t ld32(const uint8_t *p) {
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}
// the 32 bytes at p (any alignment), both loads in flight together.  Written out because hipcc narrows two 16-byte
// memcpy()s to the fields used and sinks them behind the conditions that need them: three dependent memory latencies
// per record of a chain instead of one.
__device__ __forceinline__ void ld2x16(const uint8_t *p, uint4 &a, uint4 &b) {
    asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:16\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(a), "=&v"(b)
                 : "v"(p)
                 : "memory");
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

// bam_reader.h aux_find_cg, on the device (only the rare lane whose CIGAR is the long-CIGAR placeholder walks its tags):
// offset from p of the CG:B,I tag's operations and their number, or 0
__device__ uint64_t aux_find_cg(const uint8_t *p0, const uint8_t *end, uint32_t *n_ops) {
    const uint8_t *p = p0;
    while (end - p >= 4) {
        const uint8_t t0 = p[0], t1 = p[1], ty = p[2];
        p += 3;
        uint64_t n = 0;
        if (ty == 'A' || ty == 'c' || ty == 'C') n = 1;
        else if (ty == 's' || ty == 'S') n = 2;
        else if (ty == 'i' || ty == 'I' || ty == 'f') n = 4;
        else if (ty == 'Z' || ty == 'H') {
            while (p < end && *p) p++;
            if (p >= end) return 0;
            n = 1;
        } else if (ty == 'B') {
            if (end - p < 5) return 0;
            const uint8_t sub = p[0];
            const uint32_t cnt = (uint32_t)p[1] | (uint32_t)p[2] << 8 | (uint32_t)p[3] << 16 | (uint32_t)p[4] << 24;
            const uint32_t w = sub == 'c' || sub == 'C' ? 1 : sub == 's' || sub == 'S' ? 2 : sub == 'i' || sub == 'I' || sub == 'f' ? 4 : 0;
            if (!w) return 0;
            n = 5 + (uint64_t)cnt * w;
            if (t0 == 'C' && t1 == 'G' && sub == 'I') {
                if ((uint64_t)(end - p) < n) return 0;
                *n_ops = cnt;
                return (uint64_t)(p + 5 - p0);
            }
        } else {
            return 0;
        }
        if ((uint64_t)(end - p) < n) return 0;
        p += n;
    }
    return 0;
}

// The screen of k_rec_candidates, beyond the fields' ranges.  Round 4: in a file whose records carry a real mate position the
// offset TWO BYTES IN FRONT of every record passes the range tests -- its "block_size" is the last two bytes of the record
// before and the low half of the real block_size (some tens of MB: plausible wherever that much data follows in the chunk),
// its reference ids are the zero halves of block_size / refID and of l_seq / next_refID -- and, lying in front of the real
// start, it took the segment's table slots: one segment in five of an aligner-style file was walked singly (5 M records/s
// instead of 250 M).  The files of rounds 1-3 said next_pos = -1, whose halves read as a negative position: luck.  So an
// offset is listed only if, besides, its bin is one of the six bins that can hold an alignment starting at its position
// (SAM specification 5.3: reg2bin of [pos, end) is the 16 kb window of pos or one of its five ancestors; 4680 for pos -1;
// the field is 16 bits wide), its read name ends with a NUL where l_read_name says, and its first CIGAR operation has a
// defined code.  All three are properties of every record a conforming writer produces; a record that lacks one is still
// found -- its segment is walked by k_walk_one, with the host reader's rule -- only not through the table.
__device__ __forceinline__ bool bin_fits(int32_t pos, uint32_t bin) {
    return bin == ((4681u + (uint32_t)(pos >> 14)) & 0xFFFFu) || bin == ((585u + (uint32_t)(pos >> 17)) & 0xFFFFu) ||
           bin == ((73u + (uint32_t)(pos >> 20)) & 0xFFFFu) || bin == ((9u + (uint32_t)(pos >> 23)) & 0xFFFFu) ||
           bin == ((1u + (uint32_t)(pos >> 26)) & 0xFFFFu) || bin == 0u;
}

constexpr uint32_t SUB_NONE = 0xFFFFFFFFu;
#ifndef NGSQ_PARSE_THREADS
#define NGSQ_PARSE_THREADS 256
#endif
constexpr uint32_t PT = NGSQ_PARSE_THREADS; // threads per block of the fixed-pitch parse kernels

// Walk the chain from `o` until it reaches `end` (segment end) or the record at the cursor is not
// completely inside [0, n_bytes).  STRICT: apply the plausibility test.  Returns false on an invalid
// record.  *landing = cursor at the stop, *count = records passed.  put(j, rel, cnt) is called once for
// every 4 KiB piece j of the segment the chain reaches a record start in or behind: rel = offset of the first
// record of the chain at or behind the piece's start (relative to s0), cnt = records of the chain before it.
// The fixed part of a record is fetched with two 16-byte loads issued together: one memory latency per record.
template <bool STRICT, typename Put>
__device__ __forceinline__ bool walk(const uint8_t *raw, uint64_t n_bytes, uint64_t o, uint64_t s0, uint64_t end, int32_t n_ref,
                                     uint64_t *landing, uint32_t *count, Put put) {
    uint32_t n = 0, j = 0;
    bool ok = true;
    while (o < end) {
        while (j < REC_PIECES && s0 + (uint64_t)j * REC_PIECE <= o) {
            put(j, (uint32_t)(o - s0), n);
            j++;
        }
        if (o + 36 > n_bytes) { // the fixed part itself is cut by the end of the buffer
            if (o + 4 > n_bytes) break; // block_size itself is cut
            const uint32_t bs = ld32(raw + o);
            // a complete record here would be shorter than its fixed part; of an incomplete one only block_size can be tested
            if (o + 4 + (uint64_t)bs <= n_bytes || bs < 32) ok = false;
            break;
        }
        uint4 a, b; // block_size, refID, pos, l_read_name | mapq | bin;  n_cigar_op | flag, l_seq, next_refID, next_pos
        ld2x16(raw + o, a, b);
        const uint32_t bs = a.x;
        if (o + 4 + (uint64_t)bs > n_bytes) {
            if (bs < 32) ok = false;
            break;
        }
        // the host reader's validity rule (bam_reader.cpp): block_size >= 32, l_read_name != 0 and the
        // variable-length fields fit the block
        const uint32_t l_read_name = a.w & 0xFFu, n_ops = b.x & 0xFFFFu, l = b.y;
        const uint64_t need = 32ull + l_read_name + 4ull * n_ops + ((uint64_t)l + 1) / 2 + l;
        bool good = bs >= 32 && l_read_name != 0 && need <= bs;
        if (STRICT) { // used only to FIND chains quickly (never to reject a record)
            const int32_t ref = (int32_t)a.y, pos = (int32_t)a.z, mref = (int32_t)b.z, mpos = (int32_t)b.w;
            good = good && ref >= -1 && ref < n_ref && mref >= -1 && mref < n_ref && pos >= -1 && mpos >= -1;
        }
        if (!good) {
            ok = false;
            break;
        }
        o += 4 + (uint64_t)bs;
        n += 1;
    }
    if (ok && o >= end) // the pieces behind the last record start: the landing offset is not inside them
        while (j < REC_PIECES && s0 + (uint64_t)j * REC_PIECE <= o && s0 + (uint64_t)j * REC_PIECE < end) {
            put(j, (uint32_t)(o - s0), n);
            j++;
        }
    *landing = o;
    *count = n;
    return ok;
}

} // namespace

// ---- 1. candidates ----------------------------------------------------------------------------
// One wave per REC_GROUP segments.  Per segment a window of up to 1 KiB is screened 64 offsets at a time (a complete,
// plausible record: bytes inside a record read as a huge block_size look like "the record cut by the end of the buffer" --
// never a candidate; the one true cut record of a chunk is found by k_walk_one), then the offsets that passed
// (REC_CANDIDATES per segment and round) are walked at once, a lane each.  The walk is what this kernel costs: a chain
// of dependent loads as long as the segment holds records, by one or two lanes of the wave -- so the segments are short
// (16 KiB: ~56 records of 150 bases; 64 KiB until round 3: ~226) and a wave walks the chains of four of them at once
// instead of holding a quarter as many resident waves' worth of latency.
__global__ __launch_bounds__(64) void k_rec_candidates(const uint8_t *__restrict__ raw, uint64_t n_bytes, uint64_t first,
                                                       uint32_t n_seg, int32_t n_ref, RecCandidate *__restrict__ cand,
                                                       RecPieces *__restrict__ pieces, unsigned long long *__restrict__ work) {
    NGSQ_FOREGROUND_WAVE();
    // the chunk's "smallest index of an invalid record" starts at "none" (k_rec_offsets, the next kernel of the chunk on this stream,
    // lowers it; until round 5 a memset per chunk)
    if (work && blockIdx.x == 0 && threadIdx.x == 0) work[W_BAD] = ~0ull;
    constexpr uint32_t G = REC_GROUP, LIST = REC_CANDIDATES;
    static_assert(G * LIST <= 64, "a lane per chain");
    __shared__ uint64_t s_list[G * LIST];
    __shared__ uint32_t s_rel[REC_PIECES * G * LIST], s_cnt[REC_PIECES * G * LIST];
    const uint32_t lane = threadIdx.x;
    uint32_t found[G];
    uint64_t pos[G], s0[G], s1[G];
    bool live[G];
#pragma unroll
    for (uint32_t g = 0; g < G; g++) {
        const uint64_t seg = (uint64_t)blockIdx.x * G + g;
        live[g] = seg < n_seg;
        s0[g] = seg * REC_SEGMENT;
        s1[g] = min(s0[g] + REC_SEGMENT, n_bytes);
        pos[g] = max(s0[g], first); // offsets before `first` (the BAM header) are never record starts
        found[g] = 0;
    }
    for (;;) {
        // screening: up to LIST offsets per segment that still wants candidates (each extra chain costs divergent loads, and
        // the screening resumes right behind the last offset taken, so no candidate is skipped)
        uint32_t list_n[G];
        bool any = false, more = false;
#pragma unroll
        for (uint32_t g = 0; g < G; g++) {
            list_n[g] = 0;
            if (!live[g]) continue;
            for (int it = 0; it < 16 && pos[g] < s1[g] && found[g] < REC_CANDIDATES && list_n[g] < LIST; it++) {
                const uint64_t o = pos[g] + lane;
                bool pass = false;
                if (o < s1[g] && o + 36 <= n_bytes) {
                    uint4 a, b;
                    ld2x16(raw + o, a, b);
                    const uint32_t bs = a.x, l_read_name = a.w & 0xFFu, n_ops = b.x & 0xFFFFu, l = b.y;
                    const uint64_t need = 32ull + l_read_name + 4ull * n_ops + ((uint64_t)l + 1) / 2 + l;
                    const int32_t ref = (int32_t)a.y, p = (int32_t)a.z, mref = (int32_t)b.z, mpos = (int32_t)b.w;
                    pass = o + 4 + (uint64_t)bs <= n_bytes && bs >= 32 && l_read_name != 0 && need <= bs && ref >= -1 && ref < n_ref &&
                           mref >= -1 && mref < n_ref && p >= -1 && mpos >= -1 && bin_fits(p, a.w >> 16);
                    // the few offsets that get this far: the name ends with its NUL and the first CIGAR operation is one
                    if (pass) pass = raw[o + 35 + l_read_name] == 0 && (n_ops == 0 || (ld32(raw + o + 36 + l_read_name) & 15u) <= 8u);
                }
                const uint64_t m = __ballot(pass);
                const uint32_t k = (uint32_t)__popcll(m), room = LIST - list_n[g];
                const uint32_t rank = (uint32_t)__popcll(m & ((1ull << lane) - 1));
                if (pass && rank < room) s_list[g * LIST + list_n[g] + rank] = o;
                if (k > room) { // the list is full: go on behind the last offset taken, after the walk
                    uint64_t mm = m;
                    for (uint32_t q = 1; q < room; q++) mm &= mm - 1;
                    pos[g] += (uint32_t)__builtin_ctzll(mm) + 1;
                    list_n[g] = LIST;
                } else {
                    list_n[g] += k;
                    pos[g] += 64;
                }
            }
            any = any || list_n[g] != 0;
            more = more || (pos[g] < s1[g] && found[g] < REC_CANDIDATES);
        }
        if (!any) {
            if (!more) break;
            continue;
        }
        __syncthreads();
        // the walk: lane g * LIST + k takes the k-th listed offset of segment g
        const uint32_t gl = lane / LIST, kl = lane % LIST;
        uint32_t my_n = 0;
        uint64_t my_s0 = 0, my_s1 = 0;
#pragma unroll
        for (uint32_t g = 0; g < G; g++)
            if (gl == g) {
                my_n = list_n[g];
                my_s0 = s0[g];
                my_s1 = s1[g];
            }
        uint64_t o = 0, landing = 0;
        uint32_t count = 0;
        bool ok = false;
        if (lane < G * LIST && kl < my_n) {
            o = s_list[lane];
            for (uint32_t j = 0; j < REC_PIECES; j++) s_rel[j * (G * LIST) + lane] = SUB_NONE;
            ok = walk<true>(raw, n_bytes, o, my_s0, my_s1, n_ref, &landing, &count, [&](uint32_t j, uint32_t rel, uint32_t cnt) {
                s_rel[j * (G * LIST) + lane] = rel;
                s_cnt[j * (G * LIST) + lane] = cnt;
            });
        }
#pragma unroll
        for (uint32_t g = 0; g < G; g++) {
            const uint64_t m = __ballot(ok && gl == g);
            const uint32_t slot = found[g] + (uint32_t)__popcll(m & ((1ull << lane) - 1));
            if (ok && gl == g && slot < REC_CANDIDATES) {
                const uint64_t seg = (uint64_t)blockIdx.x * G + g;
                cand[seg * REC_CANDIDATES + slot] = RecCandidate{o, landing, count, 1u};
                RecPieces &pc = pieces[seg * REC_CANDIDATES + slot];
                for (uint32_t j = 0; j < REC_PIECES; j++) {
                    pc.rel[j] = s_rel[j * (G * LIST) + lane];
                    pc.cnt[j] = s_cnt[j * (G * LIST) + lane];
                }
            }
            found[g] += (uint32_t)__popcll(m);
        }
        __syncthreads();
    }
#pragma unroll
    for (uint32_t g = 0; g < G; g++) {
        if (!live[g]) continue;
        const uint64_t seg = (uint64_t)blockIdx.x * G + g;
        for (uint32_t k = min(found[g], REC_CANDIDATES) + lane; k < REC_CANDIDATES; k += 64) cand[seg * REC_CANDIDATES + k] = RecCandidate{0, 0, 0, 0u};
    }
}

// ---- 2b. the rare segment whose entry is not in the table ---------------------------------------
__global__ void k_walk_one(const uint8_t *__restrict__ raw, uint64_t n_bytes, uint64_t start, uint64_t s0, uint64_t end,
                           RecCandidate *__restrict__ out, RecPieces *__restrict__ pc) {
    NGSQ_FOREGROUND_WAVE();
    if (threadIdx.x || blockIdx.x) return;
    uint64_t landing = start;
    uint32_t count = 0;
    for (uint32_t j = 0; j < REC_PIECES; j++) pc->rel[j] = SUB_NONE;
    const bool ok = walk<false>(raw, n_bytes, start, s0, end, 0, &landing, &count, [&](uint32_t j, uint32_t rel, uint32_t cnt) {
        pc->rel[j] = rel;
        pc->cnt[j] = cnt;
    });
    *out = RecCandidate{start, landing, count, ok ? 1u : 0u};
}

// ---- 3. offsets -------------------------------------------------------------------------------
// One lane per 4 KiB piece.  chosen[s] = the candidate of segment s whose chain is the file's (REC_NO_CHAIN: no record
// starts in s), seg_base[s] = index of its first record; the piece's own entry comes from the candidate's piece table.
// bad[0] = smallest index of an invalid record (or ~0).
__global__ __launch_bounds__(PT) void k_rec_offsets(const uint8_t *__restrict__ raw, uint64_t n_bytes, uint32_t n_pieces,
                                                     const uint32_t *__restrict__ chosen, const uint32_t *__restrict__ seg_base,
                                                     const RecPieces *__restrict__ pieces, uint64_t *__restrict__ rec_off,
                                                     unsigned long long *__restrict__ bad) {
    NGSQ_FOREGROUND_WAVE();
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_pieces) return;
    const uint32_t seg = g / REC_PIECES, j = g % REC_PIECES, ch = chosen[seg];
    if (ch == REC_NO_CHAIN) return;
    const RecPieces &pc = pieces[(uint64_t)seg * REC_CANDIDATES + ch];
    if (pc.rel[j] == SUB_NONE) return;
    const uint64_t s1 = min(((uint64_t)g + 1) * REC_PIECE, n_bytes);
    uint64_t o = (uint64_t)seg * REC_SEGMENT + pc.rel[j], i = seg_base[seg] + pc.cnt[j];
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
// work (device, REC_WORK_WORDS words, kept between launches): [W_BAD] smallest index of an invalid record of the chunk (k_rec_offsets;
// ~0: none -- a launch that finds one there does nothing but report it), then this launch's tallies: max l_seq, max n_cigar_op,
// sum l_seq, (refID << 32 | pos) of the first / last record, records whose CIGAR came from a CG:B,I tag (specification 4.2.2),
// sum n_cigar_op, and a ticket.  The block that finishes LAST writes the eight numbers into `host` -- pinned host memory the device
// addresses -- and resets the tallies for the next launch: until round 4 the caller reset them with two memsets and fetched them
// with a copy kernel of its own (k_copy_words: 6 launches per chunk, 0.14 ms each beside the reader's DMA, three of them on the
// inflate stream; 13 % of the GPU's time in a file scan).
// record_id (include/ngsq.h): the record's BAM virtual offset -- the block whose data holds its first byte is found
// by bisection over the chunk's block table (out_off ascending; blocks without data never hold a byte), the record
// carried over from the previous chunk (view offset < carry) has the id it was given there.  var_base[i] / seq_src[i] = offset of record i's CIGAR / SEQ
// in raw (k_rec_var and k_rec_rows start from them).  The fixed part of a record is 32 contiguous bytes at any byte
// offset: two unaligned 16-byte loads.  cig_len (optional, n + 1 entries): n_cigar_op of every record and a 0 behind them (the
// offsets of the CIGAR column are its prefix sums).
__global__ __launch_bounds__(PT) void k_rec_fixed(const uint8_t *__restrict__ raw, const uint64_t *__restrict__ rec_off,
                                                   uint64_t n, RecColumns c, uint64_t *__restrict__ var_base,
                                                   uint64_t *__restrict__ seq_src, unsigned long long *__restrict__ work,
                                                   unsigned long long *__restrict__ host, RecOrigin org, uint64_t *__restrict__ cig_len) {
    NGSQ_FOREGROUND_WAVE();
    constexpr uint32_t NW = PT / 64;
    __shared__ uint32_t s_ml[NW], s_mo[NW];
    __shared__ unsigned long long s_sl[NW], s_so[NW];
    __shared__ uint32_t s_last;
    {
        const unsigned long long bad = __hip_atomic_load(&work[W_BAD], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (bad != ~0ull) { // (uniform over the grid: rec_off holds nothing behind that record)
            if (blockIdx.x == 0 && threadIdx.x == 0) host[H_BAD] = bad;
            return;
        }
    }
    uint32_t ml = 0, mo = 0;
    unsigned long long sl = 0, so = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * PT + threadIdx.x; i < n; i += (uint64_t)gridDim.x * PT) {
        const uint64_t o = rec_off[i] + 4;
        uint4 a, b;
        __builtin_memcpy(&a, raw + o, 16);      // refID, pos, l_read_name | mapq | bin, n_cigar_op | flag
        __builtin_memcpy(&b, raw + o + 16, 16); // l_seq, next_refID, next_pos, tlen
        const uint32_t l_read_name = a.z & 0xFFu, n_ops = a.w & 0xFFFFu, l = b.x;
        c.ref_id[i] = (int32_t)a.x;
        c.pos[i] = (int32_t)a.y;
        c.mapq[i] = (uint8_t)(a.z >> 8);
        c.flag[i] = (uint16_t)(a.w >> 16);
        c.l_seq[i] = l;
        c.mate_ref_id[i] = (int32_t)b.y;
        c.tlen[i] = (int32_t)b.w;
        uint64_t cig_at = o + 32 + l_read_name;
        uint32_t real_ops = n_ops;
        if (n_ops == 2 && l) { // the long-CIGAR placeholder <l_seq>S<span>N (specification 4.2.2): the operations are in the record's CG:B,I tag
            const uint32_t op0 = ld32(raw + cig_at), op1 = ld32(raw + cig_at + 4);
            if (op0 == (l << 4 | 4u) && (op1 & 15u) == 3u) {
                const uint32_t bs = ld32(raw + o - 4); // block_size: the record ends at o + bs
                const uint64_t need = 32ull + l_read_name + 8ull + ((uint64_t)l + 1) / 2 + l;
                uint32_t cnt = 0;
                const uint64_t at = need <= bs ? aux_find_cg(raw + o + need, raw + o + bs, &cnt) : 0;
                if (at && cnt >= 2) { // (as the host reader: a tag with fewer operations than the placeholder is ignored)
                    cig_at = o + need + at;
                    real_ops = cnt;
                    (void)atomicAdd(&work[W_LONG], 1ull);
                }
            }
        }
        c.n_cigar[i] = (uint16_t)min(real_ops, 0xFFFFu);
        var_base[i] = cig_at;
        seq_src[i] = o + 32 + l_read_name + 4ull * n_ops;
        if (cig_len) {
            cig_len[i] = real_ops;
            if (i == n - 1) cig_len[n] = 0;
        }
        if (org.record_id) {
            const uint64_t at = o - 4; // view offset of the record's first byte
            uint64_t id = org.carry_id;
            if (at >= org.carry) {
                const uint64_t u = at - org.carry;
                uint32_t lo_k = 0, hi_k = org.n_blocks; // last block with out_off <= u
                while (hi_k - lo_k > 1) {
                    const uint32_t mid = (lo_k + hi_k) >> 1;
                    if (org.blocks[mid].out_off <= u) lo_k = mid;
                    else hi_k = mid;
                }
                id = org.coff[lo_k] << 16 | (u - org.blocks[lo_k].out_off);
            }
            org.record_id[i] = id;
        }
        if (i == 0) (void)atomicExch(&work[W_FIRST], (unsigned long long)a.x << 32 | a.y);
        if (i == n - 1) (void)atomicExch(&work[W_LAST], (unsigned long long)a.x << 32 | a.y);
        ml = max(ml, l);
        mo = max(mo, real_ops);
        sl += l;
        so += real_ops;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        ml = max(ml, (uint32_t)__shfl_xor((int)ml, o, 64));
        mo = max(mo, (uint32_t)__shfl_xor((int)mo, o, 64));
        sl += __shfl_xor(sl, o, 64);
        so += __shfl_xor(so, o, 64);
    }
    // this thread's tallies (W_LONG / W_FIRST / W_LAST) have reached the L2 before the block's barrier is passed (the
    // barrier itself only waits for LDS, and the atomics return nothing a later instruction would wait for)
    NGSQ_WAIT_VMEM();
    const uint32_t w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        s_ml[w] = ml;
        s_mo[w] = mo;
        s_sl[w] = sl;
        s_so[w] = so;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t a = 0, b = 0;
        unsigned long long t = 0, u = 0;
        for (uint32_t k = 0; k < NW; k++) {
            a = max(a, s_ml[k]);
            b = max(b, s_mo[k]);
            t += s_sl[k];
            u += s_so[k];
        }
        // the block's tallies, then its ticket.  What puts them in front of it is an explicit `s_waitcnt vmcnt(0)`: the tallies
        // are device-scope atomics executed at the L2, acknowledged in order, and the ticket is not issued before the last of
        // them has been -- no release fence (on gfx950 an agent-scope release is a write-back of the XCD's L2), and nothing the
        // optimiser can fold away (until round 4 the ticket's operand was meant to depend on the atomics' results: it was
        // constant-folded and the atomics became no-return instructions nobody waited for -- ADVICE r4;
        // tests/test_abi.py::test_rec_fixed_ticket_is_ordered_behind_the_tallies reads the ISA)
        (void)atomicMax(&work[W_MAXL], (unsigned long long)a);
        (void)atomicMax(&work[W_MAXOPS], (unsigned long long)b);
        (void)atomicAdd(&work[W_SUML], t);
        (void)atomicAdd(&work[W_SUMOPS], u);
        NGSQ_WAIT_VMEM();
        const unsigned long long ticket = atomicAdd(&work[W_TICKET], 1ull);
        s_last = ticket == (unsigned long long)gridDim.x - 1ull;
        if (s_last) {
            host[H_MAXL] = atomicExch(&work[W_MAXL], 0ull);
            host[H_MAXOPS] = atomicExch(&work[W_MAXOPS], 0ull);
            host[H_SUML] = atomicExch(&work[W_SUML], 0ull);
            host[H_FIRST] = atomicExch(&work[W_FIRST], 0ull);
            host[H_LAST] = atomicExch(&work[W_LAST], 0ull);
            host[H_LONG] = atomicExch(&work[W_LONG], 0ull);
            host[H_SUMOPS] = atomicExch(&work[W_SUMOPS], 0ull);
            host[H_BAD] = ~0ull;
            (void)atomicExch(&work[W_TICKET], 0ull);
        }
    }
}

// ---- 5. per-record lengths for the offsets layout ----------------------------------------------
// Absent qualities (l_seq bytes of 0xFF, SAM/BAM spec 4.2.3; noodles yields no scores) take no bytes.
__global__ __launch_bounds__(256) void k_rec_lengths(const uint8_t *__restrict__ raw, const uint64_t *__restrict__ rec_off,
                                                     uint64_t n, uint64_t *__restrict__ seq_len,
                                                     uint64_t *__restrict__ qual_len, uint64_t *__restrict__ cig_len) {
    NGSQ_FOREGROUND_WAVE();
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t *r = raw + rec_off[i] + 4;
    const uint32_t l_read_name = r[8], n_ops = ld16(r + 12), l = ld32(r + 16);
    const uint8_t *ql = r + 32 + l_read_name + 4ull * n_ops + (l + 1) / 2;
    bool miss = l > 0;
    for (uint32_t k = 0; k < l && miss; k++) miss = ql[k] == 0xFF;
    seq_len[i] = (l + 1) / 2;
    qual_len[i] = miss ? 0 : l;
    if (cig_len) cig_len[i] = n_ops;
}

// ---- 6. variable-width columns -----------------------------------------------------------------
// Fixed-pitch rows (the common case: reads of one length) are written a DWORD of the destination per lane:
// the rows of consecutive records are one contiguous region, dword d of it belongs to record d*4/pitch, and
// unless it holds the end of a row its four bytes are four consecutive bytes of the source: one unaligned load,
// one aligned store per four bytes (a byte per lane, sixteen lanes per record, was 4 + 4 instructions).
// Padding as bam_reader.cpp pads: zero nibbles for SEQ, 0xFF for QUAL.
template <bool QUAL>
__device__ __forceinline__ uint32_t row_tail(const uint8_t *__restrict__ raw, const uint64_t *__restrict__ seq_src,
                                             const uint32_t *__restrict__ l_seq, uint64_t n, uint32_t r, uint32_t k, uint32_t pitch,
                                             uint32_t len, const uint8_t *src) {
    // the end of a row: its padding, and the first bytes of the next row -- of the next THREE rows when the pitch is
    // below four bytes (reads of one to three bases)
    constexpr uint32_t FILL = QUAL ? 0xFFu : 0u;
    uint32_t v = 0;
#pragma unroll
    for (uint32_t b = 0; b < 4; b++) {
        uint32_t byte = FILL, rb = r, kb = k + b;
        while (kb >= pitch) {
            kb -= pitch;
            rb += 1;
        }
        if (rb == r) {
            if (kb < len) byte = src[kb];
        } else if (rb < n) {
            const uint32_t l2 = l_seq[rb], sb2 = (l2 + 1) / 2;
            if (kb < (QUAL ? l2 : sb2)) byte = raw[seq_src[rb] + (QUAL ? sb2 : 0u) + kb];
        }
        v |= byte << (8 * b);
    }
    return v;
}

// seq_src[i] = offset of record i's SEQ in raw.  Four destination dwords per thread, their loads issued together:
// record index -> (seq_src, l_seq) -> source bytes is two dependent memory latencies.  This kernel serves rows
// narrower than a dword (reads of one to three bases / one to six bases: a dword then spans up to four rows);
// k_rec_rows below is the one for everything else.
template <bool QUAL>
__global__ __launch_bounds__(256) void k_rec_rows_narrow(const uint8_t *__restrict__ raw, const uint64_t *__restrict__ seq_src,
                                                         const uint32_t *__restrict__ l_seq, uint64_t n, uint32_t *__restrict__ dst,
                                                         uint32_t pitch, uint64_t n_dwords) {
    NGSQ_FOREGROUND_WAVE();
    constexpr uint32_t FILL = QUAL ? 0xFFu : 0u;
    constexpr int U = 4;
    for (uint64_t d0 = (uint64_t)blockIdx.x * (256 * U) + threadIdx.x; d0 < n_dwords; d0 += (uint64_t)gridDim.x * (256 * U)) {
        uint32_t r[U], k[U], len[U], v[U];
        const uint8_t *src[U];
        bool in[U], fast[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint64_t d = d0 + (uint64_t)u * 256;
            const uint32_t a = (uint32_t)d * 4u; // pitch * n < 2^32 (launcher)
            r[u] = a / pitch;
            k[u] = a - r[u] * pitch;
            in[u] = d < n_dwords && r[u] < n;
            const uint32_t rr = in[u] ? r[u] : 0u;
            const uint32_t l = l_seq[rr], sb = (l + 1) / 2;
            len[u] = QUAL ? l : sb;
            src[u] = raw + seq_src[rr] + (QUAL ? sb : 0u);
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            fast[u] = in[u] && k[u] + 4 <= len[u];
            v[u] = ld32(fast[u] ? src[u] + k[u] : src[u]);
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint64_t d = d0 + (uint64_t)u * 256;
            if (d >= n_dwords) break;
            if (!in[u]) v[u] = FILL * 0x01010101u;
            else if (!fast[u]) v[u] = row_tail<QUAL>(raw, seq_src, l_seq, n, r[u], k[u], pitch, len[u], src[u]);
            dst[d] = v[u];
        }
    }
}

// The kernel for rows of four bytes and more.  A destination dword holds bytes of at most two rows: its own row's bytes
// and the first bytes of the next row are BOTH fetched (one unaligned dword each) and the result is put together with
// byte masks -- no branch.  With the end of a row handled on a path of its own, every wave held a few such lanes and
// walked that path's dependent loads with the others idle: 0.64 ms for the two columns of a 512 MiB chunk, 0.36 ms this
// way (a plain copy of the same bytes: 0.15 ms; tools/cand_probe.hip's sibling measurements in DESIGN.md section 9).
__device__ __forceinline__ uint32_t byte_mask(uint32_t n_bytes) { return n_bytes >= 4 ? 0xFFFFFFFFu : (1u << (8 * n_bytes)) - 1u; }

template <bool QUAL>
__global__ __launch_bounds__(PT) void k_rec_rows(const uint8_t *__restrict__ raw, const uint64_t *__restrict__ seq_src,
                                                  const uint32_t *__restrict__ l_seq, uint64_t n, uint32_t *__restrict__ dst,
                                                  uint32_t pitch, uint64_t n_dwords) {
    NGSQ_FOREGROUND_WAVE();
    constexpr uint32_t FILLW = QUAL ? 0xFFFFFFFFu : 0u;
    constexpr int U = 4;
    for (uint64_t d0 = (uint64_t)blockIdx.x * (PT * U) + threadIdx.x; d0 < n_dwords; d0 += (uint64_t)gridDim.x * (PT * U)) {
        uint32_t k[U], len_a[U], len_b[U];
        const uint8_t *pa[U], *pb[U];
        bool in[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint64_t d = d0 + (uint64_t)u * PT;
            const uint32_t a = (uint32_t)d * 4u; // pitch * n < 2^32 (launcher)
            const uint32_t r = a / pitch;
            k[u] = a - r * pitch;
            in[u] = d < n_dwords && r < n;
            const bool next = in[u] && (uint64_t)r + 1 < n;
            const uint32_t r0 = in[u] ? r : 0u, r1 = next ? r + 1 : r0;
            const uint32_t l0 = l_seq[r0], l1 = l_seq[r1];
            const uint32_t sb0 = (l0 + 1) / 2, sb1 = (l1 + 1) / 2;
            len_a[u] = QUAL ? l0 : sb0;
            len_b[u] = next ? (QUAL ? l1 : sb1) : 0u;
            // (a dword in a short read's padding reads the row's first bytes instead of bytes far behind it: nothing of it
            // is kept; the raw buffer has slack for the three bytes a dword may reach behind the data)
            pa[u] = raw + seq_src[r0] + (QUAL ? sb0 : 0u) + (k[u] < len_a[u] ? k[u] : 0u);
            pb[u] = raw + seq_src[r1] + (QUAL ? sb1 : 0u);
        }
        uint32_t va[U], vb[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            va[u] = ld32(pa[u]);
            vb[u] = ld32(pb[u]);
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint64_t d = d0 + (uint64_t)u * PT;
            if (d >= n_dwords) break;
            const uint32_t t = pitch - k[u];                                 // bytes of this dword that lie in row r (>= 1)
            const uint32_t n_a = len_a[u] > k[u] ? len_a[u] - k[u] : 0u;     // ... of them inside the row's data
            const uint32_t m_a = in[u] ? byte_mask(n_a < t ? n_a : t) : 0u;
            const uint32_t n_b = t < 4 ? (len_b[u] < 4 - t ? len_b[u] : 4 - t) : 0u; // bytes of the next row's data
            const uint32_t m_b = t < 4 ? byte_mask(n_b) << (8 * t) : 0u;
            const uint32_t sh_b = t < 4 ? vb[u] << (8 * t) : 0u;
            dst[d] = (va[u] & m_a) | (sh_b & m_b) | (FILLW & ~(m_a | m_b));
        }
    }
}

// 16 lanes per record: the offsets layout (reads of different lengths) copies l_seq / (l_seq+1)/2 / n_ops units;
// parts: 1 = CIGAR, 2 = SEQ and QUAL (fixed-pitch rows are k_rec_rows' unless the pitch rules it out).
__global__ __launch_bounds__(256) void k_rec_var(const uint8_t *__restrict__ raw, const uint64_t *__restrict__ var_base,
                                                 uint64_t n, RecColumns c, uint32_t parts) {
    NGSQ_FOREGROUND_WAVE();
    const uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const uint32_t t = threadIdx.x & 15u;
    if (i >= n) return;
    const uint32_t l = c.l_seq[i];
    // (a record whose operations sit in its CG tag: their number is in the offsets, var_base points into the tag, SEQ is where
    // the record's own fields say)
    const uint32_t n_ops = c.cigar_off ? (uint32_t)(c.cigar_off[i + 1] - c.cigar_off[i]) : c.n_cigar[i];
    const uint8_t *cg = raw + var_base[i];
    const uint8_t *sq = raw + var_base[n + i];
    const uint8_t *ql = sq + (l + 1) / 2;
    const uint32_t sb = (l + 1) / 2;
    if (parts & 1u) {
        if (c.cigar_off) {
            uint32_t *dst = c.cigar + c.cigar_off[i];
            for (uint32_t k = t; k < n_ops; k += 16) dst[k] = ld32(cg + 4 * k);
        } else if (t == 0) {
            c.cigar[i] = n_ops ? ld32(cg) : 0u;
        }
    }
    if (!(parts & 2u)) return;
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

// The CIGAR column alone in the offsets layout (fixed-pitch SEQ / QUAL rows beside it: an aligner's file, where one read in
// seven has clips or indels): FOUR lanes per record -- a record has 1.3 operations on average, and with k_rec_var's sixteen
// lanes fifteen of them only ran the kernel's prologue (0.3 ms per 1.4 M records beside the decoders).
__global__ __launch_bounds__(256) void k_rec_cigar_var(const uint8_t *__restrict__ raw, const uint64_t *__restrict__ var_base, uint64_t n,
                                                       const uint64_t *__restrict__ cigar_off, uint32_t *__restrict__ cigar) {
    NGSQ_FOREGROUND_WAVE();
    const uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
    const uint32_t t = threadIdx.x & 3u;
    if (i >= n) return;
    const uint64_t c0 = cigar_off[i];
    const uint32_t n_ops = (uint32_t)(cigar_off[i + 1] - c0);
    if (t >= n_ops) return;
    const uint8_t *cg = raw + var_base[i];
    for (uint32_t k = t; k < n_ops; k += 4) cigar[c0 + k] = ld32(cg + 4 * k);
}

// one CIGAR operation per record (or none): the first one, or 0
__global__ __launch_bounds__(PT) void k_rec_cigar1(const uint8_t *__restrict__ raw, const uint64_t *__restrict__ var_base,
                                                    const uint16_t *__restrict__ n_cigar, uint64_t n, uint32_t *__restrict__ cigar) {
    NGSQ_FOREGROUND_WAVE();
    for (uint64_t i = (uint64_t)blockIdx.x * PT + threadIdx.x; i < n; i += (uint64_t)gridDim.x * PT)
        cigar[i] = n_cigar[i] ? ld32(raw + var_base[i]) : 0u;
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
    NGSQ_FOREGROUND_WAVE();
    if (threadIdx.x || blockIdx.x) return;
    uint64_t lo = 0, hi = n;
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (a[mid] < value) lo = mid + 1;
        else hi = mid;
    }
    *out = lo;
}

// Small tables and results between host and device WITHOUT the DMA engines: one side is pinned host memory the device
// addresses directly.  Every hipMemcpyAsync on any stream -- pinned or pageable, either direction -- queues behind the
// reader thread's 64 MiB host-to-device copies of the next chunk's compressed bytes (in-order DMA: up to 1.2 ms each,
// tools/stream_order_probe.hip); a dozen such copies per chunk sat on the pipeline's critical path and were what kept
// the parse of one chunk from running beside the inflate of the next.
// (16 bytes per lane where both sides are 16-byte aligned: the segment verdicts of a chunk -- 512 KB read out of pinned host
// memory over PCIe -- took 130 us a dword per lane, a fifth of the parse kernels' time beside the inflate)
__global__ __launch_bounds__(256) void k_copy_words(uint32_t *__restrict__ dst, const uint32_t *__restrict__ src, uint64_t n) {
    NGSQ_FOREGROUND_WAVE();
    uint64_t n4 = 0;
    if (((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 15u) == 0) {
        n4 = n / 4;
        const uint4 *const s4 = reinterpret_cast<const uint4 *>(src);
        uint4 *const d4 = reinterpret_cast<uint4 *>(dst);
        for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (uint64_t)gridDim.x * 256) d4[i] = s4[i];
    }
    for (uint64_t i = 4 * n4 + (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) dst[i] = src[i];
}

// device-to-device bytes at any alignment (the record cut by a chunk's end moves in front of the next chunk: a few hundred
// bytes -- as a hipMemcpyAsync it went through the DMA queue, BEHIND the reader thread's 180 MB of the next chunk's compressed
// bytes, and held the context's stream for up to 3.5 ms per chunk: most of what index_records seemed to take, round 4)
__global__ __launch_bounds__(256) void k_copy_bytes(uint8_t *__restrict__ dst, const uint8_t *__restrict__ src, uint64_t n) {
    NGSQ_FOREGROUND_WAVE();
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) dst[i] = src[i];
}

// ---- launchers ----------------------------------------------------------------------------------
hipError_t launch_copy_bytes(void *dst, const void *src, uint64_t n_bytes, hipStream_t s) {
    if (!n_bytes) return hipSuccess;
    hipLaunchKernelGGL(k_copy_bytes, dim3((uint32_t)std::min<uint64_t>((n_bytes + 255) / 256, 4096)), dim3(256), 0, s, static_cast<uint8_t *>(dst),
                       static_cast<const uint8_t *>(src), n_bytes);
    return hipGetLastError();
}
hipError_t launch_copy_words(void *dst, const void *src, uint64_t n_bytes, hipStream_t s) {
    const uint64_t n = (n_bytes + 3) / 4; // (the buffers are allocated in whole words)
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(k_copy_words, dim3((uint32_t)std::min<uint64_t>((n / 4 + 255) / 256 + 1, 1024)), dim3(256), 0, s,
                       static_cast<uint32_t *>(dst), static_cast<const uint32_t *>(src), n);
    return hipGetLastError();
}
hipError_t launch_rec_candidates(const uint8_t *raw, uint64_t n_bytes, uint64_t first, uint32_t n_seg, int32_t n_ref,
                                 RecCandidate *cand, RecPieces *pieces, unsigned long long *work, hipStream_t s) {
    if (!n_seg) return hipSuccess;
    hipLaunchKernelGGL(k_rec_candidates, dim3((n_seg + REC_GROUP - 1) / REC_GROUP), dim3(64), 0, s, raw, n_bytes, first, n_seg, n_ref, cand, pieces, work);
    return hipGetLastError();
}
hipError_t launch_walk_one(const uint8_t *raw, uint64_t n_bytes, uint64_t start, uint64_t seg_start, uint64_t end, RecCandidate *out,
                           RecPieces *pieces, hipStream_t s) {
    hipLaunchKernelGGL(k_walk_one, dim3(1), dim3(64), 0, s, raw, n_bytes, start, seg_start, end, out, pieces);
    return hipGetLastError();
}
hipError_t launch_rec_offsets(const uint8_t *raw, uint64_t n_bytes, uint32_t n_pieces, const uint32_t *chosen, const uint32_t *seg_base,
                              const RecPieces *pieces, uint64_t *rec_off, unsigned long long *work, hipStream_t s) {
    if (!n_pieces) return hipSuccess;
    hipLaunchKernelGGL(k_rec_offsets, dim3((n_pieces + PT - 1) / PT), dim3(PT), 0, s, raw, n_bytes, n_pieces, chosen, seg_base, pieces,
                       rec_off, work + W_BAD);
    return hipGetLastError();
}
hipError_t launch_count_below_u64(const uint64_t *a, uint64_t n, uint64_t value, unsigned long long *out, hipStream_t s) {
    hipLaunchKernelGGL(k_count_below_u64, dim3(1), dim3(64), 0, s, a, n, value, out);
    return hipGetLastError();
}
hipError_t launch_rec_fixed(const uint8_t *raw, const uint64_t *rec_off, uint64_t n, const RecColumns &c, uint64_t *var_base,
                            unsigned long long *work, unsigned long long *host, const RecOrigin &org, uint64_t *cig_len, hipStream_t s) {
    if (!n) return hipSuccess;
    const uint32_t blocks = (uint32_t)std::min<uint64_t>((n + PT - 1) / PT, 2048 * (256 / PT));
    hipLaunchKernelGGL(k_rec_fixed, dim3(blocks), dim3(PT), 0, s, raw, rec_off, n, c, var_base, var_base + n, work, host, org, cig_len);
    return hipGetLastError();
}
hipError_t launch_rec_lengths(const uint8_t *raw, const uint64_t *rec_off, uint64_t n, uint64_t *seq_len,
                              uint64_t *qual_len, uint64_t *cig_len, hipStream_t s) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(k_rec_lengths, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, raw, rec_off, n, seq_len,
                       qual_len, cig_len);
    return hipGetLastError();
}
// ---- exclusive prefix sums of 64-bit entries in place (the offsets of the variable-width columns) ----------------
// Two small launches: the sum of every 4096-entry piece; every piece scanned behind its carry, which the piece's block adds up
// itself from the sums in front of it (a chunk of a file is a few hundred pieces).  Beyond SCAN_OWN_CARRY pieces a launch
// between the two scans the sums instead.  (Until round 5 this was hipcub::DeviceScan::ExclusiveSum, the one library call on the
// path.  The first version of this had the middle launch always, ONE block of 1024 threads: it needs a whole compute unit's
// register file at once, and beside the persistent inflate decoders of the next chunk, which hold 24 waves on every compute unit
// until their launch ends, it waited for exactly that -- 1.27 ms per chunk of an aligner's file for a scan of 355 numbers,
// profiles/r05_ingest_kernel_stats.csv; the library's look-back scan had taken 15 us there.)
constexpr uint64_t SCAN_OWN_CARRY = 8192;
constexpr uint32_t SCAN_PIECE = 4096, SCAN_T = 256, SCAN_PER = SCAN_PIECE / SCAN_T;
__device__ __forceinline__ uint64_t scan_wave_inclusive(uint64_t v, uint32_t lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint64_t up = __shfl_up(v, o, 64);
        if ((int)lane >= o) v += up;
    }
    return v;
}
__global__ __launch_bounds__(SCAN_T) void k_scan_sums(const uint64_t *__restrict__ data, uint64_t n, uint64_t *__restrict__ sums) {
    NGSQ_FOREGROUND_WAVE();
    __shared__ uint64_t s_w[SCAN_T / 64];
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_PIECE + (uint64_t)threadIdx.x * SCAN_PER;
    uint64_t t = 0;
#pragma unroll
    for (uint32_t k = 0; k < SCAN_PER; k++)
        if (base + k < n) t += data[base + k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) t += __shfl_down(t, o, 64);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) sums[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}
__global__ __launch_bounds__(256) void k_scan_of_sums(uint64_t *__restrict__ sums, uint64_t n) {
    NGSQ_FOREGROUND_WAVE();
    __shared__ uint64_t s_part[256];
    const uint64_t per = (n + 255) / 256, lo = (uint64_t)threadIdx.x * per, hi = lo + per < n ? lo + per : n;
    uint64_t t = 0;
    for (uint64_t i = lo; i < hi; i++) t += sums[i];
    s_part[threadIdx.x] = t;
    __syncthreads();
    if (threadIdx.x < 64) { // the 256 partial sums: four per lane of the first wave
        uint64_t loc[4], run = 0;
#pragma unroll
        for (uint32_t k = 0; k < 4; k++) {
            loc[k] = run;
            run += s_part[threadIdx.x * 4 + k];
        }
        const uint64_t before = scan_wave_inclusive(run, threadIdx.x) - run;
#pragma unroll
        for (uint32_t k = 0; k < 4; k++) s_part[threadIdx.x * 4 + k] = before + loc[k];
    }
    __syncthreads();
    uint64_t run = s_part[threadIdx.x];
    for (uint64_t i = lo; i < hi; i++) {
        const uint64_t v = sums[i];
        sums[i] = run;
        run += v;
    }
}
// OWN_CARRY: sums[] holds the pieces' sums and the block adds up those in front of its piece; else sums[] holds their exclusive scan
template <bool OWN_CARRY>
__global__ __launch_bounds__(SCAN_T) void k_scan_apply(uint64_t *__restrict__ data, uint64_t n, const uint64_t *__restrict__ sums) {
    NGSQ_FOREGROUND_WAVE();
    __shared__ uint64_t s_w[SCAN_T / 64], s_c[SCAN_T / 64];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t carry = 0;
    if (OWN_CARRY) {
        for (uint32_t i = threadIdx.x; i < blockIdx.x; i += SCAN_T) carry += sums[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) carry += __shfl_down(carry, o, 64);
        if (lane == 0) s_c[wave] = carry;
    }
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_PIECE + (uint64_t)threadIdx.x * SCAN_PER;
    uint64_t v[SCAN_PER], t = 0;
#pragma unroll
    for (uint32_t k = 0; k < SCAN_PER; k++) {
        v[k] = base + k < n ? data[base + k] : 0;
        t += v[k];
    }
    const uint64_t inc = scan_wave_inclusive(t, lane);
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    uint64_t run = (OWN_CARRY ? s_c[0] + s_c[1] + s_c[2] + s_c[3] : sums[blockIdx.x]) + inc - t;
    static_assert(SCAN_T == 256, "four waves");
    for (uint32_t w = 0; w < wave; w++) run += s_w[w];
#pragma unroll
    for (uint32_t k = 0; k < SCAN_PER; k++) {
        if (base + k < n) data[base + k] = run;
        run += v[k];
    }
}
// tmp == nullptr: *tmp_bytes = the scratch the scan of n_plus_1 entries needs
hipError_t launch_exclusive_scan_u64(uint64_t *data, uint64_t n_plus_1, void *tmp, size_t *tmp_bytes, hipStream_t s) {
    const uint64_t pieces = (n_plus_1 + SCAN_PIECE - 1) / SCAN_PIECE;
    if (pieces > 0x7FFFFFFFull) return hipErrorInvalidValue;
    if (!tmp) {
        *tmp_bytes = (size_t)(pieces + 1) * sizeof(uint64_t);
        return hipSuccess;
    }
    if (!n_plus_1) return hipSuccess;
    if (*tmp_bytes < (size_t)(pieces + 1) * sizeof(uint64_t)) return hipErrorInvalidValue;
    uint64_t *const sums = static_cast<uint64_t *>(tmp);
    hipLaunchKernelGGL(k_scan_sums, dim3((uint32_t)pieces), dim3(SCAN_T), 0, s, data, n_plus_1, sums);
    const char *const e = getenv("NGSQ_SCAN_OWN_CARRY"); // tests: the pieces up to which a block adds up its own carry (0: the three-launch path)
    if (pieces <= (e ? strtoull(e, nullptr, 10) : SCAN_OWN_CARRY)) {
        hipLaunchKernelGGL(k_scan_apply<true>, dim3((uint32_t)pieces), dim3(SCAN_T), 0, s, data, n_plus_1, sums);
    } else {
        hipLaunchKernelGGL(k_scan_of_sums, dim3(1), dim3(256), 0, s, sums, pieces);
        hipLaunchKernelGGL(k_scan_apply<false>, dim3((uint32_t)pieces), dim3(SCAN_T), 0, s, data, n_plus_1, sums);
    }
    return hipGetLastError();
}
hipError_t launch_rec_var(const uint8_t *raw, const uint64_t *var_base, uint64_t n, const RecColumns &c, uint64_t seq_bytes,
                          uint64_t qual_bytes, hipStream_t s) {
    if (n) {
        // fixed-pitch rows by destination dword (byte offsets must fit 32 bits), everything else 16 lanes per record
        const bool rows = !c.seq_off && (uint64_t)c.qual_pitch * n < ((uint64_t)1 << 32) - 64 && c.seq_pitch && c.qual_pitch;
        uint32_t parts = rows ? 0u : 2u;
        if (c.cigar_off) parts |= 1u;
        if (rows) {
            const uint64_t ds = ((uint64_t)c.seq_pitch * n + 3) / 4, dq = ((uint64_t)c.qual_pitch * n + 3) / 4;
            const uint64_t per_block = (uint64_t)PT * 4, cap_blocks = (uint64_t)(1u << 16) * (256 / PT);
            const dim3 gs((uint32_t)std::min<uint64_t>((ds + per_block - 1) / per_block, cap_blocks)), gq((uint32_t)std::min<uint64_t>((dq + per_block - 1) / per_block, cap_blocks));
            uint32_t *const seq32 = reinterpret_cast<uint32_t *>(c.seq), *const qual32 = reinterpret_cast<uint32_t *>(c.qual);
            if (c.seq_pitch >= 4) hipLaunchKernelGGL(k_rec_rows<false>, gs, dim3(PT), 0, s, raw, var_base + n, c.l_seq, n, seq32, c.seq_pitch, ds);
            else hipLaunchKernelGGL(k_rec_rows_narrow<false>, gs, dim3(256), 0, s, raw, var_base + n, c.l_seq, n, seq32, c.seq_pitch, ds);
            if (c.qual_pitch >= 4) hipLaunchKernelGGL(k_rec_rows<true>, gq, dim3(PT), 0, s, raw, var_base + n, c.l_seq, n, qual32, c.qual_pitch, dq);
            else hipLaunchKernelGGL(k_rec_rows_narrow<true>, gq, dim3(256), 0, s, raw, var_base + n, c.l_seq, n, qual32, c.qual_pitch, dq);
        }
        if (!c.cigar_off)
            hipLaunchKernelGGL(k_rec_cigar1, dim3((uint32_t)std::min<uint64_t>((n + PT - 1) / PT, 2048 * (256 / PT))), dim3(PT), 0, s, raw, var_base,
                               c.n_cigar, n, c.cigar);
        if (parts == 1u)
            hipLaunchKernelGGL(k_rec_cigar_var, dim3((uint32_t)((n * 4 + 255) / 256)), dim3(256), 0, s, raw, var_base, n, c.cigar_off, c.cigar);
        else if (parts)
            hipLaunchKernelGGL(k_rec_var, dim3((uint32_t)((n * 16 + 255) / 256)), dim3(256), 0, s, raw, var_base, n, c, parts);
    }
    hipLaunchKernelGGL(k_fill_slack, dim3(1), dim3(64), 0, s, c.seq + seq_bytes, c.qual + qual_bytes);
    return hipGetLastError();
}

} // namespace ngsq
