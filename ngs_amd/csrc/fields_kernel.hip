// fields_kernel.hip -- one pass over the fixed-width record columns and the CIGARs:
//   General flag tallies        general.rs:31-100
//   General CIGAR-op tallies    general.rs:103-121
//   Template Length             template_length.rs:79-87
//   Coverage range-add          coverage.rs:148-180 behind noodles' query() filter
// Reads every fixed-width column exactly once (25 B/record) plus 4 B per CIGAR op.
//
// Layout of the work: a block walks TILES of 1024 consecutive records; thread t
// owns records 4t..4t+3 of the tile, so each column is fetched with one vector
// load per thread (8 or 16 bytes) and a wave covers 256 consecutive records.
//
// Coverage is accumulated as a DIFFERENCE array (+1 at alignment_start, -1 at
// alignment_end+1, uint32 wrap-around; prefix-summed at teardown).  In a
// coordinate-sorted file the 1024 records of a tile start inside a few thousand
// positions, so every WAVE keeps its own LDS window of the reference axis anchored
// at the first of its 256 records: both updates of a record are LDS atomics, and
// the touched part of the window is then flushed by the same wave with coalesced
// global atomics (one wave instruction = 256 contiguous bytes).  Wave-private
// windows need no block barrier: the tile loop has none.  Records that do not fit
// the window (unsorted input, other sequence, long skips, very sparse files) fall
// back to direct global atomics -- always correct, just slower.
//
// STREAM (contexts created with sorted_input): Coverage is finished by cov_stream.hip from the
// (pos, cov_end) columns instead; this kernel then only applies noodles' query() filter, writes
// every record's exclusive alignment end into the scratch column st.cov_end, counts adjacent
// records that are out of coordinate order, and tracks the largest end per sequence / span per batch.
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace ngsq {

typedef unsigned long long u64;

#ifndef NGSQ_FT_PREFETCH_OFFS
#define NGSQ_FT_PREFETCH_OFFS 1 // 0: the offsets are loaded when the tile is processed (A/B builds)
#endif
constexpr uint32_t FT_THREADS = 256;
constexpr uint32_t FT_PER_THREAD = 4;
constexpr uint32_t FT_TILE = FT_THREADS * FT_PER_THREAD; // records per tile
constexpr uint32_t FT_WINDOW = 2048;                     // positions in one wave's LDS window
constexpr uint32_t FT_NSLOT = 16 + 3 + 18 + 1 + 1;       // block tallies: see slot_counter()

__device__ __forceinline__ uint32_t ft_wave_sum(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// One chunk-sum update per wave for the lanes that share the leader's chunk (the common
// case in a sorted file); the others add on their own.  `active` may be false in any lane.
__device__ __forceinline__ void ft_chunk_add(uint32_t *chunk_sums, bool active, uint32_t chunk, uint32_t val) {
    const unsigned long long m = __ballot(active);
    if (!m) return;
    const int leader = __ffsll((long long)m) - 1;
    const uint32_t c0 = __shfl(chunk, leader, 64);
    const bool same = active && chunk == c0;
    const uint32_t s = ft_wave_sum(same ? val : 0u);
    if ((threadIdx.x & 63) == 0 && s) atomicAdd(&chunk_sums[c0], s);
    if (active && !same) atomicAdd(&chunk_sums[chunk], val);
}

struct FieldsArgs {
    uint32_t do_general, do_tlen, do_cov;
};

// the raw columns of the four records of one thread
struct FtRaw {
    uint2 flag, ncig;
    uint32_t mapq;
    int4 ref, mate, tlen, pos;
    uint4 cig;
    int32_t prev_ref, prev_pos; // STREAM: the record in front of this thread's first one
    uint32_t co0;      // CIG_OFF: cigar_off of the thread's first record, less the tile's first (co_base); the others follow from n_cigar
    uint64_t co_base;
};

// number of lanes of the wave for which `c` holds: a wave-uniform value (SALU)
__device__ __forceinline__ uint32_t ft_count(bool c) { return (uint32_t)__popcll(__ballot(c)); }

__device__ __forceinline__ uint32_t slot_counter(uint32_t slot) {
    return slot < 16    ? C_GENERAL + slot
           : slot == 16 ? C_ERR + E_MISSING_REF
           : slot == 17 ? C_TLEN_PROCESSED
           : slot == 18 ? C_TLEN_IGNORED
           : slot < 28  ? C_CIGAR1 + (slot - 19)
           : slot < 37  ? C_CIGAR2 + (slot - 28)
           : slot == 37 ? C_ERR + E_BAD_CIGAR
                        : C_COV_UNSORTED;
}

// CIG_OFF: CIGARs addressed through cigar_off (a compile-time choice: a conditional load in the
// tile loop would make hipcc drain the in-order vmcnt queue, i.e. the prefetch, on every tile)
template <bool CIG_OFF, bool STREAM>
__global__ __launch_bounds__(FT_THREADS, 4) void k_fields(DeviceState st, DeviceBatch b, FieldsArgs a) {
    NGSQ_FOREGROUND_WAVE();
    extern __shared__ uint32_t s_dyn[];
    uint32_t *const s_tlen = s_dyn;                                 // tlen_cap + 1
    uint32_t *const s_win = s_dyn + ((st.tlen_cap + 1 + 3) & ~3u);  // one FT_WINDOW per wave
    __shared__ u64 s_acc[FT_NSLOT];
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    for (uint32_t i = tid; i <= st.tlen_cap; i += FT_THREADS) s_tlen[i] = 0;
    if (a.do_cov && !STREAM)
        for (uint32_t i = tid; i < (FT_THREADS / 64) * FT_WINDOW; i += FT_THREADS) s_win[i] = 0;
    if (tid < FT_NSLOT) s_acc[tid] = 0;
    __syncthreads();

    // wave-uniform tallies (ballot + popcount): 16 RecordMetrics, missing ref, tlen processed/ignored
    uint32_t g[20]; // [19]: STREAM, records out of coordinate order
#pragma unroll
    for (int k = 0; k < 20; k++) g[k] = 0;
    int32_t run_ref = -1; // STREAM: largest end on the sequence this thread is currently in
    uint32_t run_end = 0, max_span = 0;
    uint32_t one[9], two[9]; // per-lane CIGAR-op tallies
#pragma unroll
    for (int k = 0; k < 9; k++) one[k] = two[k] = 0;
    // ... accumulated as nine 7-bit fields of a 64-bit word each (field q = op code q): one shift and two adds per
    // op instead of eighteen compare-and-adds; unpacked into one[] / two[] before a field can wrap
    u64 pk1 = 0, pk2 = 0;
    uint32_t pk_n = 0; // ops added to the packed words since they were last unpacked
    auto unpack_ops = [&]() {
#pragma unroll
        for (int q = 0; q < 9; q++) {
            one[q] += (uint32_t)(pk1 >> (7 * q)) & 127u;
            two[q] += (uint32_t)(pk2 >> (7 * q)) & 127u;
        }
        pk1 = pk2 = 0;
        pk_n = 0;
    };
    uint32_t bad_op = 0;
    u64 nonsensical = 0;
    int32_t seen_ref = -1; // run-length tally of records Coverage processed, per sequence
    uint32_t seen_cnt = 0;
    u64 t_lo = ~0ull, t_hi = 0; // entries of the depth block this thread has written: [t_lo, t_hi)

    const uint64_t n_tiles = (b.n + FT_TILE - 1) / FT_TILE;
    const bool cigar_vec = !CIG_OFF && b.cigar_stride == 1;

    auto load_tile = [&](uint64_t tile) -> FtRaw {
        // full tiles only: every column with one aligned vector load per thread
        const uint64_t r0 = tile * FT_TILE + (uint64_t)tid * FT_PER_THREAD;
        FtRaw r;
        r.flag = *reinterpret_cast<const uint2 *>(b.flag + r0);
        r.ncig = make_uint2(0, 0);
        r.mapq = 0;
        r.ref = r.mate = r.tlen = r.pos = make_int4(0, 0, 0, 0);
        r.cig = make_uint4(0, 0, 0, 0);
        if (a.do_general || a.do_cov) {
            r.ref = *reinterpret_cast<const int4 *>(b.ref_id + r0);
            r.ncig = *reinterpret_cast<const uint2 *>(b.n_cigar + r0);
            if (cigar_vec) r.cig = *reinterpret_cast<const uint4 *>(b.cigar + r0);
        }
        if (a.do_general) {
            r.mapq = *reinterpret_cast<const uint32_t *>(b.mapq + r0);
            r.mate = *reinterpret_cast<const int4 *>(b.mate_ref_id + r0);
        }
        if (a.do_tlen) r.tlen = *reinterpret_cast<const int4 *>(b.tlen + r0);
        if (a.do_cov) r.pos = *reinterpret_cast<const int4 *>(b.pos + r0);
        r.co0 = 0;
        r.co_base = 0;
        if (CIG_OFF && NGSQ_FT_PREFETCH_OFFS && (a.do_general || a.do_cov)) {
            // Round 5: the thread's FIRST offset comes with the tile (a tile ahead), as a 32-bit distance from the tile's first (a scalar
            // load); the other three follow from the n_cigar column, which is here anyway -- so the operations are requested at the
            // top of process() instead of behind a load of their own.  (All five offsets prefetched cost ten registers held across
            // the tile loop, six of them spilled: 1.26 ms against 1.04.)
            r.co_base = b.cigar_off[tile * FT_TILE];
            r.co0 = (uint32_t)(b.cigar_off[r0] - r.co_base);
        }
        r.prev_ref = r.prev_pos = 0;
        if (STREAM) { // the record in front of the WAVE's first one: a wave-uniform address, i.e. a scalar load;
                      // the other lanes take their predecessor from the neighbouring lane (process())
            const uint64_t w0 = tile * FT_TILE + (uint64_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6)) * 256;
            const uint64_t pi = w0 ? w0 - 1 : 0;
            r.prev_ref = b.ref_id[pi];
            r.prev_pos = b.pos[pi];
        }
        return r;
    };
    auto load_tail = [&](uint64_t tile, uint32_t &nrec) -> FtRaw {
        // the last, partial tile: scalar loads, absent records read as zeros
        const uint64_t r0 = tile * FT_TILE + (uint64_t)tid * FT_PER_THREAD;
        uint32_t flag[4] = {0, 0, 0, 0}, ncig[4] = {0, 0, 0, 0}, mapq[4] = {0, 0, 0, 0}, cig[4] = {0, 0, 0, 0};
        int32_t ref[4] = {0, 0, 0, 0}, mate[4] = {0, 0, 0, 0}, tlen[4] = {0, 0, 0, 0}, pos[4] = {0, 0, 0, 0};
        nrec = 0;
        for (uint32_t j = 0; j < FT_PER_THREAD; j++) {
            if (r0 + j >= b.n) break;
            nrec = j + 1;
            flag[j] = b.flag[r0 + j];
            if (a.do_general || a.do_cov) {
                ref[j] = b.ref_id[r0 + j];
                ncig[j] = b.n_cigar[r0 + j];
                if (cigar_vec) cig[j] = b.cigar[r0 + j];
            }
            if (a.do_general) {
                mapq[j] = b.mapq[r0 + j];
                mate[j] = b.mate_ref_id[r0 + j];
            }
            if (a.do_tlen) tlen[j] = b.tlen[r0 + j];
            if (a.do_cov) pos[j] = b.pos[r0 + j];
        }
        FtRaw r;
        r.flag = make_uint2(flag[0] | (flag[1] << 16), flag[2] | (flag[3] << 16));
        r.ncig = make_uint2(ncig[0] | (ncig[1] << 16), ncig[2] | (ncig[3] << 16));
        r.mapq = mapq[0] | (mapq[1] << 8) | (mapq[2] << 16) | (mapq[3] << 24);
        r.ref = make_int4(ref[0], ref[1], ref[2], ref[3]);
        r.mate = make_int4(mate[0], mate[1], mate[2], mate[3]);
        r.tlen = make_int4(tlen[0], tlen[1], tlen[2], tlen[3]);
        r.pos = make_int4(pos[0], pos[1], pos[2], pos[3]);
        r.cig = make_uint4(cig[0], cig[1], cig[2], cig[3]);
        r.co0 = 0;
        r.co_base = 0;
        r.prev_ref = r.prev_pos = 0;
        if (STREAM && nrec) {
            const uint64_t pi = r0 ? r0 - 1 : 0;
            r.prev_ref = b.ref_id[pi];
            r.prev_pos = b.pos[pi];
        }
        return r;
    };

    // facts of the sequence the window was last anchored on (reloaded only when it changes)
    int32_t meta_ref = -1;
    uint64_t meta_off = NO_DEPTH, meta_L = 0;
    auto process = [&](const FtRaw &raw, uint64_t tile, uint32_t nrec, bool full_tile) {
        const uint64_t r0 = tile * FT_TILE + (uint64_t)tid * FT_PER_THREAD;
        const uint32_t flag[4] = {raw.flag.x & 0xFFFFu, raw.flag.x >> 16, raw.flag.y & 0xFFFFu, raw.flag.y >> 16};
        const uint32_t ncig[4] = {raw.ncig.x & 0xFFFFu, raw.ncig.x >> 16, raw.ncig.y & 0xFFFFu, raw.ncig.y >> 16};
        const int32_t ref[4] = {raw.ref.x, raw.ref.y, raw.ref.z, raw.ref.w};
        const int32_t mate[4] = {raw.mate.x, raw.mate.y, raw.mate.z, raw.mate.w};
        const int32_t tlen[4] = {raw.tlen.x, raw.tlen.y, raw.tlen.z, raw.tlen.w};
        const int32_t pos[4] = {raw.pos.x, raw.pos.y, raw.pos.z, raw.pos.w};
        const uint32_t cig1[4] = {raw.cig.x, raw.cig.y, raw.cig.z, raw.cig.w};

        // ---- coverage window of this wave: anchored at the wave's first record (lane 0 holds it).
        // Not placed / not a covered sequence => no window for this round.
        uint32_t *const win = s_win + (tid >> 6) * FT_WINDOW;
        int32_t win_ref = -1;
        uint32_t win_base = 0;
        uint64_t win_off = NO_DEPTH, win_L = 0;
        if (a.do_cov) {
            const int32_t fr = __builtin_amdgcn_readfirstlane(ref[0]), fp = __builtin_amdgcn_readfirstlane(pos[0]);
            if (fr >= 0 && (uint32_t)fr < st.n_refs && fp >= 0) {
                if (fr != meta_ref) { // wave-uniform branch; the loaded facts leave it in scalar registers,
                                      // so the join needs no vmcnt wait (which would drain the prefetch)
                    const uint64_t o = st.ref_depth_off[fr];
                    const uint32_t l = st.ref_len[fr];
                    meta_ref = fr;
                    meta_off = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(o >> 32)) << 32) |
                               (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)o);
                    meta_L = (uint32_t)__builtin_amdgcn_readfirstlane((int)l);
                }
                win_off = meta_off;
                win_L = meta_L;
                if (win_off != NO_DEPTH || STREAM) {
                    win_ref = fr;
                    win_base = (uint32_t)fp & ~3u; // <= alignment_start of that record
                }
            }
        }
        // ---- the per-sequence run-length tallies of the lanes (`seen`, STREAM: the largest end) are flushed HERE, once per wave,
        // when the wave has moved on to another sequence.  A block strides through the file a grid's worth of tiles at a time: on a
        // real header (195 sequences) nearly every tile of a block lies on another sequence than its last one, and until round 6
        // every LANE then flushed its run with an atomic of its own on the same few words -- 256 same-address atomics per tile
        // and block: 42 ms for 20 M records spread over the GRCh38 sequences, against 0.16 ms for the same records on chr1 alone
        // (bench.py whole_genome; no test or leg had more than four sequences before).
        if (a.do_cov) {
            const int32_t fr = __builtin_amdgcn_readfirstlane(ref[0]);
            const bool stale = seen_cnt != 0 && seen_ref != fr;
            const u64 sm = __ballot(stale);
            if (sm) {
                const int32_t rr = __shfl(seen_ref, __ffsll((long long)sm) - 1, 64);
                const bool same = stale && seen_ref == rr;
                const uint32_t sum = ft_wave_sum(same ? seen_cnt : 0u);
                if (lane == 0 && sum) atomicAdd(&st.counters[st.off_seen + rr], (u64)sum);
                if (stale && !same) atomicAdd(&st.counters[st.off_seen + seen_ref], (u64)seen_cnt);
                if (stale) seen_cnt = 0;
            }
            if (STREAM) {
                const bool stale_e = run_end != 0 && run_ref != fr;
                const u64 em = __ballot(stale_e);
                if (em) {
                    const int32_t rr = __shfl(run_ref, __ffsll((long long)em) - 1, 64);
                    const bool same = stale_e && run_ref == rr;
                    uint32_t m = same ? run_end : 0u;
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor(m, o, 64));
                    if (lane == 0 && m) atomicMax(&st.end_acc[rr], m);
                    if (stale_e && !same) atomicMax(&st.end_acc[run_ref], run_end);
                    if (stale_e) run_end = 0;
                }
            }
        }
        uint32_t my_max = 0;
        uint32_t cend[4] = {0, 0, 0, 0}; // STREAM: exclusive alignment end of each record (0: covers nothing)
        if (STREAM) { // coordinate order of adjacent records; unplaced (-1) sorts last, as in a sorted BAM
            auto key = [](int32_t rf, int32_t ps) -> u64 { // st_key() of cov_stream.hip
                return rf < 0 ? 0xFFFFFFFF00000000ull : ((u64)(uint32_t)rf << 32) | (uint32_t)(ps + 1);
            };
            u64 kp = key(raw.prev_ref, raw.prev_pos);
            if (full_tile) { // raw.prev_* is the wave's predecessor: lanes 1.. look at the lane in front
                const int32_t nr = __shfl_up(ref[3], 1, 64), np_ = __shfl_up(pos[3], 1, 64);
                if (lane) kp = key(nr, np_);
            }
#pragma unroll
            for (uint32_t j = 0; j < FT_PER_THREAD; j++) {
                const u64 kj = key(ref[j], pos[j]);
                g[19] += ft_count(j < nrec && kj < kp);
                kp = kj;
            }
        }

        // CIG_OFF: the four records' offsets into the CIGAR column and their first operations are requested here, together, and
        // arrive while the flag tallies below are made.  (Until round 4 each record's offset was loaded when its walk began and
        // its first operation behind that: eight dependent latencies per thread and tile; `k_fields<true, *>` took twice the
        // time of the fixed-pitch variants on the 50-300 bp workload.)
        // Round 5: the SECOND and THIRD operations come with the first (a record of an aligner's file has one to three; the walk below
        // then loads nothing for them -- it used to fetch every operation behind the first when it got there, one dependent latency
        // per operation and record, with the other lanes of the wave waiting), and a full tile's offsets have come with the tile
        // (load_tile).
        uint64_t coff[FT_PER_THREAD + 1] = {0, 0, 0, 0, 0};
        uint32_t cigf[FT_PER_THREAD] = {0, 0, 0, 0}, cig2[FT_PER_THREAD] = {0, 0, 0, 0}, cig3[FT_PER_THREAD] = {0, 0, 0, 0};
        if (CIG_OFF && (a.do_general || a.do_cov)) {
            const bool saturated = ncig[0] == 0xFFFFu || ncig[1] == 0xFFFFu || ncig[2] == 0xFFFFu || ncig[3] == 0xFFFFu; // (the count is in the offsets then)
            if (full_tile && NGSQ_FT_PREFETCH_OFFS && !saturated) {
                coff[0] = raw.co_base + raw.co0;
#pragma unroll
                for (uint32_t j = 0; j < FT_PER_THREAD; j++) coff[j + 1] = coff[j] + ncig[j];
            } else if (full_tile) {
                ulonglong2 c01, c23;
                __builtin_memcpy(&c01, b.cigar_off + r0, 16);
                __builtin_memcpy(&c23, b.cigar_off + r0 + 2, 16);
                coff[0] = c01.x, coff[1] = c01.y, coff[2] = c23.x, coff[3] = c23.y;
                coff[4] = b.cigar_off[r0 + 4];
            } else {
#pragma unroll
                for (uint32_t j = 0; j <= FT_PER_THREAD; j++)
                    if (j <= nrec && nrec) coff[j] = b.cigar_off[r0 + j];
            }
#pragma unroll
            for (uint32_t j = 0; j < FT_PER_THREAD; j++) {
                const uint64_t nj = j < nrec ? coff[j + 1] - coff[j] : 0;
                if (nj > 0) cigf[j] = b.cigar[coff[j]];
                if (nj > 1) cig2[j] = b.cigar[coff[j] + 1];
                if (nj > 2) cig3[j] = b.cigar[coff[j] + 2];
            }
        }
#pragma unroll
        for (uint32_t j = 0; j < FT_PER_THREAD; j++) {
            const bool live = j < nrec;
            bool fb = false;          // this record went straight to the arrays (outside the window)
            uint32_t fb_c0 = 0, fb_c1 = 0;
            const uint32_t f = flag[j];
            if (a.do_general) {
                // general.rs:31-100 as mask tests; every tally is a ballot popcount (SALU)
                const bool prim = live && !(f & 0x900u);             // neither secondary nor supplementary
                const bool pair = prim && (f & 0x1u);                // :60
                const bool pmap = pair && !(f & 0x4u);               // :71
                const bool mm = pmap && !(f & 0x8u);                 // :79 mate mapped
                const bool noid = mm && (ref[j] < 0 || mate[j] < 0); // :81-83 unwrap() on None
                const bool mis = mm && !noid && ref[j] != mate[j];   // :85-86
                const uint32_t mq = (raw.mapq >> (8 * j)) & 0xFFu;
                g[0] += ft_count(live);                        // total :33
                g[1] += ft_count(live && (f & 0x4u));          // unmapped :37-39
                g[2] += ft_count(live && (f & 0x400u));        // duplicate :41-43
                g[3] += ft_count(prim);                        // :50
                g[4] += ft_count(live && (f & 0x100u));        // secondary :45-46
                g[5] += ft_count(live && (f & 0x900u) == 0x800u); // supplementary :47-48
                g[6] += ft_count(prim && !(f & 0x4u));         // primary_mapped :52-54
                g[7] += ft_count(prim && (f & 0x400u));        // primary_duplicate :56-58
                g[8] += ft_count(pair);                        // paired
                g[9] += ft_count(pair && (f & 0x40u));         // read_1 :63-65
                g[10] += ft_count(pair && (f & 0x80u));        // read_2 :67-69
                g[11] += ft_count(pmap && (f & 0x2u));         // proper_pair :72-74
                g[12] += ft_count(pmap && (f & 0x8u));         // singleton :76-77
                g[13] += ft_count(mm);                         // mate_mapped
                g[14] += ft_count(mis);                        // mismatch
                g[15] += ft_count(mis && mq >= 5u);            // :88-95 (255 = missing counts as HQ)
                g[16] += ft_count(noid);
            }
            if (a.do_tlen) {
                // template_length.rs:80  `as usize`: negatives wrap above any capacity
                const int32_t t = tlen[j];
                const bool inr = live && t >= 0 && (uint32_t)t <= st.tlen_cap;
                if (inr) atomicAdd(&s_tlen[t], 1u);
                g[17] += ft_count(inr);
                g[18] += ft_count(live && !inr);
            }
            if ((a.do_general || a.do_cov) && live) {
                // ---- CIGAR walk: op tallies (general.rs:103-121) and the alignment span
                uint32_t n_ops = ncig[j];
                const uint32_t r1 = (f >> 6) & 1u; // first segment -> "read one"
                uint64_t span = 0;
                uint64_t cbase = (r0 + j) * (uint64_t)b.cigar_stride;
                if (CIG_OFF) {
                    cbase = coff[j];
                    if (n_ops == 0xFFFFu) n_ops = (uint32_t)(coff[j + 1] - cbase); // (65535 = "or more": include/ngsq.h)
                }
                for (uint32_t k = 0; k < n_ops; k++) {
                    const uint32_t cg = (cigar_vec && k == 0) ? cig1[j]
                                        : (CIG_OFF && k < 3) ? (k == 0 ? cigf[j] : k == 1 ? cig2[j] : cig3[j])
                                                             : b.cigar[cbase + k];
                    const uint32_t op = cg & 0xFu, len = cg >> 4;
                    if (op > 8u) {
                        bad_op += 1;
                        continue;
                    }
                    if ((0x18Du >> op) & 1u) span += len; // utils/cigar.rs:6-11  M D N = X
                    if (a.do_general) {
                        const u64 inc = 1ull << (7u * op);
                        pk1 += r1 ? inc : 0ull;
                        pk2 += r1 ? 0ull : inc;
                        if (++pk_n == 127u) unpack_ops(); // per lane; a record can hold more ops than a field counts
                    }
                }
                if (a.do_cov) {
                    const int32_t rf = ref[j];
                    const int32_t ps = pos[j];
                    // one record's range-add; `in_seq` = it lies on the window's sequence
                    auto cover = [&](uint64_t off, uint64_t L, bool in_seq) {
                        const uint64_t s = (uint64_t)ps + 1, e = s + span - 1;
                        // noodles query(): alignment_end must be Some (>= 1) and [s,e] must meet [1,L]
                        if (off == NO_DEPTH || e == 0 || s > L) return;
                        if (rf != seen_ref) {
                            if (seen_cnt) atomicAdd(&st.counters[st.off_seen + seen_ref], (u64)seen_cnt);
                            seen_ref = rf;
                            seen_cnt = 0;
                        }
                        seen_cnt += 1;
                        const uint64_t ec = e < L ? e : L;
                        nonsensical += e - ec; // coverage.rs:163-176: one per position > L
                        if (s > ec) return;
                        if (STREAM) {
                            cend[j] = (uint32_t)ec + 1;
                            if (rf != run_ref) {
                                if (run_end) atomicMax(&st.end_acc[run_ref], run_end);
                                run_ref = rf;
                                run_end = 0;
                            }
                            run_end = max(run_end, (uint32_t)ec + 1);
                            max_span = max(max_span, (uint32_t)(ec + 1 - s));
                            return;
                        }
                        const uint64_t i0 = s - win_base, i1 = ec + 1 - win_base;
                        if (in_seq && s >= win_base && i1 < FT_WINDOW) {
                            atomicAdd(&win[i0], 1u);
                            atomicAdd(&win[i1], 0xFFFFFFFFu);
                            my_max = max(my_max, (uint32_t)i1);
                        } else { // outside the window: straight to the arrays
                            const uint64_t g0 = off + s, g1 = off + ec + 1;
                            atomicAdd(&st.depth[g0], 1u);
                            atomicAdd(&st.depth[g1], 0xFFFFFFFFu);
                            fb = true;
                            t_lo = g0 < t_lo ? g0 : t_lo;
                            t_hi = g1 + 1 > t_hi ? g1 + 1 : t_hi;
                            fb_c0 = (uint32_t)(g0 / COV_CHUNK);
                            fb_c1 = (uint32_t)(g1 / COV_CHUNK);
                        }
                    };
                    if (rf >= 0 && (uint32_t)rf < st.n_refs && ps >= 0) {
                        if (rf == win_ref) {
                            cover(win_off, win_L, true);
                        } else { // another sequence: its facts are loaded and consumed inside this branch
                            cover(st.ref_depth_off[rf], st.ref_len[rf], false);
                        }
                    }
                }
            }
            if (a.do_cov && !STREAM) { // chunk sums of the records that bypassed the window (wave-aggregated)
                ft_chunk_add(st.chunk_sums, fb, fb_c0, 1u);
                ft_chunk_add(st.chunk_sums, fb, fb_c1, 0xFFFFFFFFu);
            }
        }

        if (STREAM) {
            if (nrec == FT_PER_THREAD) {
                *reinterpret_cast<uint4 *>(st.cov_end + r0) = make_uint4(cend[0], cend[1], cend[2], cend[3]);
            } else {
                for (uint32_t j = 0; j < nrec; j++) st.cov_end[r0 + j] = cend[j];
            }
        }
        if (a.do_cov && !STREAM) {
            // ---- the wave flushes the touched part of its own window with coalesced global
            // atomics and leaves it zeroed.  No barrier: LDS operations of one wave execute in order.
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) my_max = max(my_max, (uint32_t)__shfl_xor(my_max, o, 64));
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const uint32_t top = my_max;
            if (win_ref >= 0 && top) {
                const uint64_t goff = win_off + win_base; // element index of window entry 0
                uint32_t *dst = st.depth + goff;
                t_lo = goff < t_lo ? goff : t_lo;
                t_hi = goff + top + 1 > t_hi ? goff + top + 1 : t_hi;
                // running sum per scan chunk (cov_scan.hip): the window spans at most two chunks
                const uint32_t c0 = (uint32_t)(goff / COV_CHUNK);
                const uint32_t split = (uint32_t)((uint64_t)(c0 + 1) * COV_CHUNK - goff); // first entry of chunk c0+1
                uint32_t sa = 0, sb = 0;
                // four independent LDS reads in flight per pass (entries past `top` are zero)
                for (uint32_t ib = 0; ib <= top; ib += 256) {
                    uint32_t v[4];
#pragma unroll
                    for (uint32_t k = 0; k < 4; k++) {
                        const uint32_t i = ib + 64 * k + lane;
                        v[k] = i < FT_WINDOW ? win[i] : 0u;
                    }
#pragma unroll
                    for (uint32_t k = 0; k < 4; k++) {
                        const uint32_t i = ib + 64 * k + lane;
                        if (v[k]) {
                            atomicAdd(&dst[i], v[k]);
                            win[i] = 0;
                            if (i < split) sa += v[k]; else sb += v[k];
                        }
                    }
                }
                sa = ft_wave_sum(sa);
                sb = ft_wave_sum(sb);
                if (lane == 0) {
                    if (sa) atomicAdd(&st.chunk_sums[c0], sa);
                    if (sb) atomicAdd(&st.chunk_sums[c0 + 1], sb);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
    };

    const uint64_t n_full = b.n / FT_TILE; // tiles with all 1024 records
    uint64_t tile = blockIdx.x;
    if (tile < n_full) {
        FtRaw cur = load_tile(tile);
        while (tile < n_full) {
            const uint64_t nt = tile + gridDim.x;
            // branch-free prefetch of this block's next full tile (past the end: re-read this one)
            const FtRaw nxt = load_tile(nt < n_full ? nt : tile);
            process(cur, tile, FT_PER_THREAD, true);
            cur = nxt;
            tile = nt;
        }
    }
    if (tile == n_full && n_full < n_tiles) { // the partial tile belongs to exactly one block
        uint32_t nrec = 0;
        const FtRaw tail = load_tail(tile, nrec);
        process(tail, tile, nrec, false);
    }

    // ---- block epilogue
    { // `seen`: one atomic per wave when the wave saw a single sequence
        const u64 act = __ballot(seen_cnt != 0);
        if (act) {
            const int leader = __ffsll((long long)act) - 1;
            const int32_t rr = __shfl(seen_ref, leader, 64);
            const bool same = seen_cnt != 0 && seen_ref == rr;
            const uint32_t sum = ft_wave_sum(same ? seen_cnt : 0u);
            if (lane == 0) atomicAdd(&st.counters[st.off_seen + rr], (u64)sum);
            if (seen_cnt != 0 && !same) atomicAdd(&st.counters[st.off_seen + seen_ref], (u64)seen_cnt);
        }
    }
    if (STREAM) { // one atomic per block: a quarter of a million same-address atomics would take milliseconds
        __shared__ int32_t s_end_ref;
        __shared__ uint32_t s_end_max, s_span_max;
        if (tid == 0) {
            s_end_ref = -1;
            s_end_max = 0;
            s_span_max = 0;
        }
        __syncthreads();
        { // the block's sequence: that of the first thread (of any wave) that covered something
            const u64 act = __ballot(run_end != 0);
            const int32_t wr = __shfl(run_ref, act ? __ffsll((long long)act) - 1 : 0, 64);
            if (act && lane == 0) atomicCAS(&s_end_ref, -1, wr);
        }
        __syncthreads();
        const int32_t lead = s_end_ref;
        const bool same = run_end != 0 && run_ref == lead;
        uint32_t m = same ? run_end : 0u;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            m = max(m, (uint32_t)__shfl_xor(m, o, 64));
            max_span = max(max_span, (uint32_t)__shfl_xor(max_span, o, 64));
        }
        if (lane == 0 && m) atomicMax(&s_end_max, m);
        if (lane == 0 && max_span) atomicMax(&s_span_max, max_span);
        if (run_end != 0 && !same) atomicMax(&st.end_acc[run_ref], run_end); // the block crossed a sequence boundary
        __syncthreads();
        if (tid == 0) {
            if (s_end_max) atomicMax(&st.end_acc[lead], s_end_max);
            if (s_span_max) atomicMax(st.batch_span, s_span_max);
        }
    }
    if (a.do_cov && !STREAM) { // written range of the depth block (min / max over the wave, then one atomic each)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const u64 l2 = __shfl_xor(t_lo, o, 64), h2 = __shfl_xor(t_hi, o, 64);
            t_lo = l2 < t_lo ? l2 : t_lo;
            t_hi = h2 > t_hi ? h2 : t_hi;
        }
        if (lane == 0 && t_hi > 0) {
            atomicMin(&st.touched[0], t_lo);
            atomicMax(&st.touched[1], t_hi);
        }
    }
    {
        u64 ns = nonsensical;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ns += __shfl_down(ns, o, 64);
        if (lane == 0 && ns) atomicAdd(&st.counters[C_COV_NONSENSICAL], ns);
    }
    // wave-uniform tallies: lane 0 of each wave holds the wave's total already
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 19; k++)
            if (g[k]) atomicAdd(&s_acc[k], (u64)g[k]);
        if (g[19]) atomicAdd(&s_acc[38], (u64)g[19]);
    }
    unpack_ops();
#pragma unroll
    for (int k = 0; k < 9; k++) {
        const uint32_t r1 = ft_wave_sum(one[k]), r2 = ft_wave_sum(two[k]);
        if (lane == 0 && r1) atomicAdd(&s_acc[19 + k], (u64)r1);
        if (lane == 0 && r2) atomicAdd(&s_acc[28 + k], (u64)r2);
    }
    {
        const uint32_t r = ft_wave_sum(bad_op);
        if (lane == 0 && r) atomicAdd(&s_acc[37], (u64)r);
    }
    __syncthreads();
    for (uint32_t i = tid; i <= st.tlen_cap; i += FT_THREADS) {
        const uint32_t v = s_tlen[i];
        if (v) atomicAdd(&st.counters[st.off_tlen + i], (u64)v);
    }
    if (tid < FT_NSLOT) {
        const u64 r = s_acc[tid];
        if (r) atomicAdd(&st.counters[slot_counter(tid)], r);
    }
}

hipError_t launch_fields(const LaunchInfo &li, const DeviceState &st, const DeviceBatch &b, uint32_t rec_facets,
                         int coverage, hipStream_t s) {
    if (!b.n) return hipSuccess;
    FieldsArgs a;
    a.do_general = (rec_facets & NGSQ_FACET_GENERAL) ? 1 : 0;
    a.do_tlen = (rec_facets & NGSQ_FACET_TEMPLATE_LENGTH) ? 1 : 0;
    a.do_cov = coverage ? 1 : 0;
    const bool stream = coverage == 2;
    const size_t lds = (((size_t)st.tlen_cap + 1 + 3) & ~(size_t)3) * 4 + (coverage == 1 ? (FT_THREADS / 64) * FT_WINDOW * 4 : 0);
    static bool attr = false;
    if (!attr) {
        const void *fns[4] = {reinterpret_cast<const void *>(k_fields<false, false>), reinterpret_cast<const void *>(k_fields<true, false>),
                              reinterpret_cast<const void *>(k_fields<false, true>), reinterpret_cast<const void *>(k_fields<true, true>)};
        for (const void *f : fns) {
            hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
            if (e != hipSuccess) return e;
        }
        attr = true;
    }
    uint64_t g = (b.n + FT_TILE - 1) / FT_TILE;
    const uint64_t cap = (uint64_t)li.n_cu * 4;
    if (g > cap) g = cap;
    if (b.cigar_off && stream)
        hipLaunchKernelGGL((k_fields<true, true>), dim3((uint32_t)g), dim3(FT_THREADS), lds, s, st, b, a);
    else if (b.cigar_off)
        hipLaunchKernelGGL((k_fields<true, false>), dim3((uint32_t)g), dim3(FT_THREADS), lds, s, st, b, a);
    else if (stream)
        hipLaunchKernelGGL((k_fields<false, true>), dim3((uint32_t)g), dim3(FT_THREADS), lds, s, st, b, a);
    else
        hipLaunchKernelGGL((k_fields<false, false>), dim3((uint32_t)g), dim3(FT_THREADS), lds, s, st, b, a);
    return hipGetLastError();
}

} // namespace ngsq
