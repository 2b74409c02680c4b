// ingest_kernels.h -- device ingest: BGZF inflate and BAM record
// parse -> structure-of-arrays columns on the GPU.  Launchers only; no torch, no host I/O.
#pragma once

#include <hip/hip_runtime_api.h>
#include <stdint.h>

// The kernels of the context's stream (record index, columns, facets) share the CUs with the BGZF decoders of the next chunk
// (another stream, persistent waves that keep the scalar and vector issue ports busy): their waves ask for the highest
// issue priority, or a latency-bound kernel like the record-chain walk runs ten times slower beside the decoders than alone.
#ifndef NGSQ_FOREGROUND_WAVE
#define NGSQ_FOREGROUND_WAVE() __builtin_amdgcn_s_setprio(3)
#endif

// Every vector-memory operation this wave has issued (loads, stores, atomics without a result) has been acknowledged by the L2
// when this returns: orders device-scope atomics in front of a later one without a release fence (k_rec_fixed's ticket).
#ifndef NGSQ_WAIT_VMEM
#define NGSQ_WAIT_VMEM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#endif

namespace ngsq {

// one BGZF block (SAM/BAM specification 4.1): the raw DEFLATE payload of one gzip member
struct BgzfBlock {
    uint64_t in_off;  // byte offset of the DEFLATE payload in the compressed buffer
    uint64_t out_off; // byte offset of the block's data in the decompressed buffer
    uint32_t in_len;  // payload bytes (BSIZE + 1 - header - 8)
    uint32_t isize;   // ISIZE: decompressed bytes (<= 65536)
    uint32_t crc;     // CRC32 of the decompressed bytes
    uint32_t pad;
};

// per-block result codes written by the inflate kernel
enum InflateStatus : uint32_t {
    INF_OK = 0,
    INF_BAD_BLOCK_TYPE = 1,
    INF_BAD_STORED_LEN = 2,
    INF_BAD_CODE_LENGTHS = 3,
    INF_BAD_SYMBOL = 4,
    INF_BAD_DISTANCE = 5,
    INF_OUTPUT_OVERRUN = 6,
    INF_INPUT_OVERRUN = 7,
    INF_SIZE_MISMATCH = 8,
    INF_CRC_MISMATCH = 9,
};

// The compressed buffer must be readable for INFLATE_IN_SLACK bytes past the last payload.
constexpr uint32_t INFLATE_IN_SLACK = 1024;

// Inflate blocks [0, n_blocks): one wavefront per block at a time, a resident grid of decoders taking block after block
// from *counter (device memory, 4 bytes, zeroed by the launcher on the stream).  status[k] = InflateStatus of block k.
// counter_base (optional, host): the value *counter holds when this launch begins -- a caller that keeps one counter word for all
// its launches on a stream passes it and gets the value behind this launch back (every decoder leaves the counter one past the
// blocks: n_blocks + the grid); no memset per launch then.  NULL: *counter is zeroed on the stream first.
hipError_t launch_bgzf_inflate(const uint8_t *comp, const BgzfBlock *blocks, uint32_t n_blocks, uint8_t *out,
                               uint32_t *status, uint32_t *counter, bool check_crc, hipStream_t s, uint32_t *counter_base = nullptr);

// CRC32 of every block's inflated bytes against its gzip trailer (status[k] = INF_CRC_MISMATCH); what
// launch_bgzf_inflate(check_crc = true) runs second
// status_host (optional): pinned host memory the device addresses; receives every block's final status (so that the caller
// needs no copy behind the kernel)
hipError_t launch_bgzf_crc(const BgzfBlock *blocks, uint32_t n_blocks, const uint8_t *out, uint32_t *status, uint32_t *status_host, hipStream_t s);

// ---- BAM record parse (csrc/bam_device.hip) -------------------------------------------------------
#ifndef NGSQ_REC_SEGMENT
#define NGSQ_REC_SEGMENT 16384
#endif
constexpr uint64_t REC_SEGMENT = NGSQ_REC_SEGMENT; // bytes of the inflated stream whose record chain is found as one (by its own lanes of a wave)
constexpr uint32_t REC_GROUP = 4;        // segments per wave of k_rec_candidates
constexpr uint32_t REC_CANDIDATES = 2;   // chain starts kept per segment
constexpr uint32_t REC_PIECES = (uint32_t)(REC_SEGMENT / 4096); // a segment's chain is written out by that many lanes, 4 KiB each
constexpr uint64_t REC_PIECE = REC_SEGMENT / REC_PIECES;

struct RecCandidate {
    uint64_t start;   // offset of a plausible record start inside the segment
    uint64_t landing; // where its chain leaves the segment (offset of a record start, or of the cut tail)
    uint32_t count;   // records on the chain inside the segment
    uint32_t valid;
};
// a candidate's chain per 4 KiB piece of the segment (stays on the device): the first record start of the chain at or
// behind the piece's start, relative to the segment's start (0xFFFFFFFF: the chain has ended before), and the number of
// records of the chain in front of it
struct RecPieces {
    uint32_t rel[REC_PIECES], cnt[REC_PIECES];
};
constexpr uint32_t REC_NO_CHAIN = 0xFFFFFFFFu;

// device columns of one batch (include/ngsq.h layout rules)
struct RecColumns {
    uint16_t *flag, *n_cigar;
    uint8_t *mapq;
    int32_t *ref_id, *pos, *mate_ref_id, *tlen;
    uint32_t *l_seq;
    uint8_t *seq, *qual;
    uint32_t *cigar;
    const uint64_t *seq_off, *qual_off, *cigar_off; // null: fixed pitch (cigar: one op per record)
    uint32_t seq_pitch, qual_pitch;
};

// where the bytes of the current view came from, for the records' ids (include/ngsq.h record_id = BAM virtual offset)
struct RecOrigin {
    uint64_t *record_id;     // [n] out; null: no ids wanted
    const BgzfBlock *blocks; // the chunk's block table (device), out_off relative to the chunk's first byte
    const uint64_t *coff;    // [n_blocks] file offset of every block
    uint32_t n_blocks;
    uint64_t carry;          // view bytes in front of the chunk's first byte (the record cut by the previous chunk's end)
    uint64_t carry_id;       // id of that record
};

// cand / pieces: REC_CANDIDATES entries per segment
// work (optional): the REC_WORK_WORDS device words; work[W_BAD] is set to ~0 for the chunk (launch_rec_offsets lowers it)
hipError_t launch_rec_candidates(const uint8_t *raw, uint64_t n_bytes, uint64_t first, uint32_t n_seg, int32_t n_ref,
                                 RecCandidate *cand, RecPieces *pieces, unsigned long long *work, hipStream_t s);
// the chain from `start` inside the segment [seg_start, end), walked by one thread with the host reader's rules
hipError_t launch_walk_one(const uint8_t *raw, uint64_t n_bytes, uint64_t start, uint64_t seg_start, uint64_t end, RecCandidate *out,
                           RecPieces *pieces, hipStream_t s);
// chosen[s]: which candidate of segment s is on the file's chain (REC_NO_CHAIN: none starts there); seg_base[s]: index of
// its first record.  One lane per REC_PIECE bytes writes the offsets of the records that start there.
// work: the REC_WORK_WORDS device words shared with launch_rec_fixed; work[W_BAD] (set to ~0 by the caller) = smallest index of an invalid record
hipError_t launch_rec_offsets(const uint8_t *raw, uint64_t n_bytes, uint32_t n_pieces, const uint32_t *chosen, const uint32_t *seg_base,
                              const RecPieces *pieces, uint64_t *rec_off, unsigned long long *work, hipStream_t s);
// copy n_bytes (rounded up to whole 32-bit words) with a kernel: for small tables between device memory and pinned host
// memory, which a hipMemcpyAsync would queue behind the large transfers of other streams
hipError_t launch_copy_words(void *dst, const void *src, uint64_t n_bytes, hipStream_t s);
// device to device, any alignment, by a kernel (not the DMA queue)
hipError_t launch_copy_bytes(void *dst, const void *src, uint64_t n_bytes, hipStream_t s);
// out[0] = number of entries of the ascending array a[0, n) that are < value (one thread)
hipError_t launch_count_below_u64(const uint64_t *a, uint64_t n, uint64_t value, unsigned long long *out, hipStream_t s);
// also writes var_base[i] = offset of record i's CIGAR in raw and var_base[n + i] = offset of its SEQ (2 n entries:
// what launch_rec_var starts from), and -- cig_len given -- n_cigar_op of every record as n + 1 64-bit entries (the last 0).
// work: REC_WORK_WORDS device words the kernels keep between launches (initial state: rec_work_init); host: REC_HOST_WORDS words of
// pinned host memory the device addresses, written by the launch's last block: what the layout decision needs, no copy
// afterwards.  host[H_BAD] != ~0: the chunk's record chain holds an invalid record (its index), nothing else was done.
// (W_INF0 / W_INF1: not the parse kernels' -- the block counters of the decoders on the two inflate streams, never reset: launch_bgzf_inflate's counter_base)
enum RecWork : uint32_t { W_BAD = 0, W_MAXL, W_MAXOPS, W_SUML, W_FIRST, W_LAST, W_LONG, W_SUMOPS, W_TICKET, W_INF0 = 12, W_INF1 = 13, REC_WORK_WORDS = 16 };
enum RecHost : uint32_t { H_MAXL = 0, H_MAXOPS, H_SUML, H_FIRST, H_LAST, H_LONG, H_SUMOPS, H_BAD, REC_HOST_WORDS = 8 };
inline void rec_work_init(unsigned long long *w) {
    for (uint32_t k = 0; k < REC_WORK_WORDS; k++) w[k] = 0;
    w[W_BAD] = ~0ull;
}
hipError_t launch_rec_fixed(const uint8_t *raw, const uint64_t *rec_off, uint64_t n, const RecColumns &c, uint64_t *var_base,
                            unsigned long long *work, unsigned long long *host, const RecOrigin &org, uint64_t *cig_len, hipStream_t s);
hipError_t launch_rec_lengths(const uint8_t *raw, const uint64_t *rec_off, uint64_t n, uint64_t *seq_len,
                              uint64_t *qual_len, uint64_t *cig_len, hipStream_t s);
// exclusive prefix sums of n+1 entries in place (entry n = total); tmp: scratch of *tmp_bytes
hipError_t launch_exclusive_scan_u64(uint64_t *data, uint64_t n_plus_1, void *tmp, size_t *tmp_bytes, hipStream_t s);
// c.flag .. c.l_seq filled by launch_rec_fixed, var_base from it
hipError_t launch_rec_var(const uint8_t *raw, const uint64_t *var_base, uint64_t n, const RecColumns &c, uint64_t seq_bytes,
                          uint64_t qual_bytes, hipStream_t s);

} // namespace ngsq
