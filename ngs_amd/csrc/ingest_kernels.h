// ingest_kernels.h -- device ingest: BGZF inflate and BAM record
// parse -> structure-of-arrays columns on the GPU.  Launchers only; no torch, no host I/O.
#pragma once

#include <hip/hip_runtime_api.h>
#include <stdint.h>

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

// Inflate blocks [0, n_blocks): one wavefront per block.  status[k] = InflateStatus of block k.
hipError_t launch_bgzf_inflate(const uint8_t *comp, const BgzfBlock *blocks, uint32_t n_blocks, uint8_t *out,
                               uint32_t *status, bool check_crc, hipStream_t s);

} // namespace ngsq
