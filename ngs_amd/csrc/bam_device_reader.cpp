// bam_device_reader.cpp -- device ingest (include/ngsq_bam.h): compressed BGZF bytes cross PCIe,
// the GPU inflates them and parses the BAM records into structure-of-arrays columns.
#include <hip/hip_runtime_api.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/ngsq_bam.h"
#include "bgzf.h"
#include "context.h"
#include "ingest_kernels.h"

using namespace ngsq;

namespace {

int dfail(ngsq_ctx *c, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf;
    return code;
}

#define DHIP(c, expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return dfail(c, NGSQ_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

const char *inflate_status_text(uint32_t s) {
    switch (s) {
    case INF_BAD_BLOCK_TYPE: return "invalid DEFLATE block type";
    case INF_BAD_STORED_LEN: return "stored block length check failed";
    case INF_BAD_CODE_LENGTHS: return "invalid Huffman code lengths";
    case INF_BAD_SYMBOL: return "invalid literal/length or distance code";
    case INF_BAD_DISTANCE: return "match distance reaches before the block";
    case INF_OUTPUT_OVERRUN: return "more data than ISIZE";
    case INF_INPUT_OVERRUN: return "compressed data ends inside a symbol";
    case INF_SIZE_MISMATCH: return "decompressed size differs from ISIZE";
    case INF_CRC_MISMATCH: return "CRC mismatch";
    default: return "ok";
    }
}

} // namespace

extern "C" {

int ngsq_bgzf_inflate_device(ngsq_ctx *c, const uint8_t *comp, uint64_t comp_len, uint8_t *out, uint64_t out_cap,
                             uint64_t *out_len, int check_crc) {
    if (!c || !comp || !out_len) return dfail(c, NGSQ_ERR_INVALID_ARGUMENT, "null argument");
    std::vector<BgzfBlock> blocks;
    size_t consumed = 0;
    uint64_t total = 0;
    std::string err;
    if (!bgzf_split(comp, comp_len, &blocks, &consumed, &total, &err))
        return dfail(c, NGSQ_ERR_INVALID_ARGUMENT, "%s", err.c_str());
    if (consumed != comp_len) return dfail(c, NGSQ_ERR_INVALID_ARGUMENT, "truncated BGZF block at end of buffer");
    *out_len = total;
    if (total > out_cap) return dfail(c, NGSQ_ERR_INVALID_ARGUMENT, "output buffer too small: %llu > %llu",
                                      (unsigned long long)total, (unsigned long long)out_cap);
    if (blocks.empty()) return NGSQ_OK;
    DHIP(c, hipSetDevice(c->device));
    uint8_t *d_comp = nullptr, *d_out = nullptr;
    BgzfBlock *d_blocks = nullptr;
    uint32_t *d_status = nullptr;
    int rc = NGSQ_OK;
    std::vector<uint32_t> status(blocks.size());
    auto cleanup = [&]() {
        (void)hipFree(d_comp);
        (void)hipFree(d_out);
        (void)hipFree(d_blocks);
        (void)hipFree(d_status);
    };
#define TRY(expr)                                                                                   \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess) {                                                                     \
            cleanup();                                                                              \
            return dfail(c, NGSQ_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(e_));                  \
        }                                                                                           \
    } while (0)
    TRY(hipMalloc((void **)&d_comp, comp_len + INFLATE_IN_SLACK));
    TRY(hipMalloc((void **)&d_out, total + 64));
    TRY(hipMalloc((void **)&d_blocks, blocks.size() * sizeof(BgzfBlock)));
    TRY(hipMalloc((void **)&d_status, blocks.size() * sizeof(uint32_t)));
    TRY(hipMemcpyAsync(d_comp, comp, comp_len, hipMemcpyHostToDevice, c->stream));
    TRY(hipMemsetAsync(d_comp + comp_len, 0, INFLATE_IN_SLACK, c->stream));
    TRY(hipMemcpyAsync(d_blocks, blocks.data(), blocks.size() * sizeof(BgzfBlock), hipMemcpyHostToDevice, c->stream));
    TRY(launch_bgzf_inflate(d_comp, d_blocks, (uint32_t)blocks.size(), d_out, d_status, check_crc != 0, c->stream));
    TRY(hipMemcpyAsync(status.data(), d_status, blocks.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    if (total) TRY(hipMemcpyAsync(out, d_out, total, hipMemcpyDeviceToHost, c->stream));
    TRY(hipStreamSynchronize(c->stream));
#undef TRY
    for (size_t k = 0; k < blocks.size() && rc == NGSQ_OK; k++)
        if (status[k] != INF_OK)
            rc = dfail(c, NGSQ_ERR_INVALID_ARGUMENT, "BGZF block %zu: %s", k, inflate_status_text(status[k]));
    cleanup();
    return rc;
}

} // extern "C"
