// bgzf.h -- BGZF framing (SAM/BAM specification 4.1): split a byte buffer into the gzip members
// ("blocks") it holds.  Header-only, host side; shared by the host and the device ingest.
#pragma once

#include <stddef.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "ingest_kernels.h"

namespace ngsq {

inline uint32_t bgzf_rd32(const uint8_t *p) {
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
inline uint32_t bgzf_rd16(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }

// Append every COMPLETE block of c[0..n) to `blocks` (in_off relative to c, out_off running from
// *out_total) and set *consumed to the bytes they span.  Returns false (with *err) for bytes that
// are not BGZF.  A trailing incomplete block is left unconsumed, and so is everything from the first
// block that would take the decompressed total past out_limit.
inline bool bgzf_split(const uint8_t *c, size_t n, std::vector<BgzfBlock> *blocks, size_t *consumed,
                       uint64_t *out_total, std::string *err, uint64_t out_limit = ~0ull) {
    size_t p = 0;
    while (n - p >= 18) {
        if (c[p] != 31 || c[p + 1] != 139 || c[p + 2] != 8 || !(c[p + 3] & 4)) {
            *err = "not a BGZF block (bad gzip header)";
            return false;
        }
        const uint32_t xlen = bgzf_rd16(c + p + 10);
        if (n - p < 12 + (size_t)xlen) break;
        uint32_t bsize = 0;
        bool found = false;
        for (size_t q = p + 12; q + 4 <= p + 12 + xlen;) { // a subfield (SI1 SI2 SLEN data) must end inside the extra field
            const uint32_t slen = bgzf_rd16(c + q + 2);
            if (q + 4 + slen > p + 12 + xlen) {
                *err = "corrupt BGZF extra field";
                return false;
            }
            if (c[q] == 'B' && c[q + 1] == 'C' && slen == 2) {
                bsize = bgzf_rd16(c + q + 4) + 1;
                found = true;
            }
            q += 4 + slen;
        }
        if (!found) {
            *err = "BGZF block without BC subfield";
            return false;
        }
        if (bsize < 12 + xlen + 8) {
            *err = "corrupt BGZF block size";
            return false;
        }
        if (n - p < bsize) break; // incomplete block: wait for more bytes
        BgzfBlock bl{};
        bl.in_off = p + 12 + xlen;
        bl.in_len = bsize - 12 - xlen - 8;
        bl.crc = bgzf_rd32(c + p + bsize - 8);
        bl.isize = bgzf_rd32(c + p + bsize - 4);
        bl.out_off = *out_total;
        if (bl.isize > 65536) {
            *err = "BGZF ISIZE > 64 KiB";
            return false;
        }
        if (*out_total + bl.isize > out_limit) break;
        *out_total += bl.isize;
        blocks->push_back(bl);
        p += bsize;
    }
    *consumed = p;
    return true;
}

} // namespace ngsq
