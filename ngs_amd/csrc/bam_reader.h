// bam_reader.h -- private definition of ngsq_bam (shared by the host reader, bam_reader.cpp, and
// the device ingest, bam_device_reader.cpp)
#pragma once

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <string.h>

#include <string>
#include <thread>
#include <vector>

#include "../../include/ngsq_bam.h"

namespace ngsq {

// uninitialised growable byte buffer (std::vector would zero gigabytes on one core)
struct RawBuf {
    uint8_t *p = nullptr;
    size_t cap = 0;
    uint8_t *reserve(size_t n) {
        if (n > cap) {
            free(p);
            cap = n + n / 8 + 4096;
            p = (uint8_t *)malloc(cap);
        }
        return p;
    }
    ~RawBuf() { free(p); }
};

struct DeviceIngest; // bam_device_reader.cpp

// The CG:B,I tag in the auxiliary data [p, end) of a BAM record (SAM specification 4.2.4: tag[2] type[1] value) -- the real
// CIGAR of a record whose CIGAR field is the placeholder <l_seq>S<span>N (4.2.2).  Returns a pointer to its 32-bit operations
// and their number, or nullptr (no such tag; malformed data ends the walk).  (bam_device.hip has the same walk for the device reader.)
inline const uint8_t *aux_find_cg(const uint8_t *p, const uint8_t *end, uint32_t *n_ops) {
    while (end - p >= 4) {
        const uint8_t t0 = p[0], t1 = p[1], ty = p[2];
        p += 3;
        size_t n = 0;
        switch (ty) {
        case 'A': case 'c': case 'C': n = 1; break;
        case 's': case 'S': n = 2; break;
        case 'i': case 'I': case 'f': n = 4; break;
        case 'Z': case 'H':
            while (p < end && *p) p++;
            if (p >= end) return nullptr;
            n = 1;
            break;
        case 'B': {
            if (end - p < 5) return nullptr;
            const uint8_t sub = p[0];
            const uint32_t cnt = (uint32_t)p[1] | (uint32_t)p[2] << 8 | (uint32_t)p[3] << 16 | (uint32_t)p[4] << 24;
            const size_t w = sub == 'c' || sub == 'C' ? 1 : sub == 's' || sub == 'S' ? 2 : sub == 'i' || sub == 'I' || sub == 'f' ? 4 : 0;
            if (!w) return nullptr;
            n = 5 + (size_t)cnt * w;
            if (t0 == 'C' && t1 == 'G' && sub == 'I') {
                if ((size_t)(end - p) < n) return nullptr;
                *n_ops = cnt;
                return p + 5;
            }
            break;
        }
        default: return nullptr;
        }
        if ((size_t)(end - p) < n) return nullptr;
        p += n;
    }
    return nullptr;
}

// cores this process may really use: the cgroup's CPU quota when there is one (the MI355X boxes of this pool show 256
// online CPUs under a quota of 16: more runnable threads than that get throttled for the rest of the period)
inline int effective_cores() {
    int n = (int)std::thread::hardware_concurrency();
    if (n < 1) n = 1;
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char quota[32] = {0};
        long period = 0;
        if (fscanf(f, "%31s %ld", quota, &period) == 2 && strcmp(quota, "max") != 0 && period > 0) {
            const long q = atol(quota) / period;
            if (q >= 1 && q < n) n = (int)q;
        }
        fclose(f);
    }
    return n;
}

} // namespace ngsq

struct ngsq_bam {
    FILE *f = nullptr;
    std::string path;
    int n_threads = 1;
    bool eof = false;
    std::vector<uint8_t> comp;   // compressed bytes not yet consumed
    std::vector<uint8_t> data;   // decompressed bytes not yet parsed (starts at a record boundary after the header)
    size_t data_pos = 0;
    std::string header_text;
    std::vector<std::string> ref_names;
    std::vector<uint32_t> ref_lens;
    uint64_t n_read = 0;
    // batch columns
    std::vector<uint16_t> flag, n_cigar;
    std::vector<uint8_t> mapq, missing;
    ngsq::RawBuf seq, qual; // large: grown without initialisation, padded by the fill threads
    std::vector<int32_t> ref_id, pos, mate_ref_id, tlen;
    std::vector<uint32_t> l_seq, cigar;
    std::vector<uint64_t> seq_off, qual_off, cigar_off, record_id;
    // where the decompressed bytes came from: one entry per BGZF block with data that `data` still holds any of,
    // ascending; a record's id is its virtual offset, coff << 16 | (offset in the decompressed stream - abs_off)
    struct BlockOrigin {
        uint64_t abs_off; // offset of the block's first byte in the decompressed stream (data_base + index into data)
        uint64_t coff;    // file offset of the block
    };
    std::vector<BlockOrigin> origin;
    uint64_t comp_file_off = 0; // file offset of comp[0]
    // bookkeeping shared with the device ingest
    size_t read_chunk = (size_t)64 << 20; // compressed bytes per read
    uint64_t data_base = 0;               // offset in the decompressed stream of data[0]
    uint64_t header_bytes = 0;            // decompressed bytes before the first record
    bool host_mode = false;               // ngsq_bam_next_batch has been called
    ngsq::DeviceIngest *dev = nullptr;    // set by ngsq_bam_next_batch_device
    void (*dev_free)(ngsq::DeviceIngest *) = nullptr;
};

// bam_device_reader.cpp: ngsq_bam_shard_end that also says whether the shard's first record was an assumption
int ngsq_bam_shard_peek(ngsq_bam *b, ngsq_bam_shard_info *out, int *assumed);

// message of this thread's last failing ngsq_bam_* call (bam_reader.cpp)
int ngsq_bam_fail(int code, const char *fmt, ...);


