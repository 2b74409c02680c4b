// bam_reader.cpp -- host ingest (include/ngsq_bam.h): BGZF blocks inflated in parallel
// with zlib, BAM records parsed into structure-of-arrays batches.
//
// Replaces, for this path, what the reference gets from noodles-bgzf 0.20 / noodles-bam 0.28
// (`reader.records(&header)`, src/qc/command.rs:305; open_and_parse,
// src/utils/formats/bam.rs:77-123).  Format facts: SAM/BAM specification sections 4.1 (BGZF),
// 4.2 (BAM header, alignment records), 5.2 (BAI).
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/ngsq_bam.h"
#include "bam_reader.h"

static thread_local std::string g_bam_err;

int ngsq_bam_fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_bam_err = buf;
    return code;
}
#define bfail ngsq_bam_fail

using ngsq::RawBuf;

namespace {

inline uint32_t rd32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
inline uint16_t rd16(const uint8_t *p) { return (uint16_t)(p[0] | (p[1] << 8)); }

template <typename F> void parallel_for(int n_threads, uint64_t n, F fn) {
    const int nt = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)n_threads, n / 4096 + 1));
    if (nt == 1) {
        fn((uint64_t)0, n);
        return;
    }
    std::vector<std::thread> pool;
    const uint64_t per = (n + nt - 1) / nt;
    for (int t = 0; t < nt; t++) {
        const uint64_t lo = std::min(n, per * t), hi = std::min(n, lo + per);
        if (lo < hi) pool.emplace_back([=]() { fn(lo, hi); });
    }
    for (auto &th : pool) th.join();
}

struct Block {
    size_t in_off, in_len; // deflate payload inside the compressed buffer
    size_t out_off;
    uint32_t isize, crc;
    size_t start; // of the gzip member inside the compressed buffer
};

} // namespace

namespace {

// Read more compressed bytes and inflate every COMPLETE block in the buffer, appending to b->data.
// Returns 0, or a negative status.  Sets b->eof when the file is exhausted.
int inflate_more(ngsq_bam *b, size_t want_compressed) {
    if (!b->eof && want_compressed) {
        const size_t old = b->comp.size();
        b->comp.resize(old + want_compressed);
        const size_t got = fread(b->comp.data() + old, 1, want_compressed, b->f);
        b->comp.resize(old + got);
        if (got < want_compressed) {
            if (ferror(b->f)) return bfail(NGSQ_ERR_INVALID_ARGUMENT, "read error on %s", b->path.c_str());
            b->eof = true;
        }
    }
    // ---- split into BGZF blocks (spec 4.1: gzip member with the BC extra subfield)
    std::vector<Block> blocks;
    size_t p = 0, out_total = 0;
    const uint8_t *c = b->comp.data();
    const size_t n = b->comp.size();
    while (n - p >= 18) {
        if (c[p] != 31 || c[p + 1] != 139 || c[p + 2] != 8 || !(c[p + 3] & 4))
            return bfail(NGSQ_ERR_INVALID_ARGUMENT, "%s: not a BGZF block (bad gzip header)", b->path.c_str());
        const uint32_t xlen = rd16(c + p + 10);
        if (n - p < 12 + (size_t)xlen) break;
        uint32_t bsize = 0;
        bool found = false;
        for (size_t q = p + 12; q + 4 <= p + 12 + xlen;) { // a subfield (SI1 SI2 SLEN data) must end inside the extra field
            const uint32_t slen = rd16(c + q + 2);
            if (q + 4 + slen > p + 12 + xlen)
                return bfail(NGSQ_ERR_INVALID_ARGUMENT, "%s: corrupt BGZF extra field", b->path.c_str());
            if (c[q] == 'B' && c[q + 1] == 'C' && slen == 2) {
                bsize = (uint32_t)rd16(c + q + 4) + 1;
                found = true;
            }
            q += 4 + slen;
        }
        if (!found) return bfail(NGSQ_ERR_INVALID_ARGUMENT, "%s: BGZF block without BC subfield", b->path.c_str());
        if (bsize < 12 + xlen + 8) return bfail(NGSQ_ERR_INVALID_ARGUMENT, "%s: corrupt BGZF block size", b->path.c_str());
        if (n - p < bsize) break; // incomplete block: wait for more bytes
        Block bl;
        bl.start = p;
        bl.in_off = p + 12 + xlen;
        bl.in_len = bsize - 12 - xlen - 8;
        bl.crc = rd32(c + p + bsize - 8);
        bl.isize = rd32(c + p + bsize - 4);
        bl.out_off = out_total;
        if (bl.isize > 65536) return bfail(NGSQ_ERR_INVALID_ARGUMENT, "%s: BGZF ISIZE > 64 KiB", b->path.c_str());
        out_total += bl.isize;
        blocks.push_back(bl);
        p += bsize;
    }
    // ---- compact the unparsed tail of `data`, then inflate the blocks in parallel behind it
    if (b->data_pos) {
        b->data.erase(b->data.begin(), b->data.begin() + (ptrdiff_t)b->data_pos);
        b->data_base += b->data_pos;
        b->data_pos = 0;
    }
    const size_t base = b->data.size();
    b->data.resize(base + out_total);
    {   // origins of the bytes `data` holds: forget the blocks that have been parsed completely, note the new ones
        size_t k = 0;
        while (k + 1 < b->origin.size() && b->origin[k + 1].abs_off <= b->data_base) k++;
        b->origin.erase(b->origin.begin(), b->origin.begin() + (ptrdiff_t)k);
        for (const Block &bl : blocks)
            if (bl.isize) b->origin.push_back({b->data_base + base + bl.out_off, b->comp_file_off + bl.start});
    }
    std::atomic<size_t> next{0};
    std::atomic<int> bad{0};
    auto worker = [&]() {
        for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= blocks.size()) break;
            const Block &bl = blocks[i];
            if (bl.isize == 0) continue; // the empty EOF marker block
            z_stream zs;
            memset(&zs, 0, sizeof zs);
            if (inflateInit2(&zs, -15) != Z_OK) {
                bad = 1;
                continue;
            }
            zs.next_in = const_cast<Bytef *>(c + bl.in_off);
            zs.avail_in = (uInt)bl.in_len;
            zs.next_out = b->data.data() + base + bl.out_off;
            zs.avail_out = bl.isize;
            const int rc = inflate(&zs, Z_FINISH);
            inflateEnd(&zs);
            if (rc != Z_STREAM_END || zs.total_out != bl.isize) {
                bad = 1;
                continue;
            }
            if ((uint32_t)crc32(0L, b->data.data() + base + bl.out_off, bl.isize) != bl.crc) bad = 2;
        }
    };
    const int nt = (int)std::min<size_t>((size_t)b->n_threads, std::max<size_t>(blocks.size(), 1));
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; t++) pool.emplace_back(worker);
    worker();
    for (auto &t : pool) t.join();
    if (bad) return bfail(NGSQ_ERR_INVALID_ARGUMENT, "%s: BGZF %s", b->path.c_str(), bad == 2 ? "CRC mismatch" : "inflate failed");
    b->comp.erase(b->comp.begin(), b->comp.begin() + (ptrdiff_t)p);
    b->comp_file_off += p;
    return NGSQ_OK;
}

// make sure at least `need` unparsed decompressed bytes are available (fewer only at end of file)
int ensure(ngsq_bam *b, size_t need) {
    while (b->data.size() - b->data_pos < need) {
        if (b->eof && b->comp.empty()) break;
        const size_t a0 = b->data.size() - b->data_pos, c0 = b->comp.size();
        const bool e0 = b->eof;
        const int rc = inflate_more(b, e0 ? 0 : b->read_chunk);
        if (rc) return rc;
        if (e0 && b->data.size() - b->data_pos == a0 && b->comp.size() == c0)
            return bfail(NGSQ_ERR_INVALID_ARGUMENT, "%s: truncated BGZF block at end of file", b->path.c_str());
    }
    return NGSQ_OK;
}

} // namespace

extern "C" {

const char *ngsq_bam_last_error(void) { return g_bam_err.c_str(); }

int ngsq_bam_open(const char *path, int n_threads, ngsq_bam **out) {
    if (!path || !out) return bfail(NGSQ_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    FILE *f = fopen(path, "rb");
    if (!f) return bfail(NGSQ_ERR_INVALID_ARGUMENT, "opening BAM file: %s", path);
    ngsq_bam *b = new ngsq_bam();
    b->f = f;
    b->path = path;
    b->n_threads = n_threads > 0 ? n_threads : ngsq::effective_cores();
    b->read_chunk = (size_t)1 << 20; // the header needs little; the device ingest re-reads the file itself
#define OPEN_TRY(expr)            \
    do {                          \
        int rc_ = (expr);         \
        if (rc_) {                \
            ngsq_bam_close(b);    \
            return rc_;           \
        }                         \
    } while (0)
    // ---- header (spec 4.2): magic, l_text, text, n_ref, then per reference l_name, name, l_ref
    OPEN_TRY(ensure(b, 12));
    if (b->data.size() - b->data_pos < 12 || memcmp(b->data.data() + b->data_pos, "BAM\1", 4) != 0) {
        ngsq_bam_close(b);
        return bfail(NGSQ_ERR_INVALID_ARGUMENT, "reading BAM header: invalid magic number in %s", path);
    }
    const uint32_t l_text = rd32(b->data.data() + b->data_pos + 4);
    OPEN_TRY(ensure(b, 12 + (size_t)l_text));
    if (b->data.size() - b->data_pos < 12 + (size_t)l_text) {
        ngsq_bam_close(b);
        return bfail(NGSQ_ERR_INVALID_ARGUMENT, "reading BAM header: truncated header text");
    }
    b->header_text.assign((const char *)b->data.data() + b->data_pos + 8, l_text);
    while (!b->header_text.empty() && b->header_text.back() == '\0') b->header_text.pop_back();
    const uint32_t n_ref = rd32(b->data.data() + b->data_pos + 8 + l_text);
    b->data_pos += 12 + l_text;
    for (uint32_t r = 0; r < n_ref; r++) {
        OPEN_TRY(ensure(b, 4));
        if (b->data.size() - b->data_pos < 4) {
            ngsq_bam_close(b);
            return bfail(NGSQ_ERR_INVALID_ARGUMENT, "reading BAM reference sequences: truncated");
        }
        const uint32_t l_name = rd32(b->data.data() + b->data_pos);
        OPEN_TRY(ensure(b, 8 + (size_t)l_name));
        if (l_name == 0 || b->data.size() - b->data_pos < 8 + (size_t)l_name) {
            ngsq_bam_close(b);
            return bfail(NGSQ_ERR_INVALID_ARGUMENT, "reading BAM reference sequences: truncated");
        }
        const char *nm = (const char *)b->data.data() + b->data_pos + 4;
        b->ref_names.emplace_back(nm, strnlen(nm, l_name));
        b->ref_lens.push_back(rd32(b->data.data() + b->data_pos + 4 + l_name));
        b->data_pos += 8 + l_name;
    }
#undef OPEN_TRY
    b->header_bytes = b->data_base + b->data_pos;
    b->read_chunk = (size_t)64 << 20;
    *out = b;
    return NGSQ_OK;
}

void ngsq_bam_close(ngsq_bam *b) {
    if (!b) return;
    if (b->dev && b->dev_free) b->dev_free(b->dev);
    if (b->f) fclose(b->f);
    delete b;
}

uint32_t ngsq_bam_n_refs(const ngsq_bam *b) { return b ? (uint32_t)b->ref_names.size() : 0; }
const char *ngsq_bam_ref_name(const ngsq_bam *b, uint32_t i) { return b && i < b->ref_names.size() ? b->ref_names[i].c_str() : nullptr; }
uint32_t ngsq_bam_ref_len(const ngsq_bam *b, uint32_t i) { return b && i < b->ref_lens.size() ? b->ref_lens[i] : 0; }
const char *ngsq_bam_header_text(const ngsq_bam *b, uint64_t *len) {
    if (!b) return nullptr;
    if (len) *len = b->header_text.size();
    return b->header_text.c_str();
}
uint64_t ngsq_bam_records_read(const ngsq_bam *b) { return b ? b->n_read : 0; }

int ngsq_bam_next_batch(ngsq_bam *b, uint64_t max_records, ngsq_batch *out) {
    if (!b || !out) return bfail(NGSQ_ERR_INVALID_ARGUMENT, "null argument");
    if (b->dev) return bfail(NGSQ_ERR_STATE, "%s: this reader is in device ingest mode", b->path.c_str());
    b->host_mode = true;
    memset(out, 0, sizeof *out);
    out->struct_size = sizeof *out;
    out->location = NGSQ_MEM_HOST;
    out->first_record_index = b->n_read;
    if (max_records == 0) return NGSQ_OK;
    // ---- index the records of this batch inside the decompressed buffer
    std::vector<size_t> recs; // offsets (relative to data_pos) of block_size fields
    // the records whose CIGAR is in a CG tag: (index in recs, (offset of the tag's operations relative to data_pos, their number))
    std::vector<std::pair<size_t, std::pair<size_t, uint32_t>>> long_cigar;
    size_t cursor = 0;
    uint32_t max_l = 0, max_ops = 0;
    uint64_t sum_qual = 0;
    while (recs.size() < max_records) {
        int rc = ensure(b, cursor + 4);
        if (rc) return rc;
        const size_t avail = b->data.size() - b->data_pos;
        if (avail < cursor + 4) {
            if (avail != cursor) return bfail(NGSQ_ERR_INVALID_ARGUMENT, "%s: truncated record", b->path.c_str());
            break; // clean end of file
        }
        const uint32_t block_size = rd32(b->data.data() + b->data_pos + cursor);
        if (block_size < 32) return bfail(NGSQ_ERR_INVALID_ARGUMENT, "%s: invalid record block_size %u", b->path.c_str(), block_size);
        rc = ensure(b, cursor + 4 + (size_t)block_size);
        if (rc) return rc;
        if (b->data.size() - b->data_pos < cursor + 4 + (size_t)block_size)
            return bfail(NGSQ_ERR_INVALID_ARGUMENT, "%s: truncated record", b->path.c_str());
        const uint8_t *r = b->data.data() + b->data_pos + cursor + 4;
        const uint32_t l_read_name = r[8], n_ops = rd16(r + 12), l = rd32(r + 16);
        const uint64_t need = 32ull + l_read_name + 4ull * n_ops + (l + 1) / 2 + l;
        if (l_read_name == 0 || need > block_size)
            return bfail(NGSQ_ERR_INVALID_ARGUMENT, "%s: malformed record %llu", b->path.c_str(),
                         (unsigned long long)(b->n_read + recs.size()));
        // SAM/BAM specification 4.2.2: a CIGAR of more than 65535 operations is stored in a CG:B,I tag and the CIGAR field
        // holds the placeholder <l_seq>S<reference span>N.  noodles resolves the tag while it decodes the record, so the
        // facets see the real operations ([N10] in oracle/oracle.h): so do the batches (ABI 5: n_cigar saturates at
        // 65535, the real count is in cigar_off).  The placeholder alone -- no tag -- is an odd but real alignment.
        uint32_t real_ops = n_ops;
        if (n_ops == 2 && l > 0) {
            const uint8_t *cg = r + 32 + l_read_name;
            if (rd32(cg) == (l << 4 | 4u) && (rd32(cg + 4) & 15u) == 3u) {
                uint32_t cnt = 0;
                const uint8_t *ops = ngsq::aux_find_cg(r + need, r + block_size, &cnt);
                if (ops && cnt >= 2) { // (the convention is for more than 65535 operations; a tag with fewer than the placeholder's two is ignored)
                    long_cigar.emplace_back(recs.size(), std::make_pair((size_t)(ops - (b->data.data() + b->data_pos)), cnt));
                    real_ops = cnt;
                }
            }
        }
        recs.push_back(cursor);
        max_l = std::max(max_l, l);
        max_ops = std::max(max_ops, real_ops);

        sum_qual += l;

        cursor += 4 + (size_t)block_size;
    }
    const uint64_t n = recs.size();
    if (!n) return NGSQ_OK;
    // ---- choose the layout (include/ngsq.h): fixed-pitch rows are the device fast path
    const uint32_t pitch_q = max_l, pitch_s = (max_l + 1) / 2;
    const bool fixed = max_l >= 1 && max_l <= 320 && (uint64_t)pitch_q * n <= sum_qual + sum_qual / 2 + 4096;
    const bool cig1 = max_ops <= 1;
    b->flag.resize(n); b->n_cigar.resize(n); b->mapq.resize(n + 16);
    b->ref_id.resize(n); b->pos.resize(n); b->mate_ref_id.resize(n); b->tlen.resize(n); b->l_seq.resize(n);
    b->record_id.resize(n);
    const uint8_t *const D = b->data.data() + b->data_pos;
    uint64_t so = 0, qo = 0, co = 0;
    if (!fixed || !cig1) {
        // offsets layout: absent qualities (l_seq bytes of 0xFF, spec 4.2.3; noodles yields no scores)
        // take no bytes, so the offsets need the `missing` flags first (parallel), then one prefix pass
        b->missing.resize(n);
        if (!fixed) {
            parallel_for(b->n_threads, n, [&](uint64_t lo, uint64_t hi) {
                for (uint64_t i = lo; i < hi; i++) {
                    const uint8_t *r = D + recs[i] + 4;
                    const uint32_t l = rd32(r + 16);
                    const uint8_t *ql = r + 32 + r[8] + 4ull * rd16(r + 12) + (l + 1) / 2;
                    bool miss = l > 0;
                    for (uint32_t k = 0; k < l && miss; k++) miss = ql[k] == 0xFF;
                    b->missing[i] = miss;
                }
            });
            b->seq_off.resize(n + 1);
            b->qual_off.resize(n + 1);
        }
        if (!cig1) b->cigar_off.resize(n + 1);
        size_t lc = 0;
        for (uint64_t i = 0; i < n; i++) {
            const uint8_t *r = D + recs[i] + 4;
            const uint32_t n_ops = rd16(r + 12), l = rd32(r + 16);
            if (!fixed) {
                b->seq_off[i] = so;
                b->qual_off[i] = qo;
                so += (l + 1) / 2;
                if (!b->missing[i]) qo += l;
            }
            if (!cig1) {
                b->cigar_off[i] = co;
                co += n_ops;
                if (lc < long_cigar.size() && long_cigar[lc].first == i) co += long_cigar[lc++].second.second - n_ops; // (the tag's count instead of the placeholder's 2)
            }
        }
    }
    uint8_t *const SEQ = b->seq.reserve(fixed ? (size_t)pitch_s * n + 64 : (size_t)so + 64);
    uint8_t *const QUAL = b->qual.reserve(fixed ? (size_t)pitch_q * n + 64 : (size_t)qo + 64);
    b->cigar.resize((cig1 ? n : co) + 16);
    parallel_for(b->n_threads, n, [&](uint64_t lo, uint64_t hi) {
        for (uint64_t i = lo; i < hi; i++) {
            const uint8_t *r = D + recs[i] + 4;
            const uint32_t l_read_name = r[8], n_ops = rd16(r + 12), l = rd32(r + 16);
            b->ref_id[i] = (int32_t)rd32(r);
            b->pos[i] = (int32_t)rd32(r + 4);
            b->mapq[i] = r[9];
            b->n_cigar[i] = (uint16_t)n_ops; // (a record with a CG tag: below)
            b->flag[i] = rd16(r + 14);
            b->l_seq[i] = l;
            b->mate_ref_id[i] = (int32_t)rd32(r + 20);
            b->tlen[i] = (int32_t)rd32(r + 28);
            {   // the record's identity: its BAM virtual offset (include/ngsq.h record_id)
                const uint64_t abs = b->data_base + b->data_pos + recs[i];
                size_t lo_k = 0, hi_k = b->origin.size(); // last origin with abs_off <= abs
                while (hi_k - lo_k > 1) {
                    const size_t mid = (lo_k + hi_k) / 2;
                    if (b->origin[mid].abs_off <= abs) lo_k = mid;
                    else hi_k = mid;
                }
                b->record_id[i] = b->origin[lo_k].coff << 16 | (abs - b->origin[lo_k].abs_off);
            }
            const uint8_t *cg = r + 32 + l_read_name;
            const uint8_t *sq = cg + 4ull * n_ops;
            const uint8_t *ql = sq + (l + 1) / 2;
            if (cig1) {
                b->cigar[i] = n_ops ? rd32(cg) : 0u;
            } else {
                const uint64_t c0 = b->cigar_off[i];
                for (uint32_t k = 0; k < n_ops; k++) b->cigar[c0 + k] = rd32(cg + 4 * k);
            }
            if (fixed) {
                // rows padded here: zero nibbles for SEQ, 0xFF ("no score") for QUAL; an absent-quality
                // record is already a row of 0xFF in the file
                uint8_t *srow = SEQ + (size_t)pitch_s * i, *qrow = QUAL + (size_t)pitch_q * i;
                memcpy(srow, sq, (l + 1) / 2);
                memset(srow + (l + 1) / 2, 0, pitch_s - (l + 1) / 2);
                memcpy(qrow, ql, l);
                memset(qrow + l, 0xFF, pitch_q - l);
            } else {
                memcpy(SEQ + b->seq_off[i], sq, (l + 1) / 2);
                if (!b->missing[i]) memcpy(QUAL + b->qual_off[i], ql, l);
            }
        }
    });
    for (const auto &lc : long_cigar) { // the real operations of the records whose CIGAR field is the placeholder (cig1 is false: they have two)
        const uint64_t i = lc.first, c0 = b->cigar_off[i];
        const uint32_t cnt = lc.second.second;
        const uint8_t *ops = D + lc.second.first;
        b->n_cigar[i] = (uint16_t)std::min<uint32_t>(cnt, 0xFFFFu);
        for (uint32_t k = 0; k < cnt; k++) b->cigar[c0 + k] = rd32(ops + 4 * k);
    }
    memset(SEQ + (fixed ? (size_t)pitch_s * n : (size_t)so), 0, 64);     // slack read by the device's vector loads
    memset(QUAL + (fixed ? (size_t)pitch_q * n : (size_t)qo), 0xFF, 64);
    b->data_pos += cursor;
    b->n_read += n;
    out->n_records = n;
    out->flag = b->flag.data();
    out->mapq = b->mapq.data();
    out->ref_id = b->ref_id.data();
    out->pos = b->pos.data();
    out->mate_ref_id = b->mate_ref_id.data();
    out->tlen = b->tlen.data();
    out->l_seq = b->l_seq.data();
    out->n_cigar = b->n_cigar.data();
    out->seq = b->seq.p;
    out->qual = b->qual.p;
    out->cigar = b->cigar.data();
    out->record_id = b->record_id.data();
    out->max_l_seq = max_l;
    if (fixed) {
        out->seq_stride = pitch_s;
        out->qual_stride = pitch_q;
        out->seq_bytes = (uint64_t)pitch_s * n;
        out->qual_bytes = (uint64_t)pitch_q * n;
    } else {
        b->seq_off[n] = so;
        b->qual_off[n] = qo;
        out->seq_off = b->seq_off.data();
        out->qual_off = b->qual_off.data();
        out->seq_bytes = so;
        out->qual_bytes = qo;
    }
    if (cig1) {
        out->cigar_stride = 1;
        out->cigar_ops = n;
    } else {
        b->cigar_off[n] = co;
        out->cigar_off = b->cigar_off.data();
        out->cigar_ops = co;
    }
    return NGSQ_OK;
}

// spec 5.2: magic "BAI\1", n_ref, per ref: n_bin {bin u32, n_chunk i32, chunks 16 B each}, n_intv, ioffsets.
// starts (optional, [n_starts]): smallest chunk begin of each reference (0: nothing indexed); bins (optional): number
// of bins that hold records, the metadata pseudo-bin 37450 not counted.
static int parse_bai(const char *bam_path, uint32_t n_starts, uint64_t *starts, uint64_t *bins) {
    if (!bam_path) return bfail(NGSQ_ERR_INVALID_ARGUMENT, "null argument");
    const std::string p = std::string(bam_path) + ".bai";
    FILE *f = fopen(p.c_str(), "rb");
    if (!f) return bfail(NGSQ_ERR_INVALID_ARGUMENT, "reading BAM index: cannot open %s", p.c_str());
    std::vector<uint8_t> d;
    uint8_t buf[1 << 16];
    size_t g;
    while ((g = fread(buf, 1, sizeof buf, f)) > 0) d.insert(d.end(), buf, buf + g);
    fclose(f);
    size_t q = 0;
    auto need = [&](size_t k) { return q + k <= d.size(); };
    auto rd64 = [&](size_t at) { return (uint64_t)rd32(d.data() + at) | ((uint64_t)rd32(d.data() + at + 4) << 32); };
    if (!need(8) || memcmp(d.data(), "BAI\1", 4) != 0)
        return bfail(NGSQ_ERR_INVALID_ARGUMENT, "reading BAM index: invalid BAI magic in %s", p.c_str());
    const uint32_t n_ref = rd32(d.data() + 4);
    q = 8;
    if (bins) *bins = 0;
    for (uint32_t r = 0; r < n_starts; r++) starts[r] = 0;
    for (uint32_t r = 0; r < n_ref; r++) {
        if (!need(4)) return bfail(NGSQ_ERR_INVALID_ARGUMENT, "reading BAM index: truncated (bins of reference %u)", r);
        const uint32_t n_bin = rd32(d.data() + q);
        q += 4;
        for (uint32_t k = 0; k < n_bin; k++) {
            if (!need(8)) return bfail(NGSQ_ERR_INVALID_ARGUMENT, "reading BAM index: truncated bin");
            const uint32_t bin = rd32(d.data() + q), n_chunk = rd32(d.data() + q + 4);
            q += 8;
            if (!need((size_t)n_chunk * 16)) return bfail(NGSQ_ERR_INVALID_ARGUMENT, "reading BAM index: truncated chunks");
            if (bin != 37450u) { // 37450: samtools' metadata pseudo-bin
                if (bins) *bins += 1;
                for (uint32_t c = 0; c < n_chunk; c++) {
                    const uint64_t beg = rd64(q + (size_t)c * 16);
                    if (r < n_starts && (starts[r] == 0 || beg < starts[r])) starts[r] = beg;
                }
            }
            q += (size_t)n_chunk * 16;
        }
        if (!need(4)) return bfail(NGSQ_ERR_INVALID_ARGUMENT, "reading BAM index: truncated (intervals)");
        const uint32_t n_intv = rd32(d.data() + q);
        q += 4;
        if (!need((size_t)n_intv * 8)) return bfail(NGSQ_ERR_INVALID_ARGUMENT, "reading BAM index: truncated linear index");
        q += (size_t)n_intv * 8;
    }
    if (q != d.size() && q + 8 != d.size())
        return bfail(NGSQ_ERR_INVALID_ARGUMENT, "reading BAM index: trailing bytes in %s", p.c_str());
    return NGSQ_OK;
}

int ngsq_bam_check_index(const char *bam_path) { return parse_bai(bam_path, 0, nullptr, nullptr); }

int ngsq_bam_index_ref_starts(const char *bam_path, uint32_t n_refs, uint64_t *start_voffset, uint64_t *n_bins) {
    if (n_refs && !start_voffset) return bfail(NGSQ_ERR_INVALID_ARGUMENT, "null argument");
    return parse_bai(bam_path, n_refs, start_voffset, n_bins);
}

int ngsq_bam_seek(ngsq_bam *b, uint64_t voffset) {
    if (!b) return bfail(NGSQ_ERR_INVALID_ARGUMENT, "null argument");
    if (b->dev) return bfail(NGSQ_ERR_STATE, "%s: this reader is in device ingest mode", b->path.c_str());
    if (fseeko(b->f, (off_t)(voffset >> 16), SEEK_SET) != 0)
        return bfail(NGSQ_ERR_INVALID_ARGUMENT, "%s: cannot seek to block %llu", b->path.c_str(), (unsigned long long)(voffset >> 16));
    b->comp.clear();
    b->comp_file_off = voffset >> 16;
    b->origin.clear();
    b->data.clear();
    b->data_pos = 0;
    b->data_base = 0;
    b->eof = false;
    b->read_chunk = (size_t)4 << 20; // a query usually wants a few records: read little at a time
    const size_t in_block = (size_t)(voffset & 0xFFFFu);
    const int rc = ensure(b, in_block + 1);
    if (rc) return rc;
    if (b->data.size() < in_block) // == : the offset points at the end of the data (no record follows)
        return bfail(NGSQ_ERR_INVALID_ARGUMENT, "%s: virtual offset %llu lies outside its block", b->path.c_str(),
                     (unsigned long long)voffset);
    b->data_pos = in_block;
    return NGSQ_OK;
}

} // extern "C"
