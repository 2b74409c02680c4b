// synth_bam.cpp -- write the synthetic records of include/ngsq_shared.h as a real BGZF BAM
// (+ its BAI: binning and linear index, SAM spec 5.2) so that the file-to-JSON path of `ngs qc`,
// including the region queries of `-n`, can be measured at size.
// Bench / test utility (the reference's `ngs generate` writes FASTQ from a FASTA with an
// unseeded RNG, src/generate/command.rs:59-131, and cannot produce these files).
#include <zlib.h>

#include <atomic>
#include <cstdio>
#include <map>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/ngsq_synth.h"
#include "bam_reader.h"

namespace {

void put32(std::vector<uint8_t> &v, uint32_t x) {
    for (int k = 0; k < 4; k++) v.push_back((uint8_t)(x >> (8 * k)));
}
void put16(std::vector<uint8_t> &v, uint32_t x) {
    v.push_back((uint8_t)x);
    v.push_back((uint8_t)(x >> 8));
}

uint32_t reg2bin(int64_t beg, int64_t end) { // SAM spec 5.3
    --end;
    if (beg >> 14 == end >> 14) return (uint32_t)(((1 << 15) - 1) / 7 + (beg >> 14));
    if (beg >> 17 == end >> 17) return (uint32_t)(((1 << 12) - 1) / 7 + (beg >> 17));
    if (beg >> 20 == end >> 20) return (uint32_t)(((1 << 9) - 1) / 7 + (beg >> 20));
    if (beg >> 23 == end >> 23) return (uint32_t)(((1 << 6) - 1) / 7 + (beg >> 23));
    if (beg >> 26 == end >> 26) return (uint32_t)(((1 << 3) - 1) / 7 + (beg >> 26));
    return 0;
}

struct RecIdx { // what the index needs to know about one record
    int32_t ref, pos, end; // end: exclusive, at least pos + 1
    uint32_t uoff;         // offset of the record in its block's data
};

void append_record(std::vector<uint8_t> &out, const ngsq_synth_config &cfg, uint64_t i, RecIdx *ix) {
    ngsq_synth_record r;
    ngsq_synth_record_at(&cfg, i, &r);
    ix->uoff = (uint32_t)out.size();
    char name[32];
    const int ln = snprintf(name, sizeof name, "r%llu", (unsigned long long)i) + 1;
    uint64_t span = 0;
    for (uint32_t k = 0; k < r.n_cigar; k++)
        if ((0x18Du >> (r.cigar[k] & 15)) & 1u) span += r.cigar[k] >> 4;
    ix->ref = r.ref_id;
    ix->pos = r.pos;
    ix->end = r.pos + (int32_t)(span ? span : 1);
    const uint32_t l = r.l_seq;
    const uint32_t block = 32 + (uint32_t)ln + 4 * r.n_cigar + (l + 1) / 2 + l;
    put32(out, block);
    put32(out, (uint32_t)r.ref_id);
    put32(out, (uint32_t)r.pos);
    out.push_back((uint8_t)ln);
    out.push_back(r.mapq);
    put16(out, reg2bin(r.pos, r.pos + (int64_t)(span ? span : 1)));
    put16(out, r.n_cigar);
    put16(out, r.flag);
    put32(out, l);
    put32(out, (uint32_t)r.mate_ref_id);
    put32(out, (uint32_t)-1);
    put32(out, (uint32_t)r.tlen);
    out.insert(out.end(), name, name + ln);
    for (uint32_t k = 0; k < r.n_cigar; k++) put32(out, r.cigar[k]);
    for (uint32_t j = 0; j < (l + 1) / 2; j++) out.push_back(ngsq_synth_seq_byte(&cfg, i, l, j));
    for (uint32_t j = 0; j < l; j++) out.push_back(ngsq_synth_qual_byte(&cfg, i, l, j));
}

bool bgzf_write(FILE *f, const uint8_t *data, size_t n, int level, std::vector<uint8_t> &scratch) {
    // one BGZF block (<= 64 KiB payload)
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return false;
    scratch.resize(deflateBound(&zs, (uLong)n) + 64);
    zs.next_in = const_cast<Bytef *>(data);
    zs.avail_in = (uInt)n;
    zs.next_out = scratch.data() + 18;
    zs.avail_out = (uInt)(scratch.size() - 18);
    const int rc = deflate(&zs, Z_FINISH);
    const size_t clen = zs.total_out;
    deflateEnd(&zs);
    if (rc != Z_STREAM_END) return false;
    static const uint8_t head[16] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 'B', 'C', 2, 0};
    memcpy(scratch.data(), head, 16);
    const size_t bsize = clen + 26;
    if (bsize > 65536) return false;
    scratch[16] = (uint8_t)((bsize - 1) & 0xFF);
    scratch[17] = (uint8_t)((bsize - 1) >> 8);
    uint8_t *tail = scratch.data() + 18 + clen;
    const uint32_t crc = (uint32_t)crc32(0L, data, (uInt)n), isz = (uint32_t)n;
    for (int k = 0; k < 4; k++) tail[k] = (uint8_t)(crc >> (8 * k)), tail[4 + k] = (uint8_t)(isz >> (8 * k));
    return fwrite(scratch.data(), 1, bsize, f) == bsize;
}

} // namespace

extern "C" int ngsq_synth_write_bam(const ngsq_synth_config *cfg, const char *path, uint64_t n_records, int level,
                                    int n_threads) {
    if (!cfg || !path) return NGSQ_ERR_INVALID_ARGUMENT;
    static const char *names[2] = {"chr1", "chr2"};
    const uint32_t lens[2] = {cfg->ref_len, 242193529u};
    const uint32_t n_refs = cfg->n_refs >= 2 ? 2 : 1;
    FILE *f = fopen(path, "wb");
    if (!f) return NGSQ_ERR_INVALID_ARGUMENT;
    std::vector<uint8_t> scratch;
    // ---- header block
    std::string text = "@HD\tVN:1.6\tSO:coordinate\n";
    for (uint32_t r = 0; r < n_refs; r++) text += std::string("@SQ\tSN:") + names[r] + "\tLN:" + std::to_string(lens[r]) + "\n";
    std::vector<uint8_t> head = {'B', 'A', 'M', 1};
    put32(head, (uint32_t)text.size());
    head.insert(head.end(), text.begin(), text.end());
    put32(head, n_refs);
    for (uint32_t r = 0; r < n_refs; r++) {
        put32(head, (uint32_t)strlen(names[r]) + 1);
        head.insert(head.end(), names[r], names[r] + strlen(names[r]) + 1);
        put32(head, lens[r]);
    }
    bool ok = bgzf_write(f, head.data(), head.size(), level, scratch);
    // ---- index under construction (records arrive in file order)
    std::vector<std::map<uint32_t, std::vector<std::pair<uint64_t, uint64_t>>>> bins(n_refs);
    std::vector<std::vector<uint64_t>> linear(n_refs);
    uint64_t n_no_coor = 0;
    uint64_t file_off = (uint64_t)ftello(f); // where the next block starts
    // consecutive records of a sorted file mostly fall into the same bin: one map lookup per run, not per record
    int32_t cached_ref = -1;
    uint32_t cached_bin = 0;
    std::vector<std::pair<uint64_t, uint64_t>> *cached = nullptr;
    auto index_record = [&](const RecIdx &x, uint64_t v0, uint64_t v1) {
        if (x.ref < 0 || (uint32_t)x.ref >= n_refs || x.pos < 0) {
            n_no_coor += 1;
            return;
        }
        const uint32_t bin = reg2bin(x.pos, x.end);
        if (!cached || cached_ref != x.ref || cached_bin != bin) {
            cached = &bins[x.ref][bin];
            cached_ref = x.ref;
            cached_bin = bin;
        }
        auto &chunks = *cached;
        if (!chunks.empty() && chunks.back().second == v0)
            chunks.back().second = v1; // the records of a bin that follow each other share a chunk
        else
            chunks.emplace_back(v0, v1);
        auto &lin = linear[x.ref];
        const size_t w1 = (size_t)((x.end - 1) >> 14);
        if (lin.size() <= w1) lin.resize(w1 + 1, 0);
        for (size_t w = (size_t)(x.pos >> 14); w <= w1; w++)
            if (lin[w] == 0) lin[w] = v0; // file order: the first record that overlaps the window
    };
    // ---- records: groups of records rendered and deflated in parallel, written in order
    const int nt = n_threads > 0 ? n_threads : ngsq::effective_cores();
    const uint64_t group = 100; // records per BGZF block (100 x ~270 B < 64 KiB even at 300 bp)
    const uint64_t n_groups = (n_records + group - 1) / group;
    const uint64_t wave = (uint64_t)nt * 64;
    for (uint64_t g0 = 0; g0 < n_groups && ok; g0 += wave) {
        const uint64_t g1 = std::min(n_groups, g0 + wave);
        std::vector<std::vector<uint8_t>> blocks(g1 - g0);
        std::vector<std::vector<RecIdx>> recs(g1 - g0);
        std::atomic<uint64_t> next{g0};
        std::atomic<int> bad{0};
        auto worker = [&]() {
            std::vector<uint8_t> raw, sc;
            for (;;) {
                const uint64_t g = next.fetch_add(1);
                if (g >= g1) break;
                raw.clear();
                auto &rx = recs[g - g0];
                for (uint64_t i = g * group; i < std::min(n_records, (g + 1) * group); i++) {
                    rx.emplace_back();
                    append_record(raw, *cfg, i, &rx.back());
                }
                // deflate into a memory "file"
                z_stream zs;
                memset(&zs, 0, sizeof zs);
                if (deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) { bad = 1; continue; }
                sc.resize(deflateBound(&zs, (uLong)raw.size()) + 64);
                zs.next_in = raw.data();
                zs.avail_in = (uInt)raw.size();
                zs.next_out = sc.data() + 18;
                zs.avail_out = (uInt)(sc.size() - 18);
                const int rc = deflate(&zs, Z_FINISH);
                const size_t clen = zs.total_out;
                deflateEnd(&zs);
                if (rc != Z_STREAM_END || clen + 26 > 65536 || raw.size() > 65536) { bad = 1; continue; }
                static const uint8_t hd[16] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 'B', 'C', 2, 0};
                memcpy(sc.data(), hd, 16);
                const size_t bsize = clen + 26;
                sc[16] = (uint8_t)((bsize - 1) & 0xFF);
                sc[17] = (uint8_t)((bsize - 1) >> 8);
                uint8_t *tail = sc.data() + 18 + clen;
                const uint32_t crc = (uint32_t)crc32(0L, raw.data(), (uInt)raw.size()), isz = (uint32_t)raw.size();
                for (int k = 0; k < 4; k++) tail[k] = (uint8_t)(crc >> (8 * k)), tail[4 + k] = (uint8_t)(isz >> (8 * k));
                blocks[g - g0].assign(sc.begin(), sc.begin() + (ptrdiff_t)bsize);
            }
        };
        std::vector<std::thread> pool;
        for (int t = 1; t < nt; t++) pool.emplace_back(worker);
        worker();
        for (auto &t : pool) t.join();
        if (bad) ok = false;
        for (size_t k = 0; k < blocks.size() && ok; k++) {
            const auto &b = blocks[k];
            if (fwrite(b.data(), 1, b.size(), f) != b.size()) ok = false;
            const uint64_t next_off = file_off + b.size();
            const auto &rx = recs[k];
            for (size_t j = 0; j < rx.size(); j++)
                index_record(rx[j], (file_off << 16) | rx[j].uoff, j + 1 < rx.size() ? (file_off << 16) | rx[j + 1].uoff : next_off << 16);
            file_off = next_off;
        }
    }
    static const uint8_t eof_block[28] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 0x42, 0x43, 0x02, 0,
                                          0x1b, 0, 0x03, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (ok && fwrite(eof_block, 1, 28, f) != 28) ok = false;
    fclose(f);
    if (!ok) return NGSQ_ERR_INVALID_ARGUMENT;
    // ---- the BAI: per reference the bins with their chunks, then the 16 kb linear index (gaps carry the previous value)
    FILE *bi = fopen((std::string(path) + ".bai").c_str(), "wb");
    if (!bi) return NGSQ_ERR_INVALID_ARGUMENT;
    std::vector<uint8_t> idx = {'B', 'A', 'I', 1};
    auto put64 = [&](uint64_t x) {
        for (int k = 0; k < 8; k++) idx.push_back((uint8_t)(x >> (8 * k)));
    };
    put32(idx, n_refs);
    for (uint32_t r = 0; r < n_refs; r++) {
        put32(idx, (uint32_t)bins[r].size());
        for (const auto &kv : bins[r]) {
            put32(idx, kv.first);
            put32(idx, (uint32_t)kv.second.size());
            for (const auto &c : kv.second) {
                put64(c.first);
                put64(c.second);
            }
        }
        put32(idx, (uint32_t)linear[r].size());
        uint64_t last = 0;
        for (uint64_t v : linear[r]) {
            if (v) last = v;
            put64(last);
        }
    }
    put64(n_no_coor);
    fwrite(idx.data(), 1, idx.size(), bi);
    fclose(bi);
    return NGSQ_OK;
}
