// synth_bam.cpp -- write the synthetic records of include/ngsq_shared.h as a real BGZF BAM
// (+ its BAI: binning and linear index, SAM spec 5.2) so that the file-to-JSON path of `ngs qc`,
// including the region queries of `-n`, can be measured at size.
// Bench / test utility (the reference's `ngs generate` writes FASTQ from a FASTA with an
// unseeded RNG, src/generate/command.rs:59-131, and cannot produce these files).
#include <zlib.h>

#include <algorithm>
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

// ---- what an aligner's output carries around the generator's fields (NGSQ_SYNTH_FILE_ALIGNER) ----------------------
// Shapes and sizes follow bwa-mem + samtools fixmate / markdup output on an Illumina NovaSeq run: every value is a pure
// function of (seed, record index), like the records themselves.
void put_z(std::vector<uint8_t> &v, const char tag[2], const char *text) { // XX:Z:<text>
    v.push_back((uint8_t)tag[0]);
    v.push_back((uint8_t)tag[1]);
    v.push_back('Z');
    v.insert(v.end(), text, text + strlen(text) + 1);
}
void put_c(std::vector<uint8_t> &v, const char tag[2], uint32_t x) { // XX:i:<x> in its smallest type, as htslib writes it
    v.push_back((uint8_t)tag[0]);
    v.push_back((uint8_t)tag[1]);
    if (x < 256) {
        v.push_back('C');
        v.push_back((uint8_t)x);
    } else if (x < 65536) {
        v.push_back('S');
        put16(v, x);
    } else {
        v.push_back('I');
        put32(v, x);
    }
}

// the read name: instrument, run, flowcell, lane, tile, x, y -- the two reads of a pair (records 2k, 2k+1) share it
int illumina_name(const ngsq_synth_config &cfg, uint64_t i, char *out, size_t cap) {
    const uint64_t h = ngsq_synth_hash(cfg.seed, i >> 1, NGSQ_KEY_NAME, 0);
    const uint32_t lane = 1 + (uint32_t)(h & 3), surface = 1 + (uint32_t)((h >> 2) & 1), swath = 1 + (uint32_t)((h >> 3) % 6),
                   tile = 1 + (uint32_t)((h >> 8) % 78), x = 1000 + (uint32_t)((h >> 16) % 31624), y = 1000 + (uint32_t)((h >> 32) % 36000);
    return snprintf(out, cap, "A00741:215:HG7WKDSXX:%u:%u%u%02u:%u:%u", lane, surface, swath, tile, x, y) + 1;
}

void append_aux(std::vector<uint8_t> &out, const ngsq_synth_config &cfg, uint64_t i, const ngsq_synth_record &r, const uint32_t *cig,
                uint32_t n_cig) {
    const uint64_t h = ngsq_synth_hash(cfg.seed, i, NGSQ_KEY_AUX, 0), h2 = ngsq_synth_hash(cfg.seed, i, NGSQ_KEY_AUX, 1);
    const bool mapped = !(r.flag & 0x4u), paired = r.flag & 0x1u, mate_mapped = paired && !(r.flag & 0x8u);
    char text[256];
    const uint32_t lane = 1 + (uint32_t)(ngsq_synth_hash(cfg.seed, i >> 1, NGSQ_KEY_NAME, 0) & 3);
    if (mapped) {
        // NM: 0 (70 %), 1 (20 %), 2 (6 %), 3..8; MD: the matched runs between the mismatches, ^<bases> for a deletion
        const uint32_t d = (uint32_t)(h & 0xFFFF);
        uint32_t nm = d < 45875u ? 0 : d < 58982u ? 1 : d < 62915u ? 2 : 3 + (uint32_t)((h >> 16) % 6);
        uint32_t m_total = 0, indel = 0;
        for (uint32_t k = 0; k < n_cig; k++) {
            const uint32_t op = cig[k] & 15, len = cig[k] >> 4;
            if (op == 0) m_total += len;
            if (op == 1 || op == 2) indel += len;
        }
        if (nm > m_total) nm = m_total;
        int at = 0;
        uint32_t left = nm, hh = (uint32_t)(h >> 24);
        for (uint32_t k = 0; k < n_cig; k++) {
            const uint32_t op = cig[k] & 15;
            uint32_t len = cig[k] >> 4;
            if (op == 2) {
                at += snprintf(text + at, sizeof text - (size_t)at, "^");
                for (uint32_t q = 0; q < len && at < 200; q++) text[at++] = "ACGT"[(hh >> (2 * (q & 7))) & 3];
                text[at] = 0;
            }
            if (op != 0) continue;
            // the last M run takes the mismatches that are left
            bool last_m = true;
            for (uint32_t q = k + 1; q < n_cig; q++) last_m = last_m && (cig[q] & 15) != 0;
            uint32_t here = last_m ? left : std::min(left, (uint32_t)(hh & 1));
            left -= here;
            while (here && len > 1 && at < 200) {
                const uint32_t run = (hh = hh * 1664525u + 1013904223u, hh >> 8) % (len - 1);
                at += snprintf(text + at, sizeof text - (size_t)at, "%u%c", run, "ACGT"[hh & 3]);
                len -= run + 1;
                here--;
            }
            at += snprintf(text + at, sizeof text - (size_t)at, "%u", len);
        }
        put_c(out, "NM", nm + indel);
        put_z(out, "MD", text);
    }
    if (paired) { // samtools fixmate -m: the mate's CIGAR and mapping quality
        const uint32_t l = r.l_seq, a = 1 + (uint32_t)((h2 >> 8) % 40);
        if (!mate_mapped) snprintf(text, sizeof text, "*");
        else if ((h2 & 0xFF) < 26) snprintf(text, sizeof text, "%uS%uM", a, l > a ? l - a : 1);
        else snprintf(text, sizeof text, "%uM", l);
        put_z(out, "MC", text);
    }
    if (mapped) {
        uint32_t m_total = 0;
        for (uint32_t k = 0; k < n_cig; k++)
            if ((cig[k] & 15) == 0) m_total += cig[k] >> 4;
        const uint32_t pen = 5 * (uint32_t)((h & 0xFFFF) < 45875u ? 0 : 1 + (h >> 16) % 3);
        put_c(out, "AS", m_total > pen ? m_total - pen : 0);
        put_c(out, "XS", (h2 >> 16 & 0xFF) < 154 ? 0 : 19 + (uint32_t)((h2 >> 24) % 120));
    }
    if (mate_mapped) put_c(out, "MQ", (h2 >> 40 & 0xFF) < 200 ? 60 : (uint32_t)((h2 >> 48) % 60));
    snprintf(text, sizeof text, "HG7WKDSXX.L00%u.SJNORM0415", lane);
    put_z(out, "RG", text);
    const uint64_t h3 = ngsq_synth_hash(cfg.seed, i, NGSQ_KEY_AUX, 2);
    if (mapped && ((r.flag & 0x800u) || (h3 & 0xFFFF) < 983u)) { // SA: supplementary records and 1.5 % of the others
        char cg[64];
        const uint32_t l = r.l_seq, a = 20 + (uint32_t)((h3 >> 16) % (l > 60 ? l - 40 : 1));
        snprintf(cg, sizeof cg, "%uS%uM", a, l > a ? l - a : 1);
        snprintf(text, sizeof text, "chr%u,%u,%c,%s,%u,%u;", 1 + (uint32_t)((h3 >> 32) % 22), 10000 + (uint32_t)((h3 >> 24) % 200000000u), (h3 >> 40) & 1 ? '+' : '-', cg,
                 (uint32_t)((h3 >> 44) % 61), (uint32_t)((h3 >> 52) % 4));
        put_z(out, "SA", text);
    }
    if (mapped && (h3 >> 56 & 0xFF) < 8) { // XA: one to three alternative hits on 3 %
        int at = 0;
        uint64_t x = h3;
        for (uint32_t k = 0, n = 1 + (uint32_t)((h3 >> 20) % 3); k < n; k++) {
            x = ngsq_mix64(x);
            at += snprintf(text + at, sizeof text - (size_t)at, "chr%u,%c%u,%uM,%u;", 1 + (uint32_t)(x % 22), (x >> 8) & 1 ? '+' : '-', 10000 + (uint32_t)((x >> 16) % 200000000u), r.l_seq,
                           (uint32_t)((x >> 48) % 5));
        }
        put_z(out, "XA", text);
    }
    if (((h3 >> 48) & 0xFF) < 5) { // a B array on 2 %: per-base values of some downstream tool, 8..40 of them
        const uint32_t n = 8 + (uint32_t)((h3 >> 8) % 33);
        out.push_back('Z');
        out.push_back('B');
        out.push_back('B');
        out.push_back('S');
        put32(out, n);
        uint64_t x = h3;
        for (uint32_t k = 0; k < n; k++) {
            if ((k & 3) == 0) x = ngsq_mix64(x);
            put16(out, (uint32_t)(x >> (16 * (k & 3))) & 0xFFFF);
        }
    }
}

void append_record(std::vector<uint8_t> &out, const ngsq_synth_config &cfg, uint64_t i, RecIdx *ix, std::vector<uint8_t> &aux) {
    ngsq_synth_record r;
    ngsq_synth_record_at(&cfg, i, &r);
    ix->uoff = (uint32_t)out.size();
    const bool aligner = cfg.file_style & NGSQ_SYNTH_FILE_ALIGNER;
    char name[64];
    const int ln = aligner ? illumina_name(cfg, i, name, sizeof name) : snprintf(name, sizeof name, "r%llu", (unsigned long long)i) + 1;
    const uint32_t l = r.l_seq;
    uint32_t cig[NGSQ_SYNTH_MAX_OPS] = {r.cigar[0], r.cigar[1], r.cigar[2]}, n_cig = r.n_cigar;
    // (NGSQ_SYNTH_FILE_CIGAR_MIX: the records carry an aligner's CIGAR mix already -- ngsq_synth_record_at, shared with the device generator)
    uint64_t span = 0;
    for (uint32_t k = 0; k < n_cig; k++)
        if ((0x18Du >> (cig[k] & 15)) & 1u) span += cig[k] >> 4;
    ix->ref = r.ref_id;
    ix->pos = r.pos;
    ix->end = r.pos + (int32_t)(span ? span : 1);
    aux.clear();
    if (aligner) append_aux(aux, cfg, i, r, cig, n_cig);
    const uint32_t block = 32 + (uint32_t)ln + 4 * n_cig + (l + 1) / 2 + l + (uint32_t)aux.size();
    put32(out, block);
    put32(out, (uint32_t)r.ref_id);
    put32(out, (uint32_t)r.pos);
    out.push_back((uint8_t)ln);
    out.push_back(r.mapq);
    put16(out, reg2bin(r.pos, r.pos + (int64_t)(span ? span : 1)));
    put16(out, n_cig);
    put16(out, r.flag);
    put32(out, l);
    put32(out, (uint32_t)r.mate_ref_id);
    // (the mate's position: the files of rounds 1-3 say -1; an aligner's say where the mate starts)
    put32(out, aligner && (r.flag & 0x1u) ? (uint32_t)std::max<int64_t>(0, (int64_t)r.pos + r.tlen - (r.tlen > 0 ? (int64_t)l : -(int64_t)l)) : (uint32_t)-1);
    put32(out, (uint32_t)r.tlen);
    out.insert(out.end(), name, name + ln);
    for (uint32_t k = 0; k < n_cig; k++) put32(out, cig[k]);
    for (uint32_t j = 0; j < (l + 1) / 2; j++) out.push_back(ngsq_synth_seq_byte_of(&cfg, i, l, j, &r)); // (the record's fields are worked out once)
    for (uint32_t j = 0; j < l; j++) out.push_back(ngsq_synth_qual_byte(&cfg, i, l, j));
    out.insert(out.end(), aux.begin(), aux.end());
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

extern "C" int ngsq_synth_genome_room(const uint32_t *genome_len, uint32_t n, uint64_t *room) {
    if (!genome_len || !room) return NGSQ_ERR_INVALID_ARGUMENT;
    room[0] = 0;
    for (uint32_t r = 0; r < n; r++) room[r + 1] = room[r] + ngsq_synth_room_of(genome_len[r]);
    return NGSQ_OK;
}

static int write_bam(const ngsq_synth_config *cfg, const char *const *ref_names, const char *path, uint64_t n_records, int level, int n_threads);

extern "C" int ngsq_synth_write_bam(const ngsq_synth_config *cfg, const char *path, uint64_t n_records, int level, int n_threads) {
    if (cfg && cfg->genome_n) return NGSQ_ERR_INVALID_ARGUMENT; // (a GENOME-mode file needs its sequences' names: ngsq_synth_write_bam_named)
    return write_bam(cfg, nullptr, path, n_records, level, n_threads);
}

extern "C" int ngsq_synth_write_bam_named(const ngsq_synth_config *cfg, const char *const *ref_names, const char *path, uint64_t n_records, int level,
                                          int n_threads) {
    if (!cfg || !cfg->genome_n || !cfg->genome_len || !cfg->genome_room || !ref_names) return NGSQ_ERR_INVALID_ARGUMENT;
    return write_bam(cfg, ref_names, path, n_records, level, n_threads);
}

static int write_bam(const ngsq_synth_config *cfg, const char *const *ref_names, const char *path, uint64_t n_records, int level, int n_threads) {
    if (!cfg || !path) return NGSQ_ERR_INVALID_ARGUMENT;
    static const char *two_names[2] = {"chr1", "chr2"};
    const uint32_t two_lens[2] = {cfg->ref_len, 242193529u};
    const char *const *names = ref_names ? ref_names : two_names;
    const uint32_t *lens = ref_names ? cfg->genome_len : two_lens;
    const uint32_t n_refs = ref_names ? cfg->genome_n : (cfg->n_refs >= 2 ? 2 : 1);
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
    // Records per group = per BGZF block.  Plain files: 100 (27 KB of data at 150 bases -- the files of rounds 1-3, kept as they
    // were).  Aligner-style files fill their blocks as htslib does: as many whole records as fit 0xff00 bytes of data, here
    // by an estimate of the record size; a group that comes out larger is written as two blocks.
    constexpr size_t BLOCK_DATA = 0xff00;
    uint64_t group = 100;
    if (cfg->file_style & NGSQ_SYNTH_FILE_ALIGNER) {
        const uint64_t l = cfg->mode == NGSQ_SYNTH_FIXED ? cfg->read_len : (cfg->min_len + cfg->max_len) / 2;
        group = std::max<uint64_t>(1, BLOCK_DATA * 96 / 100 / (36 + 39 + 12 + 85 + l + (l + 1) / 2));
    }
    const uint64_t n_groups = (n_records + group - 1) / group;
    const uint64_t wave = (uint64_t)nt * 64;
    struct Group {
        std::vector<uint8_t> bytes;       // its BGZF blocks, one after the other
        std::vector<uint32_t> block_size; // of each
        std::vector<RecIdx> recs;         // uoff: offset in the data of block `blk`
        std::vector<uint32_t> blk;
    };
    for (uint64_t g0 = 0; g0 < n_groups && ok; g0 += wave) {
        const uint64_t g1 = std::min(n_groups, g0 + wave);
        std::vector<Group> groups(g1 - g0);
        std::atomic<uint64_t> next{g0};
        std::atomic<int> bad{0};
        auto worker = [&]() {
            std::vector<uint8_t> raw, sc, aux;
            std::vector<size_t> rec_at;
            for (;;) {
                const uint64_t g = next.fetch_add(1);
                if (g >= g1) break;
                raw.clear();
                rec_at.clear();
                Group &gr = groups[g - g0];
                for (uint64_t i = g * group; i < std::min(n_records, (g + 1) * group); i++) {
                    gr.recs.emplace_back();
                    rec_at.push_back(raw.size());
                    append_record(raw, *cfg, i, &gr.recs.back(), aux);
                }
                rec_at.push_back(raw.size());
                // cut into blocks of whole records (one, unless the estimate above was too low; a single record larger than a
                // block's data is cut anywhere, as htslib does)
                size_t lo = 0, r_lo = 0;
                while (lo < raw.size()) {
                    size_t r_hi = r_lo;
                    while (r_hi + 1 < rec_at.size() && rec_at[r_hi + 1] - lo <= BLOCK_DATA) r_hi++;
                    const size_t hi = r_hi > r_lo ? rec_at[r_hi] : std::min(raw.size(), lo + BLOCK_DATA);
                    for (size_t r = r_lo; r < gr.recs.size() && rec_at[r] < hi; r++) {
                        if (rec_at[r] < lo) continue; // (the record the previous block was cut in)
                        gr.recs[r].uoff = (uint32_t)(rec_at[r] - lo);
                        gr.blk.resize(gr.recs.size());
                        gr.blk[r] = (uint32_t)gr.block_size.size();
                    }
                    // deflate into a memory "file"
                    z_stream zs;
                    memset(&zs, 0, sizeof zs);
                    if (deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) { bad = 1; break; }
                    sc.resize(deflateBound(&zs, (uLong)(hi - lo)) + 64);
                    zs.next_in = raw.data() + lo;
                    zs.avail_in = (uInt)(hi - lo);
                    zs.next_out = sc.data() + 18;
                    zs.avail_out = (uInt)(sc.size() - 18);
                    const int rc = deflate(&zs, Z_FINISH);
                    const size_t clen = zs.total_out;
                    deflateEnd(&zs);
                    if (rc != Z_STREAM_END || clen + 26 > 65536 || hi - lo > 65536) { bad = 1; break; }
                    static const uint8_t hd[16] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 'B', 'C', 2, 0};
                    memcpy(sc.data(), hd, 16);
                    const size_t bsize = clen + 26;
                    sc[16] = (uint8_t)((bsize - 1) & 0xFF);
                    sc[17] = (uint8_t)((bsize - 1) >> 8);
                    uint8_t *tail = sc.data() + 18 + clen;
                    const uint32_t crc = (uint32_t)crc32(0L, raw.data() + lo, (uInt)(hi - lo)), isz = (uint32_t)(hi - lo);
                    for (int k = 0; k < 4; k++) tail[k] = (uint8_t)(crc >> (8 * k)), tail[4 + k] = (uint8_t)(isz >> (8 * k));
                    gr.bytes.insert(gr.bytes.end(), sc.begin(), sc.begin() + (ptrdiff_t)bsize);
                    gr.block_size.push_back((uint32_t)bsize);
                    lo = hi;
                    while (r_lo < gr.recs.size() && rec_at[r_lo] < lo) r_lo++;
                }
            }
        };
        std::vector<std::thread> pool;
        for (int t = 1; t < nt; t++) pool.emplace_back(worker);
        worker();
        for (auto &t : pool) t.join();
        if (bad) ok = false;
        for (size_t k = 0; k < groups.size() && ok; k++) {
            const Group &gr = groups[k];
            if (fwrite(gr.bytes.data(), 1, gr.bytes.size(), f) != gr.bytes.size()) ok = false;
            std::vector<uint64_t> start(gr.block_size.size() + 1, file_off);
            for (size_t q = 0; q < gr.block_size.size(); q++) start[q + 1] = start[q] + gr.block_size[q];
            const uint64_t next_off = start.back();
            auto voff = [&](size_t j) { return (start[gr.blk[j]] << 16) | gr.recs[j].uoff; };
            for (size_t j = 0; j < gr.recs.size(); j++) index_record(gr.recs[j], voff(j), j + 1 < gr.recs.size() ? voff(j + 1) : next_off << 16);
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
