// gff_loader.h -- the gene model of the Genomic Features facet from a GFF file, parsed by several threads.
//
// What GenomicFeaturesFacet::try_from keeps of the file (src/qc/record_based/features.rs:270-355), with the openers of
// src/utils/formats/gff.rs:19-47 (plain or gzip by extension) and noodles-gff's records(): nine tab-separated columns, '#'
// lines are comments / directives, "##FASTA" ends the records.  A GENCODE annotation is 3.4 M lines / 1.4 GB of text: read
// line by line through gzgets into a vector<string> per line (round 5) it cost several times the scan of a 30x genome; here
// the text is mapped (or inflated into one buffer), cut at line ends into one piece per thread, and every piece is parsed
// in place with memchr -- the per-line rules are exactly the old ones, and the FIRST offending line of the file (not of a
// thread) is the one reported.
#pragma once

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <map>
#include <set>
#include <string>
#include <thread>
#include <vector>

struct GeneModel {
    std::vector<uint32_t> ref, name, start, stop;
    uint32_t role_name[5] = {0, 1, 2, 3, 4};
    std::string error;       // non-empty: what the reference would have bailed with
    double seconds = 0;      // reading + parsing
    uint64_t text_bytes = 0, lines = 0;
};

namespace gff_detail {

struct Piece {
    std::vector<uint32_t> ref, name, start, stop;
    uint64_t lines = 0;
    uint64_t err_at = ~0ull; // offset of the first offending line of the piece
    int err_kind = 0;        // 1 invalid record, 2 strand
    std::string err_value;
};

inline bool parse_u64(const char *p, const char *e, unsigned long long *out) { // usize::from_str: digits, an optional '+' in front
    if (p < e && *p == '+') p++;
    if (p == e) return false;
    unsigned long long v = 0;
    for (; p < e; p++) {
        if (*p < '0' || *p > '9') return false;
        if (v > (~0ull - 9) / 10) return false;
        v = v * 10 + (unsigned)(*p - '0');
    }
    *out = v;
    return true;
}

// lines of text[lo, hi) (lo is a line start; hi is a line start or the end)
inline void parse_piece(const char *text, uint64_t lo, uint64_t hi, const std::string (&feature_name)[5], const uint32_t (&role_name)[5],
                        const std::set<std::string> &primary, const std::map<std::string, uint32_t> &ref_index, Piece *out) {
    // the sequence column repeats for thousands of lines: the verdict for a name is looked up once
    std::string last_seq;
    bool last_valid = false, last_primary = false;
    int64_t last_ref = -1;
    for (uint64_t p = lo; p < hi;) {
        const char *nl = static_cast<const char *>(memchr(text + p, '\n', hi - p));
        const uint64_t line_end = nl ? (uint64_t)(nl - text) : hi;
        const uint64_t next = nl ? line_end + 1 : hi;
        uint64_t e = line_end;
        while (e > p && (text[e - 1] == '\r' || text[e - 1] == '\n')) e--;
        const uint64_t line_at = p;
        p = next;
        out->lines++;
        if (e == line_at) continue;
        if (text[line_at] == '#') continue; // ("##FASTA" was cut off before the pieces were made)
        // nine columns exactly: column k is [col[k], col[k + 1] - 1)
        const char *col[11];
        int nc = 1;
        col[0] = text + line_at;
        const char *le = text + e;
        for (const char *q = col[0];;) {
            const char *t = static_cast<const char *>(memchr(q, '\t', (size_t)(le - q)));
            if (!t) break;
            if (nc == 9) { // a tenth column
                nc = 10;
                break;
            }
            col[nc++] = t + 1;
            q = t + 1;
        }
        col[nc] = le + 1;
        auto fail = [&](int kind, std::string v) {
            if (out->err_at == ~0ull) {
                out->err_at = line_at;
                out->err_kind = kind;
                out->err_value = std::move(v);
            }
        };
        unsigned long long start = 0, stop = 0;
        if (nc != 9 || !parse_u64(col[3], col[4] - 1, &start) || !parse_u64(col[4], col[5] - 1, &stop) || start == 0 || stop < start ||
            stop > 0xFFFFFFFFull) {
            fail(1, ""); // result.unwrap(): features.rs:291
            return;
        }
        const size_t sl = (size_t)(col[1] - 1 - col[0]);
        if (!last_valid || last_seq.size() != sl || memcmp(last_seq.data(), col[0], sl) != 0) {
            last_seq.assign(col[0], sl);
            last_valid = true;
            last_primary = primary.count(last_seq) != 0;
            const auto it = ref_index.find(last_seq);
            last_ref = it == ref_index.end() ? -1 : (int64_t)it->second;
        }
        if (!last_primary) continue; // :300-304 only primary-assembly sequences get interval stores
        // :310-312: the strand of EVERY record on a primary sequence is parsed, '+' or '-' only
        const size_t strand_len = (size_t)(col[7] - 1 - col[6]);
        if (strand_len != 1 || (col[6][0] != '+' && col[6][0] != '-')) {
            fail(2, std::string(col[6], strand_len));
            return;
        }
        const size_t tl = (size_t)(col[3] - 1 - col[2]);
        int name = -1;
        for (int k = 0; k < 5 && name < 0; k++)
            if (feature_name[k].size() == tl && memcmp(feature_name[k].data(), col[2], tl) == 0) name = (int)role_name[k];
        if (name < 0) continue;
        if (last_ref < 0) continue; // a primary sequence this BAM does not have: never looked up
        out->ref.push_back((uint32_t)last_ref);
        out->name.push_back((uint32_t)name);
        out->start.push_back((uint32_t)start);
        out->stop.push_back((uint32_t)stop);
    }
}

} // namespace gff_detail

// `format`: "GFF" or "Gzipped GFF" (the caller has sniffed the extension: utils/formats/gff.rs:19-47)
inline GeneModel load_gff_parallel(const std::string &path, bool gzipped, const std::string (&feature_name)[5], const std::set<std::string> &primary,
                                   const std::map<std::string, uint32_t> &ref_index, int n_threads) {
    using namespace gff_detail;
    GeneModel m;
    const auto t0 = std::chrono::steady_clock::now();
    // roles that are configured with the same name are one name (the reference compares strings)
    for (int k = 0; k < 5; k++)
        for (int q = 0; q <= k; q++)
            if (feature_name[q] == feature_name[k]) {
                m.role_name[k] = (uint32_t)q;
                break;
            }
    const char *text = nullptr;
    uint64_t size = 0;
    char *inflated = nullptr; // (malloc: a vector would zero a gigabyte before the first byte is inflated into it)
    void *map = nullptr;
    uint64_t map_size = 0;
    if (gzipped) {
        // flate2's MultiGzDecoder: every member of the file, one after the other (zlib's gzread does the same)
        gzFile f = gzopen(path.c_str(), "rb");
        if (!f) {
            m.error = "opening GFF file: " + path + ": No such file or directory (os error 2)";
            return m;
        }
        gzbuffer(f, 1 << 20);
        struct stat st;
        size_t cap = (stat(path.c_str(), &st) == 0 ? (size_t)st.st_size * 12 : 0) + (1 << 20);
        inflated = static_cast<char *>(malloc(cap));
        size_t n = 0;
        for (;;) {
            if (n == cap) {
                cap *= 2;
                char *bigger = static_cast<char *>(realloc(inflated, cap));
                if (!bigger) break;
                inflated = bigger;
            }
            const int r = inflated ? gzread(f, inflated + n, (unsigned)std::min<size_t>(cap - n, 1u << 30)) : -1;
            if (r < 0) {
                gzclose(f);
                free(inflated);
                m.error = "opening GFF file: " + path + ": corrupt deflate stream";
                return m;
            }
            if (r == 0) break;
            n += (size_t)r;
        }
        gzclose(f);
        text = inflated;
        size = n;
    } else {
        const int fd = open(path.c_str(), O_RDONLY | O_CLOEXEC);
        struct stat st;
        if (fd < 0 || fstat(fd, &st) != 0) {
            if (fd >= 0) close(fd);
            m.error = "opening GFF file: " + path + ": No such file or directory (os error 2)";
            return m;
        }
        size = (uint64_t)st.st_size;
        if (size) {
            map = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
            if (map == MAP_FAILED) {
                close(fd);
                m.error = "opening GFF file: " + path + ": could not map the file";
                return m;
            }
            map_size = size;
            text = static_cast<const char *>(map);
        }
        close(fd);
    }
    // "##FASTA" at the start of a line ends the records
    if (size >= 7) {
        if (memcmp(text, "##FASTA", 7) == 0) size = 0;
        else if (const void *h = memmem(text, size, "\n##FASTA", 8)) size = (uint64_t)(static_cast<const char *>(h) - text) + 1;
    }
    m.text_bytes = size;
    const int nt = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)std::max(1, n_threads), size >> 16));
    std::vector<uint64_t> cut((size_t)nt + 1, size);
    cut[0] = 0;
    for (int t = 1; t < nt; t++) {
        uint64_t p = size / (uint64_t)nt * (uint64_t)t;
        if (p < cut[(size_t)t - 1]) p = cut[(size_t)t - 1];
        const char *nl = p < size ? static_cast<const char *>(memchr(text + p, '\n', size - p)) : nullptr;
        cut[(size_t)t] = nl ? (uint64_t)(nl - text) + 1 : size;
    }
    std::vector<Piece> pieces((size_t)nt);
    {
        std::vector<std::thread> th;
        for (int t = 0; t < nt; t++)
            th.emplace_back([&, t] { parse_piece(text, cut[(size_t)t], cut[(size_t)t + 1], feature_name, m.role_name, primary, ref_index, &pieces[(size_t)t]); });
        for (auto &x : th) x.join();
    }
    size_t total = 0;
    for (const Piece &pc : pieces) {
        if (pc.err_at != ~0ull && m.error.empty()) { // pieces are in file order: the first one with an error holds the file's first
            if (pc.err_kind == 2) {
                m.error = "attempted to parse strand from value: " + pc.err_value;
            } else {
                uint64_t line_no = 1;
                for (const char *q = text; q < text + pc.err_at;) {
                    const char *nl = static_cast<const char *>(memchr(q, '\n', (size_t)(text + pc.err_at - q)));
                    if (!nl) break;
                    line_no++;
                    q = nl + 1;
                }
                m.error = "invalid GFF record on line " + std::to_string(line_no) + " of " + path;
            }
        }
        if (!m.error.empty()) break;
        total += pc.ref.size();
        m.lines += pc.lines;
    }
    if (m.error.empty()) {
        m.ref.reserve(total);
        m.name.reserve(total);
        m.start.reserve(total);
        m.stop.reserve(total);
        for (const Piece &pc : pieces) {
            m.ref.insert(m.ref.end(), pc.ref.begin(), pc.ref.end());
            m.name.insert(m.name.end(), pc.name.begin(), pc.name.end());
            m.start.insert(m.start.end(), pc.start.begin(), pc.start.end());
            m.stop.insert(m.stop.end(), pc.stop.begin(), pc.stop.end());
        }
    }
    if (map) munmap(map, map_size);
    free(inflated);
    m.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return m;
}
