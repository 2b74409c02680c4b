// results.cpp -- host finalize of the hot path: the floating-point summaries the
// reference computes in summarize()/teardown()/aggregate() from integer state,
// and the `Results` JSON (src/qc/results.rs:23-60).
//
// All f64 / f32 arithmetic happens here on the host, in the reference's
// operation order, from the integer arrays the kernels produced:
//   general.rs:126-153, template_length.rs:89-100, gc_content.rs:102-122,
//   coverage.rs:206-234 + :264-287 (f32), edits.rs:346-353,
//   utils/histogram.rs:258-337 (mean, percentile).
#include <charconv>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/ngsq.h"
#include "context.h"

using namespace ngsq;

namespace {

typedef unsigned long long u64;

// utils/histogram.rs:258-269
double hist_mean(const u64 *v, size_t n) {
    double sum = 0.0, denominator = 0.0;
    for (size_t i = 0; i < n; i++) {
        denominator += (double)v[i];
        sum += (double)(v[i] * (u64)i);
    }
    return sum / denominator;
}

// utils/histogram.rs:272-337; returns false for Ok(None) or when the reference would panic
bool hist_percentile(const u64 *v, size_t n, double percentile, double *out) {
    u64 num_items = 0;
    for (size_t i = 0; i < n; i++) num_items += v[i];
    if (num_items == 0) return false;
    const double needed_items = percentile * (double)num_items;
    double collected_items = 0.0;
    size_t index = 0;
    for (;;) {
        if (index >= n) return false;
        collected_items += (double)v[index];
        if (collected_items > needed_items) {
            *out = (double)index;
            return true;
        }
        if (collected_items == needed_items) {
            const size_t lowest = index;
            index += 1;
            while (index < n && v[index] == 0) index += 1;
            if (index >= n) return false;
            *out = (double)lowest + ((double)(index - lowest) / 2.0);
            return true;
        }
        index += 1;
    }
}

struct Json {
    std::string s;
    void indent(int d) { s.append((size_t)d * 2, ' '); }
    void key(int d, const char *k) {
        indent(d);
        s += '"';
        s += k;
        s += "\": ";
    }
    void u(u64 v) { s += std::to_string(v); }
    // shortest round-trip digits, laid out by ryu's rules (serde_json)
    template <typename F> void flt(F v) {
        if (std::isnan(v) || std::isinf(v)) {
            s += "null";
            return;
        }
        if (v == 0) {
            s += std::signbit(v) ? "-0.0" : "0.0";
            return;
        }
        char e[64];
        auto r = std::to_chars(e, e + sizeof e, v, std::chars_format::scientific);
        *r.ptr = 0;
        char digits[32];
        int nd = 0;
        bool neg = false;
        const char *q = e;
        if (*q == '-') {
            neg = true;
            q++;
        }
        for (; *q && *q != 'e'; q++)
            if (*q != '.') digits[nd++] = *q;
        const int exp10 = atoi(q + 1);
        const int k = exp10 - (nd - 1), kk = nd + k;
        const int hi = sizeof(F) == 4 ? 13 : 16, lo = sizeof(F) == 4 ? -6 : -5;
        if (neg) s += '-';
        if (0 <= k && kk <= hi) {
            s.append(digits, (size_t)nd);
            s.append((size_t)k, '0');
            s += ".0";
        } else if (0 < kk && kk <= hi) {
            s.append(digits, (size_t)kk);
            s += '.';
            s.append(digits + kk, (size_t)(nd - kk));
        } else if (lo < kk && kk <= 0) {
            s += "0.";
            s.append((size_t)(-kk), '0');
            s.append(digits, (size_t)nd);
        } else {
            s += digits[0];
            if (nd > 1) {
                s += '.';
                s.append(digits + 1, (size_t)(nd - 1));
            }
            s += 'e';
            s += std::to_string(kk - 1);
        }
    }
    void u_field(int d, const char *k, u64 v, bool last) {
        key(d, k);
        u(v);
        s += last ? "\n" : ",\n";
    }
    void f_field(int d, const char *k, double v, bool last) {
        key(d, k);
        flt(v);
        s += last ? "\n" : ",\n";
    }
    // utils/histogram.rs:152-159 field order
    void histogram(int d, const u64 *v, u64 stop) {
        s += "{\n";
        key(d + 1, "values");
        s += "[\n";
        for (u64 i = 0; i <= stop; i++) {
            indent(d + 2);
            u(v[i]);
            s += i == stop ? "\n" : ",\n";
        }
        indent(d + 1);
        s += "],\n";
        u_field(d + 1, "range_start", 0, false);
        u_field(d + 1, "range_stop", stop, true);
        indent(d);
        s += "}";
    }
    void close(int d, bool last) {
        indent(d);
        s += last ? "}\n" : "},\n";
    }
    void cigar_map(int d, const char *k, const u64 *ops, bool last) {
        static const char letters[] = "MIDNSHP=X";
        static const int order[9] = {7, 2, 5, 1, 0, 3, 6, 4, 8}; // = D H I M N P S X (sorted)
        key(d, k);
        int n = 0;
        for (int i = 0; i < 9; i++) n += ops[i] != 0;
        if (!n) {
            s += last ? "{}\n" : "{},\n";
            return;
        }
        s += "{\n";
        int done = 0;
        for (int j = 0; j < 9; j++) {
            const int op = order[j];
            if (!ops[op]) continue;
            const char name[2] = {letters[op], 0};
            u_field(d + 1, name, ops[op], ++done == n);
        }
        close(d, last);
    }
};

} // namespace

extern "C" int64_t ngsq_results_json(const ngsq_ctx *c, const char *const *ref_names, char *buf, size_t cap) {
    if (!c) return NGSQ_ERR_INVALID_ARGUMENT;
    if (!c->finalized) return NGSQ_ERR_STATE;
    const uint32_t facets = c->cfg.facets;
    const u64 *cnt = c->h_counters.data();
    Json j;
    j.s.reserve(1 << 20);
    j.s += "{\n";

    j.key(1, "general");
    if (facets & NGSQ_FACET_GENERAL) {
        const u64 *g = cnt + C_GENERAL;
        static const char *names[16] = {"total", "unmapped", "duplicate", "primary", "secondary", "supplementary",
                                        "primary_mapped", "primary_duplicate", "paired", "read_1", "read_2",
                                        "proper_pair", "singleton", "mate_mapped",
                                        "mate_reference_sequence_id_mismatch",
                                        "mate_reference_sequence_id_mismatch_hq"};
        j.s += "{\n";
        j.key(2, "records");
        j.s += "{\n";
        for (int k = 0; k < 3; k++) j.u_field(3, names[k], g[k], false);
        j.key(3, "designation");
        j.s += "{\n";
        for (int k = 3; k < 6; k++) j.u_field(4, names[k], g[k], k == 5);
        j.close(3, false);
        for (int k = 6; k < 16; k++) j.u_field(3, names[k], g[k], k == 15);
        j.close(2, false);
        j.key(2, "cigar");
        j.s += "{\n";
        j.cigar_map(3, "read_one_cigar_ops", cnt + C_CIGAR1, false);
        j.cigar_map(3, "read_two_cigar_ops", cnt + C_CIGAR2, true);
        j.close(2, false);
        // general.rs:126-153
        const double total = (double)g[0];
        j.key(2, "summary");
        j.s += "{\n";
        j.f_field(3, "duplication_pct", (double)g[2] / total * 100.0, false);
        j.f_field(3, "mapped_pct", (1.0 - (double)g[1] / total) * 100.0, false);
        j.f_field(3, "mate_reference_sequence_id_mismatch_pct", (double)g[14] / total * 100.0, false);
        j.f_field(3, "mate_reference_sequence_id_mismatch_hq_pct", (double)g[15] / total * 100.0, true);
        j.close(2, true);
        j.close(1, false);
    } else {
        j.s += "null,\n";
    }

    j.key(1, "features");
    if (facets & NGSQ_FACET_FEATURES) {
        // features/metrics.rs:10-80 (declaration order); summary: features.rs:244-262
        const u64 *f = cnt + C_FEAT;
        j.s += "{\n";
        j.key(2, "exonic_translation_regions");
        j.s += "{\n";
        j.u_field(3, "utr_five_prime_count", f[F_UTR5], false);
        j.u_field(3, "utr_three_prime_count", f[F_UTR3], false);
        j.u_field(3, "coding_sequence_count", f[F_CDS], true);
        j.close(2, false);
        j.key(2, "gene_regions");
        j.s += "{\n";
        j.u_field(3, "intergenic_count", f[F_INTERGENIC], false);
        j.u_field(3, "exonic_count", f[F_EXONIC], false);
        j.u_field(3, "intronic_count", f[F_INTRONIC], true);
        j.close(2, false);
        j.key(2, "records");
        j.s += "{\n";
        j.u_field(3, "processed", f[F_PROCESSED], false);
        j.u_field(3, "ignored_flags", f[F_IGN_FLAGS], false);
        j.u_field(3, "ignored_nonprimary_chromosome", f[F_IGN_NONPRIMARY], true);
        j.close(2, false);
        const double denom = (double)(f[F_IGN_FLAGS] + f[F_IGN_NONPRIMARY] + f[F_PROCESSED]);
        j.key(2, "summary");
        j.s += "{\n";
        j.f_field(3, "ignored_flags_pct", ((double)f[F_IGN_FLAGS] / denom) * 100.0, false);
        j.f_field(3, "ignored_nonprimary_chromosome_pct", ((double)f[F_IGN_NONPRIMARY] / denom) * 100.0, true);
        j.close(2, true);
        j.close(1, false);
    } else {
        j.s += "null,\n";
    }

    j.key(1, "gc_content");
    if (facets & NGSQ_FACET_GC_CONTENT) {
        const u64 gc = cnt[C_GC_GC], at = cnt[C_GC_AT], other = cnt[C_GC_OTHER];
        const u64 processed = cnt[C_GC_PROCESSED], ign_f = cnt[C_GC_IGN_FLAGS], ign_s = cnt[C_GC_IGN_SHORT];
        j.s += "{\n";
        j.key(2, "histogram");
        j.histogram(2, cnt + OFF_GC_HIST, 100);
        j.s += ",\n";
        j.key(2, "nucleobases");
        j.s += "{\n";
        j.u_field(3, "total_gc_count", gc, false);
        j.u_field(3, "total_at_count", at, false);
        j.u_field(3, "total_other_count", other, true);
        j.close(2, false);
        j.key(2, "records");
        j.s += "{\n";
        j.u_field(3, "processed", processed, false);
        j.u_field(3, "ignored_flags", ign_f, false);
        j.u_field(3, "ignored_too_short", ign_s, true);
        j.close(2, false);
        // gc_content.rs:102-122
        const double denom = (double)(ign_f + ign_s + processed);
        j.key(2, "summary");
        j.s += "{\n";
        j.f_field(3, "gc_content_pct", ((double)gc / (double)(gc + at + other)) * 100.0, false);
        j.f_field(3, "ignored_flags_pct", ((double)ign_f / denom) * 100.0, false);
        j.f_field(3, "ignored_too_short_pct", ((double)ign_s / denom) * 100.0, true);
        j.close(2, true);
        j.close(1, false);
    } else {
        j.s += "null,\n";
    }

    j.key(1, "template_length");
    if (facets & NGSQ_FACET_TEMPLATE_LENGTH) {
        const u64 *h = cnt + c->st.off_tlen;
        const u64 processed = cnt[C_TLEN_PROCESSED], ignored = cnt[C_TLEN_IGNORED];
        j.s += "{\n";
        j.key(2, "histogram");
        j.histogram(2, h, c->st.tlen_cap);
        j.s += ",\n";
        j.key(2, "records");
        j.s += "{\n";
        j.u_field(3, "processed", processed, false);
        j.u_field(3, "ignored", ignored, true);
        j.close(2, false);
        // template_length.rs:89-100
        const double denom = (double)processed + (double)ignored;
        j.key(2, "summary");
        j.s += "{\n";
        j.f_field(3, "template_length_unknown_pct", ((double)h[0] / denom) * 100.0, false);
        j.f_field(3, "template_length_out_of_range_pct", ((double)ignored / denom) * 100.0, true);
        j.close(2, true);
        j.close(1, false);
    } else {
        j.s += "null,\n";
    }

    j.key(1, "quality_scores");
    if (facets & NGSQ_FACET_QUALITY_SCORE) {
        // quality_scores.rs:39-42: key i exists iff some record reached cycle i; every
        // visit adds exactly one count to the row, so "row sum > 0" is equivalent.
        const u64 *q = cnt + c->st.off_qual;
        std::vector<uint32_t> rows;
        for (uint32_t i = 0; i < c->st.max_read_len; i++) {
            u64 sum = 0;
            for (uint32_t b = 0; b < QUAL_BINS; b++) sum += q[(size_t)i * QUAL_BINS + b];
            if (sum) rows.push_back(i);
        }
        j.s += "{\n";
        j.key(2, "scores");
        if (rows.empty()) {
            j.s += "{}\n";
        } else {
            j.s += "{\n";
            for (size_t k = 0; k < rows.size(); k++) {
                j.key(3, std::to_string(rows[k] + 1).c_str());
                j.histogram(3, q + (size_t)rows[k] * QUAL_BINS, NGSQ_MAX_SCORE);
                j.s += k + 1 == rows.size() ? "\n" : ",\n";
            }
            j.close(2, true);
        }
        j.close(1, false);
    } else {
        j.s += "null,\n";
    }

    j.key(1, "coverage");
    if (facets & NGSQ_FACET_COVERAGE) {
        const uint32_t nr = c->st.n_refs, cap = c->st.cov_cap;
        std::vector<uint32_t> seqs;
        for (uint32_t r = 0; r < nr; r++)
            if (c->depth_off[r] != NO_DEPTH && cnt[c->st.off_seen + r]) seqs.push_back(r);
        // coverage.rs:232-246 per sequence, then the global distribution
        std::vector<double> mean(nr, 0.0), median(nr, 0.0), mom(nr, 0.0);
        std::vector<u64> dist(cap + 1, 0);
        u64 total_positions = 0;
        for (uint32_t r : seqs) {
            const u64 *h = c->h_cov_hist.data() + (uint64_t)r * (cap + 2);
            mean[r] = hist_mean(h, cap + 1);
            double m = std::nan("");
            hist_percentile(h, cap + 1, 0.5, &m);
            median[r] = m;
            mom[r] = median[r] / mean[r];
            for (uint32_t i = 0; i <= cap; i++) dist[i] += h[i];
        }
        for (uint32_t i = 0; i <= cap; i++) total_positions += dist[i];      // coverage.rs:266
        for (uint32_t r : seqs) total_positions += c->h_cov_hist[(uint64_t)r * (cap + 2) + cap + 1]; // :270-272
        j.s += "{\n";
        for (int which = 0; which < 4; which++) {
            static const char *names[4] = {"mean_coverage", "mean_coverage_per_bin", "median_coverage",
                                           "median_over_mean_coverage"};
            j.key(2, names[which]);
            if (seqs.empty()) {
                j.s += "{},\n";
                continue;
            }
            j.s += "{\n";
            for (size_t k = 0; k < seqs.size(); k++) {
                const uint32_t r = seqs[k];
                j.key(3, ref_names[r]);
                if (which == 1) {
                    // coverage.rs:217-230: full bins divide by bin_size, the tail by L % bin_size
                    const u64 L = c->ref_len[r], B = c->cfg.bin_size;
                    const u64 nb = 1 + L / B + (L % B != 0);
                    const u64 *t = c->h_bin_totals.data() + c->bin_off[r];
                    j.s += "[\n";
                    for (u64 b = 0; b < nb; b++) {
                        const bool tail = (L % B != 0) && b + 1 == nb;
                        j.indent(4);
                        j.flt((double)t[b] / (tail ? (double)(L % B) : (double)B));
                        j.s += b + 1 == nb ? "\n" : ",\n";
                    }
                    j.indent(3);
                    j.s += "]";
                } else {
                    j.flt(which == 0 ? mean[r] : which == 2 ? median[r] : mom[r]);
                }
                j.s += k + 1 == seqs.size() ? "\n" : ",\n";
            }
            j.close(2, false);
        }
        j.key(2, "ignored");
        j.s += "{\n";
        j.u_field(3, "nonsensical_records", cnt[C_COV_NONSENSICAL], false);
        j.key(3, "pileup_too_large_positions");
        if (seqs.empty()) {
            j.s += "{}\n";
        } else {
            j.s += "{\n";
            for (size_t k = 0; k < seqs.size(); k++)
                j.u_field(4, ref_names[seqs[k]], c->h_cov_hist[(uint64_t)seqs[k] * (cap + 2) + cap + 1],
                          k + 1 == seqs.size());
            j.close(3, true);
        }
        j.close(2, false);
        j.key(2, "coverage_distribution");
        j.histogram(2, dist.data(), cap);
        j.s += ",\n";
        // coverage.rs:276-284 (f32)
        j.key(2, "genome_covered_by");
        j.s += "{\n";
        static const uint32_t check[6] = {10, 20, 30, 40, 50, 60};
        for (int k = 0; k < 6; k++) {
            u64 n = 0;
            for (uint32_t i = check[k]; i <= cap; i++) n += dist[i];
            const float v = ((float)n / (float)total_positions) * 100.0f;
            j.key(3, (std::to_string(check[k]) + "x").c_str());
            j.flt(v);
            j.s += k == 5 ? "\n" : ",\n";
        }
        j.close(2, true);
        j.close(1, false);
    } else {
        j.s += "null,\n";
    }

    j.key(1, "edits");
    if (facets & NGSQ_FACET_EDITS) {
        const u64 *e1 = cnt + c->st.off_edits1, *e2 = cnt + c->st.off_edits2;
        j.s += "{\n";
        j.key(2, "read_one_edits");
        j.histogram(2, e1, 512);
        j.s += ",\n";
        j.key(2, "read_two_edits");
        j.histogram(2, e2, 512);
        j.s += ",\n";
        j.key(2, "vaf_histogram");
        j.histogram(2, c->h_vaf.data(), 100);
        j.s += ",\n";
        j.key(2, "summary");
        j.s += "{\n";
        j.f_field(3, "mean_edits_read_one", hist_mean(e1, NGSQ_EDITS_BINS), false); // edits.rs:346-353
        j.f_field(3, "mean_edits_read_two", hist_mean(e2, NGSQ_EDITS_BINS), true);
        j.close(2, true);
        j.close(1, true);
    } else {
        j.s += "null\n";
    }
    j.s += "}";

    if (buf && cap) {
        const size_t n = j.s.size() < cap - 1 ? j.s.size() : cap - 1;
        memcpy(buf, j.s.data(), n);
        buf[n] = 0;
    }
    return (int64_t)j.s.size();
}
