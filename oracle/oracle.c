/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle.h for scope and pinning status).
 *
 * Record-at-a-time CPU restatement of the reference's `ngs qc` facets.
 * Structure follows the reference: one `process` per facet per record, pass 1
 * (record-based facets) then pass 2 (sequence-based facets), bounds-checked
 * Histogram::increment per base / per position.
 */
#include "oracle.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "../include/ngsq_shared.h"

/* ------------------------------------------------------------------ state */

typedef struct {
    /* coverage.rs:99-113 CoverageFacet */
    orc_histogram *per_position; /* [n_refs]; values == NULL => no entry in the map */
    /* coverage.rs:41-69 CoverageMetrics, keyed by sequence index instead of name */
    int *has_entry;              /* sequence was torn down with records */
    double *mean_coverage;
    double **mean_coverage_per_bin;
    uint64_t *n_bins;
    double *median_coverage;
    double *median_over_mean;
    uint64_t nonsensical_records;
    uint64_t *pileup_too_large;
    orc_histogram *per_seq_coverages; /* kept for the parity getters */
    orc_histogram coverage_distribution;
    float genome_covered_by[6];
} orc_coverage;

typedef struct {
    /* edits.rs:47-57 EditMetrics */
    orc_histogram read_one_edits, read_two_edits, vaf_histogram;
    double mean_edits_read_one, mean_edits_read_two;
    /* edits.rs:79-93: per sequence here (the reference re-creates them in setup) */
    orc_histogram *refs_per_position, *alts_per_position; /* [n_refs], lazily allocated */
} orc_edits;

/* rust_lapper::Interval<usize, FeatureNameStrand> as features.rs:314-318 builds it (the strand is
   never read back) */
struct orc_interval {
    uint32_t ref_id, name;
    uint64_t start, stop;
};

struct orc_ctx {
    ngsq_config cfg;
    uint32_t *ref_len;
    uint8_t *ref_is_primary;
    const uint8_t **ref_bases;
    uint32_t *ref_bases_len; /* NULL: ref_len bases of every sequence */
    /* general/metrics.rs */
    ngsq_general_metrics general;
    double duplication_pct, mapped_pct, mismatch_pct, mismatch_hq_pct;
    /* template_length.rs:44-53 */
    orc_histogram tlen_hist;
    uint64_t tlen_processed, tlen_ignored;
    double tlen_unknown_pct, tlen_out_of_range_pct;
    /* gc_content/metrics.rs:55-68 */
    orc_histogram gc_hist;
    uint64_t gc_count, at_count, other_count, gc_processed, gc_ignored_flags, gc_ignored_too_short;
    double gc_content_pct, gc_ignored_flags_pct, gc_ignored_too_short_pct;
    /* quality_scores.rs:15-19: scores[i] is the histogram of 1-based position i+1;
       values == NULL => key absent */
    orc_histogram *scores;
    orc_coverage cov;
    orc_edits edits;
    /* features.rs:83-106: the two interval stores (per sequence lookups are a filter on ref here)
       and the feature names, kept as the name ids of ngsq_features */
    struct orc_interval *utr_store, *gene_store;
    uint64_t n_utr, n_gene;
    uint32_t role_name[5];
    int have_features;
    ngsq_features_metrics features;
    double feat_ignored_flags_pct, feat_ignored_nonprimary_pct;
    ngsq_error_counts errors;
    int finalized;
    double cpu_seconds;
    char err[256];
};

static double now_seconds(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void set_err(orc_ctx *c, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(c->err, sizeof c->err, fmt, ap);
    va_end(ap);
}

const char *orc_last_error(const orc_ctx *c) { return c ? c->err : "null context"; }
double orc_elapsed_seconds(const orc_ctx *c) { return c->cpu_seconds; }

orc_ctx *orc_create(const ngsq_config *cfg) {
    if (!cfg || cfg->struct_size != sizeof(ngsq_config)) return NULL;
    orc_ctx *c = (orc_ctx *)calloc(1, sizeof *c);
    if (!c) return NULL;
    c->cfg = *cfg;
    if (!c->cfg.bin_size) c->cfg.bin_size = 50000; /* qc.rs:87 */
    if (!c->cfg.tlen_cap) c->cfg.tlen_cap = 1024;  /* qc.rs:62 */
    if (!c->cfg.cov_cap) c->cfg.cov_cap = 2048;    /* coverage.rs:76 */
    if (!c->cfg.max_read_len) c->cfg.max_read_len = 512;
    uint32_t n = cfg->n_refs;
    c->ref_len = (uint32_t *)calloc(n ? n : 1, sizeof(uint32_t));
    c->ref_is_primary = (uint8_t *)calloc(n ? n : 1, 1);
    c->ref_bases = (const uint8_t **)calloc(n ? n : 1, sizeof(uint8_t *));
    for (uint32_t r = 0; r < n; r++) {
        c->ref_len[r] = cfg->ref_len[r];
        c->ref_is_primary[r] = cfg->ref_is_primary ? cfg->ref_is_primary[r] : 1;
        c->ref_bases[r] = cfg->ref_bases ? cfg->ref_bases[r] : NULL;
    }
    if (cfg->ref_bases && cfg->ref_bases_len) {
        c->ref_bases_len = (uint32_t *)calloc(n ? n : 1, sizeof(uint32_t));
        for (uint32_t r = 0; r < n; r++) c->ref_bases_len[r] = cfg->ref_bases_len[r];
    }
    orc_hist_init(&c->tlen_hist, c->cfg.tlen_cap); /* template_length.rs:58-67 */
    orc_hist_init(&c->gc_hist, 100);               /* gc_content.rs:129-138 */
    c->scores = (orc_histogram *)calloc(c->cfg.max_read_len, sizeof(orc_histogram));
    orc_coverage *cv = &c->cov;
    cv->per_position = (orc_histogram *)calloc(n ? n : 1, sizeof(orc_histogram));
    cv->has_entry = (int *)calloc(n ? n : 1, sizeof(int));
    cv->mean_coverage = (double *)calloc(n ? n : 1, sizeof(double));
    cv->mean_coverage_per_bin = (double **)calloc(n ? n : 1, sizeof(double *));
    cv->n_bins = (uint64_t *)calloc(n ? n : 1, sizeof(uint64_t));
    cv->median_coverage = (double *)calloc(n ? n : 1, sizeof(double));
    cv->median_over_mean = (double *)calloc(n ? n : 1, sizeof(double));
    cv->pileup_too_large = (uint64_t *)calloc(n ? n : 1, sizeof(uint64_t));
    cv->per_seq_coverages = (orc_histogram *)calloc(n ? n : 1, sizeof(orc_histogram));
    orc_hist_init(&cv->coverage_distribution, c->cfg.cov_cap); /* coverage.rs:78-93 */
    orc_hist_init_default(&c->edits.read_one_edits);           /* edits.rs:59-68 */
    orc_hist_init_default(&c->edits.read_two_edits);
    orc_hist_init(&c->edits.vaf_histogram, 100);
    c->edits.refs_per_position = (orc_histogram *)calloc(n ? n : 1, sizeof(orc_histogram));
    c->edits.alts_per_position = (orc_histogram *)calloc(n ? n : 1, sizeof(orc_histogram));
    return c;
}

/* features.rs:300-343: every GFF record on a primary sequence whose type is a UTR/CDS name goes to the
   exonic-translation store, else one whose type is the exon or gene name to the gene-region store.
   Lapper::new sorts by start (rust-lapper 1.1.0 sorts the intervals): a stable sort by (start, stop). */
static int interval_cmp(const void *a, const void *b) {
    const struct orc_interval *x = (const struct orc_interval *)a, *y = (const struct orc_interval *)b;
    if (x->start != y->start) return x->start < y->start ? -1 : 1;
    if (x->stop != y->stop) return x->stop < y->stop ? -1 : 1;
    return 0;
}

int orc_set_features(orc_ctx *c, const ngsq_features *f) {
    if (!c || !f || f->struct_size != sizeof(ngsq_features)) return NGSQ_ERR_INVALID_ARGUMENT;
    free(c->utr_store);
    free(c->gene_store);
    c->utr_store = (struct orc_interval *)calloc(f->n ? f->n : 1, sizeof(struct orc_interval));
    c->gene_store = (struct orc_interval *)calloc(f->n ? f->n : 1, sizeof(struct orc_interval));
    c->n_utr = c->n_gene = 0;
    for (int r = 0; r < 5; r++) c->role_name[r] = f->role_name[r];
    for (uint64_t i = 0; i < f->n; i++) {
        if (f->ref_id[i] >= c->cfg.n_refs) return NGSQ_ERR_INVALID_ARGUMENT;
        if (!c->ref_is_primary[f->ref_id[i]]) continue; /* :300: only primary-assembly sequences get a store */
        struct orc_interval iv = {f->ref_id[i], f->name[i], f->start[i], f->stop[i]};
        if (iv.name == f->role_name[NGSQ_ROLE_FIVE_PRIME_UTR] || iv.name == f->role_name[NGSQ_ROLE_THREE_PRIME_UTR] ||
            iv.name == f->role_name[NGSQ_ROLE_CODING_SEQUENCE]) /* :322-326 */
            c->utr_store[c->n_utr++] = iv;
        else if (iv.name == f->role_name[NGSQ_ROLE_EXON] || iv.name == f->role_name[NGSQ_ROLE_GENE]) /* :327-331 */
            c->gene_store[c->n_gene++] = iv;
    }
    qsort(c->utr_store, c->n_utr, sizeof(struct orc_interval), interval_cmp);
    qsort(c->gene_store, c->n_gene, sizeof(struct orc_interval), interval_cmp);
    c->have_features = 1;
    return NGSQ_OK;
}

void orc_destroy(orc_ctx *c) {
    if (!c) return;
    uint32_t n = c->cfg.n_refs;
    orc_hist_free(&c->tlen_hist);
    orc_hist_free(&c->gc_hist);
    for (uint32_t i = 0; i < c->cfg.max_read_len; i++) orc_hist_free(&c->scores[i]);
    free(c->scores);
    for (uint32_t r = 0; r < n; r++) {
        orc_hist_free(&c->cov.per_position[r]);
        orc_hist_free(&c->cov.per_seq_coverages[r]);
        free(c->cov.mean_coverage_per_bin[r]);
        orc_hist_free(&c->edits.refs_per_position[r]);
        orc_hist_free(&c->edits.alts_per_position[r]);
    }
    free(c->cov.per_position);
    free(c->cov.per_seq_coverages);
    free(c->cov.has_entry);
    free(c->cov.mean_coverage);
    free(c->cov.mean_coverage_per_bin);
    free(c->cov.n_bins);
    free(c->cov.median_coverage);
    free(c->cov.median_over_mean);
    free(c->cov.pileup_too_large);
    orc_hist_free(&c->cov.coverage_distribution);
    orc_hist_free(&c->edits.read_one_edits);
    orc_hist_free(&c->edits.read_two_edits);
    orc_hist_free(&c->edits.vaf_histogram);
    free(c->edits.refs_per_position);
    free(c->edits.alts_per_position);
    free(c->utr_store);
    free(c->gene_store);
    free(c->ref_len);
    free(c->ref_is_primary);
    free(c->ref_bases);
    free(c->ref_bases_len);
    free(c);
}

/* ------------------------------------------------------- record accessors */
/* noodles sam::alignment::Record accessors restated from the SAM/BAM spec. */

static int flag_segmented(uint16_t f) { return (f & 0x1) != 0; }
static int flag_properly_aligned(uint16_t f) { return (f & 0x2) != 0; }
static int flag_unmapped(uint16_t f) { return (f & 0x4) != 0; }
static int flag_mate_unmapped(uint16_t f) { return (f & 0x8) != 0; }
static int flag_first_segment(uint16_t f) { return (f & 0x40) != 0; }
static int flag_last_segment(uint16_t f) { return (f & 0x80) != 0; }
static int flag_secondary(uint16_t f) { return (f & 0x100) != 0; }
static int flag_duplicate(uint16_t f) { return (f & 0x400) != 0; }
static int flag_supplementary(uint16_t f) { return (f & 0x800) != 0; }

/* utils/cigar.rs:6-11: M D N = X */
static int consumes_reference(uint32_t op) { return op == 0 || op == 2 || op == 3 || op == 7 || op == 8; }
/* utils/cigar.rs:14-23: M I S = X */
static int consumes_sequence(uint32_t op) { return op == 0 || op == 1 || op == 4 || op == 7 || op == 8; }

/* noodles Cigar::alignment_span: sum of lengths of reference-consuming ops */
static uint64_t alignment_span(const orc_record *r) {
    uint64_t span = 0;
    for (uint32_t k = 0; k < r->n_cigar; k++) {
        uint32_t op = r->cigar[k] & 0xF, len = r->cigar[k] >> 4;
        if (consumes_reference(op)) span += len;
    }
    return span;
}

/* base i of a packed 4-bit sequence (high nibble first) */
static uint32_t base_at(const uint8_t *seq, uint32_t i) {
    uint8_t b = seq[i >> 1];
    return (i & 1) ? (uint32_t)(b & 0xF) : (uint32_t)(b >> 4);
}

/* ----------------------------------------------- General (general.rs:31-124) */

static void general_process(orc_ctx *c, const orc_record *r) {
    ngsq_general_metrics *m = &c->general;
    /* (1) :33 */
    m->total += 1;
    /* (2) :36-100 */
    uint16_t f = r->flag;
    if (flag_unmapped(f)) m->unmapped += 1;
    if (flag_duplicate(f)) m->duplicate += 1;
    if (flag_secondary(f)) {
        m->secondary += 1;
    } else if (flag_supplementary(f)) {
        m->supplementary += 1;
    } else {
        m->primary += 1;
        if (!flag_unmapped(f)) m->primary_mapped += 1;
        if (flag_duplicate(f)) m->primary_duplicate += 1;
        if (flag_segmented(f)) {
            m->paired += 1;
            if (flag_first_segment(f)) m->read_1 += 1;
            if (flag_last_segment(f)) m->read_2 += 1;
            if (!flag_unmapped(f)) {
                if (flag_properly_aligned(f)) m->proper_pair += 1;
                if (flag_mate_unmapped(f)) {
                    m->singleton += 1;
                } else {
                    m->mate_mapped += 1;
                    /* :81-83 reference_sequence_id().unwrap(): None (-1) panics */
                    if (r->ref_id < 0 || r->mate_ref_id < 0) { /* [N1] */
                        c->errors.missing_reference_id += 1;
                    } else if (r->ref_id != r->mate_ref_id) {
                        m->mate_reference_sequence_id_mismatch += 1;
                        /* :88-91 [N2] missing MAPQ (255) maps to MISSING = 255 */
                        uint8_t mapq = r->mapq;
                        if (mapq >= 5) m->mate_reference_sequence_id_mismatch_hq += 1;
                    }
                }
            }
        }
    }
    /* (3) :103-121 every record, whatever its designation */
    int read_one = flag_first_segment(r->flag);
    for (uint32_t k = 0; k < r->n_cigar; k++) {
        uint32_t op = r->cigar[k] & 0xF;
        if (op >= NGSQ_N_CIGAR_KINDS) {
            c->errors.bad_cigar_op += 1;
            continue;
        }
        if (read_one)
            m->read_one_cigar_ops[op] += 1;
        else
            m->read_two_cigar_ops[op] += 1;
    }
}

/* general.rs:126-153 */
static void general_summarize(orc_ctx *c) {
    const ngsq_general_metrics *m = &c->general;
    c->duplication_pct = (double)m->duplicate / (double)m->total * 100.0;
    c->mapped_pct = (1.0 - (double)m->unmapped / (double)m->total) * 100.0;
    c->mismatch_pct = (double)m->mate_reference_sequence_id_mismatch / (double)m->total * 100.0;
    c->mismatch_hq_pct = (double)m->mate_reference_sequence_id_mismatch_hq / (double)m->total * 100.0;
}

/* -------------------------------- Template Length (template_length.rs:79-100) */

static void tlen_process(orc_ctx *c, const orc_record *r) {
    /* :80  record.template_length() as usize : i32 sign-extends, negatives wrap high */
    uint64_t template_len = (uint64_t)(int64_t)r->tlen;
    if (orc_hist_increment(&c->tlen_hist, template_len) == ORC_OK)
        c->tlen_processed += 1;
    else
        c->tlen_ignored += 1;
}

static void tlen_summarize(orc_ctx *c) {
    /* :90-97 */
    c->tlen_unknown_pct = ((double)orc_hist_get(&c->tlen_hist, 0) /
                           ((double)c->tlen_processed + (double)c->tlen_ignored)) * 100.0;
    c->tlen_out_of_range_pct = ((double)c->tlen_ignored /
                                ((double)c->tlen_processed + (double)c->tlen_ignored)) * 100.0;
}

/* ------------------------------------------ GC Content (gc_content.rs:38-122) */

static void gc_process(orc_ctx *c, const orc_record *r) {
    /* (1) :41-45 */
    if (flag_duplicate(r->flag) || flag_secondary(r->flag)) {
        c->gc_ignored_flags += 1;
        return;
    }
    /* (3) :59-62 */
    uint32_t sequence_length = r->l_seq;
    if (sequence_length < NGSQ_GC_WINDOW) {
        c->gc_ignored_too_short += 1;
        return;
    }
    /* (4) :68-74 -- ThreadRng replaced by the pinned offset function */
    uint64_t gc_this_read = 0;
    uint32_t offset = ngsq_gc_offset_fn(c->cfg.gc_seed, r->index, sequence_length);
    /* (5) :77-87 */
    for (uint32_t i = 0; i < NGSQ_GC_WINDOW; i++) {
        uint32_t nucleobase = base_at(r->seq, offset + i);
        if (nucleobase == 2 || nucleobase == 4) { /* C | G */
            gc_this_read += 1;
            c->gc_count += 1;
        } else if (nucleobase == 1 || nucleobase == 8) { /* A | T */
            c->at_count += 1;
        } else {
            c->other_count += 1;
        }
    }
    /* (6) :91-97 */
    uint64_t pct = (uint64_t)round(((double)gc_this_read / (double)NGSQ_GC_WINDOW) * 100.0);
    orc_hist_increment(&c->gc_hist, pct); /* .unwrap(): always in range */
    c->gc_processed += 1;
}

static void gc_summarize(orc_ctx *c) {
    /* :103-119 integer sums cast to f64 once */
    c->gc_content_pct =
        ((double)c->gc_count / (double)(c->gc_count + c->at_count + c->other_count)) * 100.0;
    double denom = (double)(c->gc_ignored_flags + c->gc_ignored_too_short + c->gc_processed);
    c->gc_ignored_flags_pct = ((double)c->gc_ignored_flags / denom) * 100.0;
    c->gc_ignored_too_short_pct = ((double)c->gc_ignored_too_short / denom) * 100.0;
}

/* ------------------------------------ Quality Score (quality_scores.rs:37-49) */

static void quality_process(orc_ctx *c, const orc_record *r) {
    for (uint32_t i = 0; i < r->n_qual; i++) {
        /* fixed-stride rows (ngsq.h): 0xFF = no score at this cycle (BAM's own
           encoding of absent qualities; noodles yields no score for them) */
        if (r->qual_fixed_row && r->qual[i] == 0xFF) continue;
        /* quality_scores.rs:18: a map entry per position -- no length is too long.  (max_read_len is where this array
           starts; the HIP path's table grows the same way, include/ngsq.h ngsq_batch.max_l_seq.) */
        if (i >= c->cfg.max_read_len) {
            uint32_t want = c->cfg.max_read_len;
            while (want <= i) want += want / 2 + 64;
            orc_histogram *bigger = (orc_histogram *)realloc(c->scores, (size_t)want * sizeof(orc_histogram));
            if (!bigger) {
                c->errors.read_too_long += 1;
                break;
            }
            memset(bigger + c->cfg.max_read_len, 0, (size_t)(want - c->cfg.max_read_len) * sizeof(orc_histogram));
            c->scores = bigger;
            c->cfg.max_read_len = want;
        }
        /* :39-42 entry(i + 1).or_insert_with(zero_based_with_capacity(93)) */
        orc_histogram *h = &c->scores[i];
        if (!h->values) orc_hist_init(h, NGSQ_MAX_SCORE);
        /* :44-45 increment(score).unwrap(); noodles rejects > 93 at decode */
        if (orc_hist_increment(h, r->qual[i]) != ORC_OK) c->errors.bad_quality_score += 1;
    }
}

/* ------------------------------------------ pass 2: query + Coverage + Edits */

/* [N3] [N4] [N5] (oracle.h) noodles bam::Reader::query over Region(name, 1..=L) (command.rs:369-373):
 * a record is yielded iff reference_sequence_id == id, alignment_start and
 * alignment_end are Some, and [start, end] intersects [1, L].
 * alignment_end = start + span - 1, None when that is 0. */
static int query_yields(const orc_ctx *c, const orc_record *r, uint64_t *start, uint64_t *end) {
    if (r->ref_id < 0 || (uint32_t)r->ref_id >= c->cfg.n_refs) return 0;
    if (r->pos < 0) return 0;
    uint64_t s = (uint64_t)r->pos + 1;
    uint64_t e = s + alignment_span(r) - 1;
    if (e == 0) return 0;
    if (s > (uint64_t)c->ref_len[r->ref_id]) return 0;
    *start = s;
    *end = e;
    return 1;
}

/* coverage.rs:148-180 */
static void coverage_process(orc_ctx *c, const orc_record *r, uint64_t start, uint64_t end) {
    orc_coverage *cv = &c->cov;
    orc_histogram *h = &cv->per_position[r->ref_id];
    if (!h->values) orc_hist_init(h, c->ref_len[r->ref_id]); /* :154-157 */
    for (uint64_t i = start; i <= end; i++) {                /* :162 */
        if (orc_hist_increment(h, i) != ORC_OK) cv->nonsensical_records += 1; /* :163-176 */
    }
}

/* coverage.rs:182-262 */
static int coverage_teardown(orc_ctx *c, uint32_t ref) {
    orc_coverage *cv = &c->cov;
    orc_histogram *positions = &cv->per_position[ref];
    if (!positions->values) return ORC_OK; /* :187-193 */

    orc_histogram coverages;
    orc_hist_init(&coverages, c->cfg.cov_cap); /* :195-196 */
    uint64_t ignored = 0;
    uint64_t total_coverage_for_bin = 0;
    const uint64_t bin_size = c->cfg.bin_size;
    uint64_t cap_bins = positions->range_stop / bin_size + 3;
    double *bins = (double *)malloc(cap_bins * sizeof(double));
    uint64_t nb = 0;

    for (uint64_t i = positions->range_start; i <= positions->range_stop; i++) { /* :206 */
        uint64_t coverage_at_position = orc_hist_get(positions, i);
        if (orc_hist_increment(&coverages, coverage_at_position) != ORC_OK) ignored += 1; /* :211-213 */
        total_coverage_for_bin += coverage_at_position; /* :216 */
        if (i % bin_size == 0) {                         /* :217-221 */
            bins[nb++] = (double)total_coverage_for_bin / (double)bin_size;
            total_coverage_for_bin = 0;
        }
    }
    uint64_t modulo = positions->range_stop % bin_size; /* :226-230 */
    if (modulo != 0) bins[nb++] = (double)total_coverage_for_bin / (double)modulo;

    double mean = orc_hist_mean(&coverages); /* :232 */
    int is_some = 0;
    double median = 0.0;
    int rc = orc_hist_median(&coverages, &is_some, &median); /* :233 .unwrap() */
    if (rc != ORC_OK || !is_some) {
        free(bins);
        orc_hist_free(&coverages);
        set_err(c, "coverage median of sequence %u would panic in the reference", ref);
        return ORC_PANIC;
    }
    double median_over_mean = median / mean; /* :234 */

    orc_hist_free(positions); /* :237 */

    for (uint64_t i = 0; i <= coverages.range_stop; i++) /* :241-246 */
        orc_hist_increment_by(&cv->coverage_distribution, i, orc_hist_get(&coverages, i));

    cv->has_entry[ref] = 1; /* :249-259 */
    cv->mean_coverage[ref] = mean;
    cv->median_coverage[ref] = median;
    cv->median_over_mean[ref] = median_over_mean;
    cv->pileup_too_large[ref] = ignored;
    cv->mean_coverage_per_bin[ref] = bins;
    cv->n_bins[ref] = nb;
    cv->per_seq_coverages[ref] = coverages;
    return ORC_OK;
}

/* coverage.rs:264-287 */
static void coverage_aggregate(orc_ctx *c) {
    orc_coverage *cv = &c->cov;
    uint64_t total_positions = orc_hist_sum(&cv->coverage_distribution); /* :266 */
    for (uint32_t r = 0; r < c->cfg.n_refs; r++)
        if (cv->has_entry[r]) total_positions += cv->pileup_too_large[r]; /* :270-272 */
    static const uint64_t COVERAGES_TO_CHECK[6] = {10, 20, 30, 40, 50, 60}; /* :276 */
    for (int k = 0; k < 6; k++) {
        uint64_t bin = COVERAGES_TO_CHECK[k];
        uint64_t n = bin <= cv->coverage_distribution.range_stop
                         ? orc_hist_count_from_top_until(&cv->coverage_distribution, bin)
                         : 0;
        /* :282 f32 arithmetic */
        cv->genome_covered_by[k] = ((float)n / (float)total_positions) * 100.0f;
    }
}

const char *orc_stepthrough_error_message(int code) {
    switch (code) {
    case 1: /* alignment.rs:70-73 */
        return "malformed record: record specifies that we should be able to consume a reference "
               "base, but no such base was found";
    case 2: /* alignment.rs:84-87 */
        return "malformed record: record specifies that we should be able to consume a record "
               "base, but no such base was found";
    case 3: /* alignment.rs:101 */
        return "reference sequence was not fully consumed";
    case 4: /* alignment.rs:103 */
        return "record sequence was not fully consumed";
    default:
        return "";
    }
}

/* alignment.rs:48-107 stepthrough with a visitor; `visit` may be NULL.
 * The CIGAR is walked op by op instead of being flattened into a Vec<Kind>
 * (alignment.rs:13-25): the visit order is identical. */
typedef void (*orc_visit_fn)(void *user, uint32_t kind, int has_ref, uint32_t ref_base,
                             size_t reference_ptr, int has_rec, uint32_t rec_base);

static int stepthrough(const uint8_t *reference, size_t n_reference, const uint8_t *record_packed,
                       const uint8_t *record_unpacked, size_t n_record, const uint32_t *cigar,
                       size_t n_cigar, orc_visit_fn visit, void *user) {
    size_t record_ptr = 0, reference_ptr = 0;
    for (size_t k = 0; k < n_cigar; k++) {
        uint32_t kind = cigar[k] & 0xF, len = cigar[k] >> 4;
        int c_ref = consumes_reference(kind), c_seq = consumes_sequence(kind);
        for (uint32_t j = 0; j < len; j++) {
            uint32_t ref_base = 0, rec_base = 0;
            if (c_ref) { /* :58-75 */
                if (reference_ptr >= n_reference) return 1;
                ref_base = reference[reference_ptr];
            }
            if (c_seq) { /* :77-91 */
                if (record_ptr >= n_record) return 2;
                rec_base = record_packed ? base_at(record_packed, (uint32_t)record_ptr)
                                         : record_unpacked[record_ptr];
            }
            if (visit) visit(user, kind, c_ref, ref_base, reference_ptr, c_seq, rec_base); /* :93 */
            if (c_ref) reference_ptr += 1;
            if (c_seq) record_ptr += 1;
        }
    }
    if (n_reference != reference_ptr) return 3; /* :100-101 */
    if (n_record != record_ptr) return 4;       /* :102-103 */
    return 0;
}

static void count_edit(void *user, uint32_t kind, int has_ref, uint32_t ref_base, size_t rp,
                       int has_rec, uint32_t rec_base) {
    (void)rp;
    (void)has_ref;
    (void)has_rec;
    /* alignment.rs:116-118: Kind::Match && reference != record */
    if (kind == 0 && ref_base != rec_base) *(uint64_t *)user += 1;
}

/* alignment.rs:113-125 */
int orc_stepthrough_edits(const uint8_t *reference, size_t n_reference, const uint8_t *record,
                          size_t n_record, const uint32_t *cigar, size_t n_cigar,
                          uint64_t *edits) {
    *edits = 0;
    return stepthrough(reference, n_reference, NULL, record, n_record, cigar, n_cigar, count_edit,
                       edits);
}

typedef struct {
    orc_ctx *c;
    uint32_t ref;
    uint64_t reference_start;
    uint64_t edits;
    int beyond; /* an `M` base at a position the per-position histograms do not have: increment(..).unwrap() panics (:283-291) */
} edits_visit_state;

static void edits_visit(void *user, uint32_t kind, int has_ref, uint32_t ref_base, size_t rp,
                        int has_rec, uint32_t rec_base) {
    (void)has_ref;
    (void)has_rec;
    edits_visit_state *s = (edits_visit_state *)user;
    if (kind == 0) { /* edits.rs:277 Kind::Match */
        uint64_t reference_position = s->reference_start + rp; /* :278-280 */
        if (ref_base != rec_base) {                             /* :282-287 */
            s->edits += 1;
            if (orc_hist_increment(&s->c->edits.alts_per_position[s->ref], reference_position) != ORC_OK) s->beyond = 1;
        } else { /* :288-291 */
            if (orc_hist_increment(&s->c->edits.refs_per_position[s->ref], reference_position) != ORC_OK) s->beyond = 1;
        }
    }
}

/* [N9] noodles-sam 0.25 record::sequence::Base: `impl TryFrom<u8>` goes through `char::from(n)` to `impl TryFrom<char>`,
 * which matches `c.to_ascii_uppercase()` against the sixteen letters "=ACMGRSVTWYHKDBN" (the crate's own unit test of that
 * impl converts 'a' as well as 'A').  edits.rs:257-261 applies it to the FASTA's raw bytes -- noodles-fasta keeps them as
 * they are in the file, case included.  Returns the 4-bit BAM code, or -1 = TryFromCharError. */
int orc_fasta_base_code(uint8_t byte) {
    static const char letters[] = "=ACMGRSVTWYHKDBN";
    uint8_t c = byte;
    if (c >= 'a' && c <= 'z') c = (uint8_t)(c - 'a' + 'A');
    for (int k = 0; k < 16; k++)
        if ((uint8_t)letters[k] == c) return k;
    return -1;
}

/* edits.rs:217-303 */
static void edits_process(orc_ctx *c, const orc_record *r, uint64_t start) {
    /* (1) :227-229 */
    if (flag_unmapped(r->flag) || flag_duplicate(r->flag)) return;
    uint32_t ref = (uint32_t)r->ref_id;
    orc_edits *e = &c->edits;
    /* (3) :241-243 */
    uint64_t reference_start = start;
    uint64_t reference_end = reference_start + alignment_span(r);
    /* :245-251 + :257-261  current_sequence.get(start..end) -> [start-1, end-1) */
    const uint8_t *bases = c->ref_bases[ref];
    /* the slice is taken from the FASTA's sequence, whose length need not be @SQ LN (ngsq_config.ref_bases_len): `get` is None
     * -- unwrap panics -- when the read runs past ITS end.  A read past LN inside a longer FASTA sequence is another matter:
     * what panics then is refs/alts_per_position.increment (:283-291, histograms of LN + 1 bins), and only an `M` base
     * increments -- a read whose bases beyond LN lie under D, N, = or X goes through (found in round 6 by the second reading of
     * the source, tests/literal_model.py: until then every read that ended beyond LN was counted as one the reference stops at). */
    uint64_t have = c->ref_bases_len ? c->ref_bases_len[ref] : c->ref_len[ref];
    if (!bases || reference_end - 1 > have) {
        c->errors.edits_bad_reference += 1;
        return;
    }
    /* :257-261 [N9] Base::try_from over every byte of the slice, collected into a Result and `?`: ONE byte it refuses (here:
     * a code above 15, what orc_fasta_base_code's -1 is stored as) fails the record -- whatever CIGAR operation lies over it,
     * and only records whose slice holds it */
    for (uint64_t i = reference_start - 1; i < reference_end - 1; i++)
        if (bases[i] > 15) {
            c->errors.edits_bad_reference += 1;
            return;
        }
    if (!e->refs_per_position[ref].values) { /* setup :211-213 */
        orc_hist_init(&e->refs_per_position[ref], c->ref_len[ref]);
        orc_hist_init(&e->alts_per_position[ref], c->ref_len[ref]);
    }
    edits_visit_state st = {c, ref, reference_start, 0, 0};
    int rc = stepthrough(bases + (reference_start - 1), (size_t)(reference_end - reference_start),
                         r->seq, NULL, r->l_seq, r->cigar, r->n_cigar, edits_visit, &st);
    if (st.beyond) { /* (the walk visits the bases in order and stops at its own first error: this one came before it) */
        c->errors.edits_bad_reference += 1;
        return;
    }
    if (rc == 2) {
        c->errors.edits_record_short += 1;
        return;
    }
    if (rc != 0) {
        c->errors.edits_not_consumed += 1;
        return;
    }
    /* :296-300 */
    orc_histogram *h = flag_first_segment(r->flag) ? &e->read_one_edits : &e->read_two_edits;
    if (orc_hist_increment(h, st.edits) != ORC_OK) c->errors.edits_too_many += 1;
}

/* edits.rs:305-344 */
static void edits_teardown(orc_ctx *c, uint32_t ref) {
    orc_edits *e = &c->edits;
    orc_histogram *refs = &e->refs_per_position[ref], *alts = &e->alts_per_position[ref];
    if (!refs->values) return; /* all-zero histograms contribute nothing */
    for (uint64_t i = refs->range_start; i <= refs->range_stop; i++) { /* :320 */
        uint64_t refs_at = orc_hist_get(refs, i), alts_at = orc_hist_get(alts, i);
        uint64_t total = refs_at + alts_at;
        if (total == 0) continue;                     /* :325-329 */
        float vaf = (float)alts_at / (float)total;    /* :331 */
        float scaled = vaf * 100.0f;                  /* :334 */
        orc_hist_increment(&e->vaf_histogram, (uint64_t)scaled);
    }
    orc_hist_free(refs);
    orc_hist_free(alts);
}

/* ------------------------------------------------------------ batch driver */

/* ------------------------------------------------------------ Genomic Features */

/* rust_lapper Interval::overlap (lib.rs, 1.1.0): self.start < stop && self.stop > start */
static int lapper_overlap(const struct orc_interval *iv, uint64_t start, uint64_t stop) {
    return iv->start < stop && iv->stop > start;
}

/* features.rs:115-242 */
static void features_process(orc_ctx *c, const orc_record *r) {
    ngsq_features_metrics *m = &c->features;
    if (flag_unmapped(r->flag)) { /* :127-130 */
        m->ignored_flags += 1;
        return;
    }
    if (r->ref_id < 0 || (uint32_t)r->ref_id >= c->cfg.n_refs) { /* :132-155 bail! */
        c->errors.features_missing_reference_id += 1;
        return;
    }
    if (!c->ref_is_primary[r->ref_id]) { /* :157-165 */
        m->ignored_nonprimary_chromosome += 1;
        return;
    }
    if (r->pos < 0) { /* :171-174 bail! */
        c->errors.features_missing_position += 1;
        return;
    }
    const uint64_t start = (uint64_t)r->pos + 1; /* 1-based alignment_start */
    const uint64_t end = start + alignment_span(r); /* :178 */
    /* :186-214: walk find(start, end + 1) of the exonic-translation store in store order */
    int counted5 = 0, counted3 = 0, counted_cds = 0;
    for (uint64_t k = 0; k < c->n_utr; k++) {
        const struct orc_interval *iv = &c->utr_store[k];
        if (iv->ref_id != (uint32_t)r->ref_id || !lapper_overlap(iv, start, end + 1)) continue;
        if (!counted5 && iv->name == c->role_name[NGSQ_ROLE_FIVE_PRIME_UTR]) {
            counted5 = 1;
            m->utr_five_prime_count += 1;
        } else if (!counted3 && iv->name == c->role_name[NGSQ_ROLE_THREE_PRIME_UTR]) {
            counted3 = 1;
            m->utr_three_prime_count += 1;
        } else if (!counted_cds && iv->name == c->role_name[NGSQ_ROLE_CODING_SEQUENCE]) {
            counted_cds = 1;
            m->coding_sequence_count += 1;
        }
    }
    /* :216-238 */
    int has_gene = 0, has_exon = 0;
    for (uint64_t k = 0; k < c->n_gene; k++) {
        const struct orc_interval *iv = &c->gene_store[k];
        if (iv->ref_id != (uint32_t)r->ref_id || !lapper_overlap(iv, start, end + 1)) continue;
        if (iv->name == c->role_name[NGSQ_ROLE_GENE]) has_gene = 1;
        else if (iv->name == c->role_name[NGSQ_ROLE_EXON]) has_exon = 1;
        if (has_gene && has_exon) break;
    }
    if (has_gene) {
        if (has_exon) m->exonic_count += 1;
        else m->intronic_count += 1;
    } else {
        m->intergenic_count += 1;
    }
    m->processed += 1; /* :240 */
}

/* features.rs:244-262 */
static void features_summarize(orc_ctx *c) {
    const ngsq_features_metrics *m = &c->features;
    const double denom = (double)(m->ignored_flags + m->ignored_nonprimary_chromosome + m->processed);
    c->feat_ignored_flags_pct = ((double)m->ignored_flags / denom) * 100.0;
    c->feat_ignored_nonprimary_pct = ((double)m->ignored_nonprimary_chromosome / denom) * 100.0;
}

static int fetch_record(const ngsq_batch *b, uint64_t i, orc_record *r) {
    r->flag = b->flag[i];
    r->mapq = b->mapq ? b->mapq[i] : 255;
    r->ref_id = b->ref_id ? b->ref_id[i] : -1;
    r->pos = b->pos ? b->pos[i] : -1;
    r->mate_ref_id = b->mate_ref_id ? b->mate_ref_id[i] : -1;
    r->tlen = b->tlen ? b->tlen[i] : 0;
    r->l_seq = b->l_seq ? b->l_seq[i] : 0;
    r->qual_fixed_row = 0;
    r->seq = b->seq ? b->seq + (b->seq_off ? b->seq_off[i] : i * (uint64_t)b->seq_stride) : NULL;
    if (b->qual) {
        if (b->qual_off) {
            r->qual = b->qual + b->qual_off[i];
            r->n_qual = (uint32_t)(b->qual_off[i + 1] - b->qual_off[i]);
        } else {
            r->qual = b->qual + i * (uint64_t)b->qual_stride;
            r->n_qual = b->qual_stride;
            r->qual_fixed_row = 1;
        }
    } else {
        r->qual = NULL;
        r->n_qual = 0;
    }
    r->n_cigar = b->n_cigar ? b->n_cigar[i] : 0;
    if (b->cigar_off) r->n_cigar = (uint32_t)(b->cigar_off[i + 1] - b->cigar_off[i]); /* [N10]: the 16-bit column saturates, the offsets hold the count */
    r->cigar = b->cigar ? b->cigar + (b->cigar_off ? b->cigar_off[i] : i * (uint64_t)b->cigar_stride)
                        : NULL;
    r->index = b->record_id ? b->record_id[i] : b->first_record_index + i; /* include/ngsq.h: the record's identity */
    return 0;
}

int orc_process_batch(orc_ctx *c, const ngsq_batch *b, uint32_t pass_mask) {
    if (!c || !b || b->struct_size != sizeof(ngsq_batch)) return NGSQ_ERR_INVALID_ARGUMENT;
    if (c->finalized) return NGSQ_ERR_STATE;
    if (b->location != NGSQ_MEM_HOST) return NGSQ_ERR_INVALID_ARGUMENT;
    double t0 = now_seconds();
    uint32_t facets = c->cfg.facets;
    orc_record r;
    /* pass 1: command.rs:305-316 */
    if ((pass_mask & NGSQ_PASS_RECORD) && (facets & NGSQ_FACET_FEATURES) && !c->have_features) return NGSQ_ERR_STATE;
    if ((pass_mask & NGSQ_PASS_RECORD) && (facets & NGSQ_FACETS_RECORD_BASED)) {
        for (uint64_t i = 0; i < b->n_records; i++) {
            fetch_record(b, i, &r);
            /* facet order: qc.rs:60-65 */
            if (facets & NGSQ_FACET_GENERAL) general_process(c, &r);
            if (facets & NGSQ_FACET_TEMPLATE_LENGTH) tlen_process(c, &r);
            if (facets & NGSQ_FACET_GC_CONTENT) gc_process(c, &r);
            if (facets & NGSQ_FACET_QUALITY_SCORE) quality_process(c, &r);
            if (facets & NGSQ_FACET_FEATURES) features_process(c, &r); /* pushed last: qc.rs:68-79 */
        }
    }
    /* pass 2: command.rs:356-397.  The reference visits sequences one at a
       time through the BAI; every per-record contribution is a sum, so
       visiting the batch in file order gives identical state. */
    if ((pass_mask & NGSQ_PASS_SEQUENCE) && (facets & NGSQ_FACETS_SEQUENCE_BASED)) {
        for (uint64_t i = 0; i < b->n_records; i++) {
            fetch_record(b, i, &r);
            uint64_t start, end;
            if (!query_yields(c, &r, &start, &end)) continue;
            if ((facets & NGSQ_FACET_COVERAGE) && c->ref_is_primary[r.ref_id]) /* :378-379 */
                coverage_process(c, &r, start, end);
            if (facets & NGSQ_FACET_EDITS) edits_process(c, &r, start);
        }
    }
    c->cpu_seconds += now_seconds() - t0;
    return NGSQ_OK;
}

static int any_error(const ngsq_error_counts *e) {
    const uint64_t *p = (const uint64_t *)e;
    for (size_t i = 0; i < sizeof(*e) / sizeof(uint64_t); i++)
        if (p[i]) return 1;
    return 0;
}

int orc_finalize(orc_ctx *c) {
    if (!c) return NGSQ_ERR_INVALID_ARGUMENT;
    if (c->finalized) return NGSQ_ERR_STATE;
    double t0 = now_seconds();
    uint32_t facets = c->cfg.facets;
    /* command.rs:328-330 */
    if (facets & NGSQ_FACET_GENERAL) general_summarize(c);
    if (facets & NGSQ_FACET_TEMPLATE_LENGTH) tlen_summarize(c);
    if (facets & NGSQ_FACET_GC_CONTENT) gc_summarize(c);
    if (facets & NGSQ_FACET_FEATURES) features_summarize(c);
    /* command.rs:392-396, per sequence in header order */
    int rc = NGSQ_OK;
    for (uint32_t r = 0; r < c->cfg.n_refs; r++) {
        if ((facets & NGSQ_FACET_COVERAGE) && c->ref_is_primary[r])
            if (coverage_teardown(c, r) != ORC_OK) rc = NGSQ_ERR_MALFORMED_RECORD;
        if (facets & NGSQ_FACET_EDITS) edits_teardown(c, r);
    }
    /* command.rs:406-414 */
    if (facets & NGSQ_FACET_COVERAGE) coverage_aggregate(c);
    if (facets & NGSQ_FACET_EDITS) { /* edits.rs:346-353 */
        c->edits.mean_edits_read_one = orc_hist_mean(&c->edits.read_one_edits);
        c->edits.mean_edits_read_two = orc_hist_mean(&c->edits.read_two_edits);
    }
    c->finalized = 1;
    c->cpu_seconds += now_seconds() - t0;
    if (any_error(&c->errors)) {
        /* a read longer than the table of this build is an implementation limit of the HIP path (mirrored here so that
           the two report alike), not a condition of the reference: quality_scores.rs:18 keeps a map per position */
        ngsq_error_counts e = c->errors;
        const uint64_t too_long = e.read_too_long;
        e.read_too_long = 0;
        if (too_long && !any_error(&e)) {
            set_err(c, "implementation limit: read(s) longer than max_read_len");
            return NGSQ_ERR_LIMIT;
        }
        set_err(c, "malformed record(s): the reference would abort this run");
        return NGSQ_ERR_MALFORMED_RECORD;
    }
    return rc;
}

/* ---------------------------------------------------------------- getters */

int orc_get_error_counts(const orc_ctx *c, ngsq_error_counts *out) {
    *out = c->errors;
    return NGSQ_OK;
}

int orc_get_features(const orc_ctx *c, ngsq_features_metrics *out) {
    *out = c->features;
    return NGSQ_OK;
}

int orc_get_general(const orc_ctx *c, ngsq_general_metrics *out) {
    *out = c->general;
    return NGSQ_OK;
}

int orc_get_template_length(const orc_ctx *c, uint64_t *histogram, size_t n_bins,
                            uint64_t *processed, uint64_t *ignored) {
    if (n_bins < (size_t)c->cfg.tlen_cap + 1) return NGSQ_ERR_BUFFER_TOO_SMALL;
    memcpy(histogram, c->tlen_hist.values, ((size_t)c->cfg.tlen_cap + 1) * sizeof(uint64_t));
    *processed = c->tlen_processed;
    *ignored = c->tlen_ignored;
    return NGSQ_OK;
}

int orc_get_gc_content(const orc_ctx *c, ngsq_gc_metrics *out) {
    memcpy(out->histogram, c->gc_hist.values, sizeof out->histogram);
    out->total_gc_count = c->gc_count;
    out->total_at_count = c->at_count;
    out->total_other_count = c->other_count;
    out->processed = c->gc_processed;
    out->ignored_flags = c->gc_ignored_flags;
    out->ignored_too_short = c->gc_ignored_too_short;
    return NGSQ_OK;
}

/* rows [0, n_rows) of the table; cycles nobody reached are rows of zeros, cycles beyond n_rows must be empty */
int orc_get_quality_scores(const orc_ctx *c, uint64_t *scores, size_t n_rows) {
    for (uint32_t i = (uint32_t)n_rows; i < c->cfg.max_read_len; i++)
        if (c->scores[i].values) return NGSQ_ERR_BUFFER_TOO_SMALL;
    for (size_t i = 0; i < n_rows; i++) {
        uint64_t *row = scores + i * (NGSQ_MAX_SCORE + 1);
        if (i < c->cfg.max_read_len && c->scores[i].values)
            memcpy(row, c->scores[i].values, (NGSQ_MAX_SCORE + 1) * sizeof(uint64_t));
        else
            memset(row, 0, (NGSQ_MAX_SCORE + 1) * sizeof(uint64_t));
    }
    return NGSQ_OK;
}

uint32_t orc_quality_rows(const orc_ctx *c) { /* 1 + the last cycle any read reached */
    uint32_t n = 0;
    for (uint32_t i = 0; i < c->cfg.max_read_len; i++)
        if (c->scores[i].values) n = i + 1;
    return n;
}

uint64_t orc_coverage_n_bins(const orc_ctx *c, uint32_t ref) {
    uint64_t L = c->ref_len[ref], b = c->cfg.bin_size;
    return 1 + L / b + (L % b != 0);
}

int orc_get_coverage_sequence(const orc_ctx *c, uint32_t ref, int *seen, uint64_t *histogram,
                              size_t n_hist_bins, uint64_t *ignored, double *bin_means,
                              size_t n_bins) {
    if (ref >= c->cfg.n_refs) return NGSQ_ERR_INVALID_ARGUMENT;
    *seen = c->cov.has_entry[ref];
    *ignored = 0;
    if (!*seen) return NGSQ_OK;
    if (n_hist_bins < (size_t)c->cfg.cov_cap + 1 || n_bins < c->cov.n_bins[ref])
        return NGSQ_ERR_BUFFER_TOO_SMALL;
    memcpy(histogram, c->cov.per_seq_coverages[ref].values,
           ((size_t)c->cfg.cov_cap + 1) * sizeof(uint64_t));
    *ignored = c->cov.pileup_too_large[ref];
    memcpy(bin_means, c->cov.mean_coverage_per_bin[ref], c->cov.n_bins[ref] * sizeof(double));
    return NGSQ_OK;
}

int orc_get_coverage_nonsensical(const orc_ctx *c, uint64_t *n) {
    *n = c->cov.nonsensical_records;
    return NGSQ_OK;
}

int orc_get_edits(const orc_ctx *c, uint64_t *r1, uint64_t *r2, size_t n_edit_bins, uint64_t *vaf,
                  size_t n_vaf_bins) {
    if (n_edit_bins < NGSQ_EDITS_BINS || n_vaf_bins < NGSQ_VAF_BINS) return NGSQ_ERR_BUFFER_TOO_SMALL;
    memcpy(r1, c->edits.read_one_edits.values, NGSQ_EDITS_BINS * sizeof(uint64_t));
    memcpy(r2, c->edits.read_two_edits.values, NGSQ_EDITS_BINS * sizeof(uint64_t));
    memcpy(vaf, c->edits.vaf_histogram.values, NGSQ_VAF_BINS * sizeof(uint64_t));
    return NGSQ_OK;
}

/* ------------------------------------------------------------------- JSON */
/* serde_json::to_string_pretty layout (results.rs:55): two-space indent,
 * `"key": value`, one array element per line, NaN / inf -> null, shortest
 * round-trip float text laid out by ryu's rules.  HashMap fields are emitted
 * in sorted / header order (the reference's order is random per run). */

typedef struct {
    char *p;
    size_t len, cap;
} sbuf;

static void sb_put(sbuf *s, const char *str, size_t n) {
    if (s->len + n + 1 > s->cap) {
        size_t nc = s->cap ? s->cap * 2 : 1 << 16;
        while (nc < s->len + n + 1) nc *= 2;
        s->p = (char *)realloc(s->p, nc);
        s->cap = nc;
    }
    memcpy(s->p + s->len, str, n);
    s->len += n;
    s->p[s->len] = 0;
}
static void sb_puts(sbuf *s, const char *str) { sb_put(s, str, strlen(str)); }
static void sb_indent(sbuf *s, int depth) {
    for (int i = 0; i < depth; i++) sb_put(s, "  ", 2);
}
static void sb_u64(sbuf *s, uint64_t v) {
    char t[32];
    int n = snprintf(t, sizeof t, "%llu", (unsigned long long)v);
    sb_put(s, t, (size_t)n);
}

/* shortest digits that round-trip, then ryu's pretty layout (f64: kk<=16, -5<kk;
 * f32: kk<=13, -6<kk) */
static void sb_float(sbuf *s, double v, int is_f32) {
    if (isnan(v) || isinf(v)) {
        sb_puts(s, "null");
        return;
    }
    if (v == 0.0) {
        sb_puts(s, signbit(v) ? "-0.0" : "0.0");
        return;
    }
    char e[64];
    int prec;
    for (prec = 1; prec <= 17; prec++) {
        snprintf(e, sizeof e, "%.*e", prec - 1, v);
        if (is_f32 ? (strtof(e, NULL) == (float)v) : (strtod(e, NULL) == v)) break;
    }
    /* e = [-]d[.ddd]e[+-]xx */
    char digits[32];
    int nd = 0, neg = 0;
    const char *q = e;
    if (*q == '-') {
        neg = 1;
        q++;
    }
    for (; *q && *q != 'e'; q++)
        if (*q != '.') digits[nd++] = *q;
    int exp10 = atoi(q + 1);
    while (nd > 1 && digits[nd - 1] == '0') nd--; /* cannot happen for shortest, kept for safety */
    int k = exp10 - (nd - 1); /* value = digits * 10^k */
    int kk = nd + k;
    const int hi = is_f32 ? 13 : 16, lo = is_f32 ? -6 : -5;
    char out[64];
    int n = 0;
    if (neg) out[n++] = '-';
    if (0 <= k && kk <= hi) {
        memcpy(out + n, digits, (size_t)nd);
        n += nd;
        for (int i = 0; i < k; i++) out[n++] = '0';
        out[n++] = '.';
        out[n++] = '0';
    } else if (0 < kk && kk <= hi) {
        memcpy(out + n, digits, (size_t)kk);
        n += kk;
        out[n++] = '.';
        memcpy(out + n, digits + kk, (size_t)(nd - kk));
        n += nd - kk;
    } else if (lo < kk && kk <= 0) {
        out[n++] = '0';
        out[n++] = '.';
        for (int i = 0; i < -kk; i++) out[n++] = '0';
        memcpy(out + n, digits, (size_t)nd);
        n += nd;
    } else {
        out[n++] = digits[0];
        if (nd > 1) {
            out[n++] = '.';
            memcpy(out + n, digits + 1, (size_t)(nd - 1));
            n += nd - 1;
        }
        n += snprintf(out + n, sizeof out - (size_t)n, "e%d", kk - 1);
    }
    sb_put(s, out, (size_t)n);
}

static void js_key(sbuf *s, int depth, const char *key) {
    sb_indent(s, depth);
    sb_put(s, "\"", 1);
    sb_puts(s, key);
    sb_puts(s, "\": ");
}
static void js_u64_field(sbuf *s, int depth, const char *key, uint64_t v, int last) {
    js_key(s, depth, key);
    sb_u64(s, v);
    sb_puts(s, last ? "\n" : ",\n");
}
static void js_f64_field(sbuf *s, int depth, const char *key, double v, int last) {
    js_key(s, depth, key);
    sb_float(s, v, 0);
    sb_puts(s, last ? "\n" : ",\n");
}
/* histogram.rs:152-159 serde field order: values, range_start, range_stop */
static void js_histogram(sbuf *s, int depth, const uint64_t *values, uint64_t stop) {
    sb_puts(s, "{\n");
    js_key(s, depth + 1, "values");
    sb_puts(s, "[\n");
    for (uint64_t i = 0; i <= stop; i++) {
        sb_indent(s, depth + 2);
        sb_u64(s, values[i]);
        sb_puts(s, i == stop ? "\n" : ",\n");
    }
    sb_indent(s, depth + 1);
    sb_puts(s, "],\n");
    js_u64_field(s, depth + 1, "range_start", 0, 0);
    js_u64_field(s, depth + 1, "range_stop", stop, 1);
    sb_indent(s, depth);
    sb_puts(s, "}");
}

static const char CIGAR_LETTERS[NGSQ_N_CIGAR_KINDS + 1] = "MIDNSHP=X";

static void js_cigar_map(sbuf *s, int depth, const char *key, const uint64_t *ops, int last) {
    /* general/metrics.rs:99-103: only kinds that were seen have a key; sorted by letter */
    static const int order[NGSQ_N_CIGAR_KINDS] = {7, 2, 5, 1, 0, 3, 6, 4, 8}; /* = D H I M N P S X */
    js_key(s, depth, key);
    int n = 0;
    for (int i = 0; i < NGSQ_N_CIGAR_KINDS; i++) n += ops[i] != 0;
    if (!n) {
        sb_puts(s, last ? "{}\n" : "{},\n");
        return;
    }
    sb_puts(s, "{\n");
    int done = 0;
    for (int j = 0; j < NGSQ_N_CIGAR_KINDS; j++) {
        int op = order[j];
        if (!ops[op]) continue;
        char k[2] = {CIGAR_LETTERS[op], 0};
        js_u64_field(s, depth + 1, k, ops[op], ++done == n);
    }
    sb_indent(s, depth);
    sb_puts(s, last ? "}\n" : "},\n");
}

int64_t orc_results_json(const orc_ctx *c, const char *const *ref_names, char *buf, size_t cap) {
    if (!c || !c->finalized) return NGSQ_ERR_STATE;
    sbuf sb = {0, 0, 0};
    sbuf *s = &sb;
    uint32_t facets = c->cfg.facets;
    sb_puts(s, "{\n");
    /* results.rs:23-45 field order */
    js_key(s, 1, "general");
    if (facets & NGSQ_FACET_GENERAL) {
        const ngsq_general_metrics *m = &c->general;
        sb_puts(s, "{\n");
        js_key(s, 2, "records");
        sb_puts(s, "{\n");
        js_u64_field(s, 3, "total", m->total, 0);
        js_u64_field(s, 3, "unmapped", m->unmapped, 0);
        js_u64_field(s, 3, "duplicate", m->duplicate, 0);
        js_key(s, 3, "designation");
        sb_puts(s, "{\n");
        js_u64_field(s, 4, "primary", m->primary, 0);
        js_u64_field(s, 4, "secondary", m->secondary, 0);
        js_u64_field(s, 4, "supplementary", m->supplementary, 1);
        sb_indent(s, 3);
        sb_puts(s, "},\n");
        js_u64_field(s, 3, "primary_mapped", m->primary_mapped, 0);
        js_u64_field(s, 3, "primary_duplicate", m->primary_duplicate, 0);
        js_u64_field(s, 3, "paired", m->paired, 0);
        js_u64_field(s, 3, "read_1", m->read_1, 0);
        js_u64_field(s, 3, "read_2", m->read_2, 0);
        js_u64_field(s, 3, "proper_pair", m->proper_pair, 0);
        js_u64_field(s, 3, "singleton", m->singleton, 0);
        js_u64_field(s, 3, "mate_mapped", m->mate_mapped, 0);
        js_u64_field(s, 3, "mate_reference_sequence_id_mismatch",
                     m->mate_reference_sequence_id_mismatch, 0);
        js_u64_field(s, 3, "mate_reference_sequence_id_mismatch_hq",
                     m->mate_reference_sequence_id_mismatch_hq, 1);
        sb_indent(s, 2);
        sb_puts(s, "},\n");
        js_key(s, 2, "cigar");
        sb_puts(s, "{\n");
        js_cigar_map(s, 3, "read_one_cigar_ops", m->read_one_cigar_ops, 0);
        js_cigar_map(s, 3, "read_two_cigar_ops", m->read_two_cigar_ops, 1);
        sb_indent(s, 2);
        sb_puts(s, "},\n");
        js_key(s, 2, "summary");
        sb_puts(s, "{\n");
        js_f64_field(s, 3, "duplication_pct", c->duplication_pct, 0);
        js_f64_field(s, 3, "mapped_pct", c->mapped_pct, 0);
        js_f64_field(s, 3, "mate_reference_sequence_id_mismatch_pct", c->mismatch_pct, 0);
        js_f64_field(s, 3, "mate_reference_sequence_id_mismatch_hq_pct", c->mismatch_hq_pct, 1);
        sb_indent(s, 2);
        sb_puts(s, "}\n");
        sb_indent(s, 1);
        sb_puts(s, "},\n");
    } else {
        sb_puts(s, "null,\n");
    }
    js_key(s, 1, "features");
    if (facets & NGSQ_FACET_FEATURES) { /* features/metrics.rs:66-80 */
        const ngsq_features_metrics *m = &c->features;
        sb_puts(s, "{\n");
        js_key(s, 2, "exonic_translation_regions");
        sb_puts(s, "{\n");
        js_u64_field(s, 3, "utr_five_prime_count", m->utr_five_prime_count, 0);
        js_u64_field(s, 3, "utr_three_prime_count", m->utr_three_prime_count, 0);
        js_u64_field(s, 3, "coding_sequence_count", m->coding_sequence_count, 1);
        sb_indent(s, 2);
        sb_puts(s, "},\n");
        js_key(s, 2, "gene_regions");
        sb_puts(s, "{\n");
        js_u64_field(s, 3, "intergenic_count", m->intergenic_count, 0);
        js_u64_field(s, 3, "exonic_count", m->exonic_count, 0);
        js_u64_field(s, 3, "intronic_count", m->intronic_count, 1);
        sb_indent(s, 2);
        sb_puts(s, "},\n");
        js_key(s, 2, "records");
        sb_puts(s, "{\n");
        js_u64_field(s, 3, "processed", m->processed, 0);
        js_u64_field(s, 3, "ignored_flags", m->ignored_flags, 0);
        js_u64_field(s, 3, "ignored_nonprimary_chromosome", m->ignored_nonprimary_chromosome, 1);
        sb_indent(s, 2);
        sb_puts(s, "},\n");
        js_key(s, 2, "summary");
        sb_puts(s, "{\n");
        js_f64_field(s, 3, "ignored_flags_pct", c->feat_ignored_flags_pct, 0);
        js_f64_field(s, 3, "ignored_nonprimary_chromosome_pct", c->feat_ignored_nonprimary_pct, 1);
        sb_indent(s, 2);
        sb_puts(s, "}\n");
        sb_indent(s, 1);
        sb_puts(s, "},\n");
    } else {
        sb_puts(s, "null,\n");
    }
    js_key(s, 1, "gc_content");
    if (facets & NGSQ_FACET_GC_CONTENT) {
        sb_puts(s, "{\n");
        js_key(s, 2, "histogram");
        js_histogram(s, 2, c->gc_hist.values, 100);
        sb_puts(s, ",\n");
        js_key(s, 2, "nucleobases");
        sb_puts(s, "{\n");
        js_u64_field(s, 3, "total_gc_count", c->gc_count, 0);
        js_u64_field(s, 3, "total_at_count", c->at_count, 0);
        js_u64_field(s, 3, "total_other_count", c->other_count, 1);
        sb_indent(s, 2);
        sb_puts(s, "},\n");
        js_key(s, 2, "records");
        sb_puts(s, "{\n");
        js_u64_field(s, 3, "processed", c->gc_processed, 0);
        js_u64_field(s, 3, "ignored_flags", c->gc_ignored_flags, 0);
        js_u64_field(s, 3, "ignored_too_short", c->gc_ignored_too_short, 1);
        sb_indent(s, 2);
        sb_puts(s, "},\n");
        js_key(s, 2, "summary");
        sb_puts(s, "{\n");
        js_f64_field(s, 3, "gc_content_pct", c->gc_content_pct, 0);
        js_f64_field(s, 3, "ignored_flags_pct", c->gc_ignored_flags_pct, 0);
        js_f64_field(s, 3, "ignored_too_short_pct", c->gc_ignored_too_short_pct, 1);
        sb_indent(s, 2);
        sb_puts(s, "}\n");
        sb_indent(s, 1);
        sb_puts(s, "},\n");
    } else {
        sb_puts(s, "null,\n");
    }
    js_key(s, 1, "template_length");
    if (facets & NGSQ_FACET_TEMPLATE_LENGTH) {
        sb_puts(s, "{\n");
        js_key(s, 2, "histogram");
        js_histogram(s, 2, c->tlen_hist.values, c->cfg.tlen_cap);
        sb_puts(s, ",\n");
        js_key(s, 2, "records");
        sb_puts(s, "{\n");
        js_u64_field(s, 3, "processed", c->tlen_processed, 0);
        js_u64_field(s, 3, "ignored", c->tlen_ignored, 1);
        sb_indent(s, 2);
        sb_puts(s, "},\n");
        js_key(s, 2, "summary");
        sb_puts(s, "{\n");
        js_f64_field(s, 3, "template_length_unknown_pct", c->tlen_unknown_pct, 0);
        js_f64_field(s, 3, "template_length_out_of_range_pct", c->tlen_out_of_range_pct, 1);
        sb_indent(s, 2);
        sb_puts(s, "}\n");
        sb_indent(s, 1);
        sb_puts(s, "},\n");
    } else {
        sb_puts(s, "null,\n");
    }
    js_key(s, 1, "quality_scores");
    if (facets & NGSQ_FACET_QUALITY_SCORE) {
        sb_puts(s, "{\n");
        js_key(s, 2, "scores");
        uint32_t n = 0;
        for (uint32_t i = 0; i < c->cfg.max_read_len; i++) n += c->scores[i].values != NULL;
        if (!n) {
            sb_puts(s, "{}\n");
        } else {
            sb_puts(s, "{\n");
            uint32_t done = 0;
            for (uint32_t i = 0; i < c->cfg.max_read_len; i++) {
                if (!c->scores[i].values) continue;
                char k[16];
                snprintf(k, sizeof k, "%u", i + 1);
                js_key(s, 3, k);
                js_histogram(s, 3, c->scores[i].values, NGSQ_MAX_SCORE);
                sb_puts(s, ++done == n ? "\n" : ",\n");
            }
            sb_indent(s, 2);
            sb_puts(s, "}\n");
        }
        sb_indent(s, 1);
        sb_puts(s, "},\n");
    } else {
        sb_puts(s, "null,\n");
    }
    js_key(s, 1, "coverage");
    if (facets & NGSQ_FACET_COVERAGE) {
        const orc_coverage *cv = &c->cov;
        uint32_t n = 0, nr = c->cfg.n_refs;
        for (uint32_t r = 0; r < nr; r++) n += cv->has_entry[r] != 0;
        sb_puts(s, "{\n");
        const char *f64_maps[1] = {"mean_coverage"};
        (void)f64_maps;
        /* mean_coverage */
        for (int which = 0; which < 4; which++) {
            static const char *names[4] = {"mean_coverage", "mean_coverage_per_bin",
                                           "median_coverage", "median_over_mean_coverage"};
            js_key(s, 2, names[which]);
            if (!n) {
                sb_puts(s, "{},\n");
                continue;
            }
            sb_puts(s, "{\n");
            uint32_t done = 0;
            for (uint32_t r = 0; r < nr; r++) {
                if (!cv->has_entry[r]) continue;
                js_key(s, 3, ref_names[r]);
                if (which == 1) {
                    sb_puts(s, "[\n");
                    for (uint64_t b = 0; b < cv->n_bins[r]; b++) {
                        sb_indent(s, 4);
                        sb_float(s, cv->mean_coverage_per_bin[r][b], 0);
                        sb_puts(s, b + 1 == cv->n_bins[r] ? "\n" : ",\n");
                    }
                    sb_indent(s, 3);
                    sb_puts(s, "]");
                } else {
                    double v = which == 0   ? cv->mean_coverage[r]
                               : which == 2 ? cv->median_coverage[r]
                                            : cv->median_over_mean[r];
                    sb_float(s, v, 0);
                }
                sb_puts(s, ++done == n ? "\n" : ",\n");
            }
            sb_indent(s, 2);
            sb_puts(s, "},\n");
        }
        js_key(s, 2, "ignored");
        sb_puts(s, "{\n");
        js_u64_field(s, 3, "nonsensical_records", cv->nonsensical_records, 0);
        js_key(s, 3, "pileup_too_large_positions");
        if (!n) {
            sb_puts(s, "{}\n");
        } else {
            sb_puts(s, "{\n");
            uint32_t done = 0;
            for (uint32_t r = 0; r < nr; r++) {
                if (!cv->has_entry[r]) continue;
                js_u64_field(s, 4, ref_names[r], cv->pileup_too_large[r], ++done == n);
            }
            sb_indent(s, 3);
            sb_puts(s, "}\n");
        }
        sb_indent(s, 2);
        sb_puts(s, "},\n");
        js_key(s, 2, "coverage_distribution");
        js_histogram(s, 2, cv->coverage_distribution.values, c->cfg.cov_cap);
        sb_puts(s, ",\n");
        js_key(s, 2, "genome_covered_by");
        sb_puts(s, "{\n");
        static const char *cx[6] = {"10x", "20x", "30x", "40x", "50x", "60x"};
        for (int k = 0; k < 6; k++) {
            js_key(s, 3, cx[k]);
            sb_float(s, (double)cv->genome_covered_by[k], 1);
            sb_puts(s, k == 5 ? "\n" : ",\n");
        }
        sb_indent(s, 2);
        sb_puts(s, "}\n");
        sb_indent(s, 1);
        sb_puts(s, "},\n");
    } else {
        sb_puts(s, "null,\n");
    }
    js_key(s, 1, "edits");
    if (facets & NGSQ_FACET_EDITS) {
        const orc_edits *e = &c->edits;
        sb_puts(s, "{\n");
        js_key(s, 2, "read_one_edits");
        js_histogram(s, 2, e->read_one_edits.values, 512);
        sb_puts(s, ",\n");
        js_key(s, 2, "read_two_edits");
        js_histogram(s, 2, e->read_two_edits.values, 512);
        sb_puts(s, ",\n");
        js_key(s, 2, "vaf_histogram");
        js_histogram(s, 2, e->vaf_histogram.values, 100);
        sb_puts(s, ",\n");
        js_key(s, 2, "summary");
        sb_puts(s, "{\n");
        js_f64_field(s, 3, "mean_edits_read_one", e->mean_edits_read_one, 0);
        js_f64_field(s, 3, "mean_edits_read_two", e->mean_edits_read_two, 1);
        sb_indent(s, 2);
        sb_puts(s, "}\n");
        sb_indent(s, 1);
        sb_puts(s, "}\n");
    } else {
        sb_puts(s, "null\n");
    }
    sb_puts(s, "}");
    int64_t need = (int64_t)sb.len;
    if (buf && cap) {
        size_t n = sb.len < cap - 1 ? sb.len : cap - 1;
        memcpy(buf, sb.p, n);
        buf[n] = 0;
    }
    free(sb.p);
    return need;
}
