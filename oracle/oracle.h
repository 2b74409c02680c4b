/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the
 * product: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may build, load or run it, and only as the checker / reported baseline.
 *
 * CPU restatement (plain C, single thread, record at a time, two passes) of
 * the reference's `ngs qc` facet loop.  Each function cites the reference
 * file:line it follows (paths relative to /root/reference).
 *
 * PINNING STATUS
 *   - Histogram (histogram.c) and the CIGAR/reference walk (orc_stepthrough)
 *     are pinned against every known-answer test the reference holds for
 *     them (src/utils/histogram.rs:405-523, src/utils/alignment.rs:134-202,
 *     src/qc/record_based/gc_content.rs:145-151): tests/test_oracle_kat.py.
 *   - The facets' process/summarize/teardown/aggregate have NO tests, golden
 *     files or fixtures in the reference, and the
 *     reference is Rust, which cannot be built here (no cargo/rustc, crates
 *     not vendored, no network): for them this oracle is PARITY UNPINNED --
 *     a line-by-line restatement checked against hand-derived goldens
 *     (tests/golden/hand_six_records.json: the record facets and one covered
 *     sequence; tests/golden/hand_edits_multiseq.json: the Edits walk over every
 *     CIGAR operation, several sequences, a pileup beyond the capacity, the f32
 *     VAF edge) for the arithmetic, and, for what the
 *     reference's SOURCE fixes without running it, against fixtures derived from
 *     that source by scripts committed next to them (tests/golden/make_*.py):
 *     the shape of the Results document (field names, declaration order, types),
 *     the fixed capacities / thresholds / facet names, the error texts.
 *   - The record decode the reference delegates to noodles-bam 0.28.0 /
 *     noodles-sam 0.25.0 (Cargo.lock:902-905,1057-1059) is not in the tree;
 *     the accessor semantics used here are restated from the SAM/BAM
 *     specification.  ALL of them, in one place (each is UNVERIFIED against the
 *     crate's source, which is not in this container; oracle.c marks the lines that
 *     rely on one with its tag):
 *       [N1] reference_sequence_id() / mate_reference_sequence_id(): None for -1
 *            (general.rs:81-83 unwrap()s them: counted as missing_reference_id);
 *       [N2] mapping_quality(): None for 255; general.rs:88-95 maps None to 255,
 *            so a missing MAPQ counts as high quality;
 *       [N3] alignment_start(): None for pos -1, else pos + 1 (1-based);
 *       [N4] alignment_end() = start + span - 1 with span = sum of the lengths of
 *            M D N = X; None when that is 0 (start 1, no reference-consuming op);
 *       [N5] Reader::query(region = whole sequence) yields a record iff its
 *            reference id is the sequence, start and end are Some and [start, end]
 *            meets [1, L]: a placed read without CIGAR (span 0) at start >= 2 is
 *            yielded and covers nothing; a read starting beyond L is not;
 *       [N6] quality scores: BAM's 0xFF-filled QUAL decodes to an EMPTY score list
 *            (no increments, quality_scores.rs:38); a score above 93 is a decode
 *            error that aborts the run (counted as bad_quality_score);
 *       [N7] sequence bases decode to the 16 codes "=ACMGRSVTWYHKDBN"; GC Content
 *            counts C/G, A/T and "other" on them (gc_content.rs:77-87);
 *       [N8] CIGAR op codes 0..8 = MIDNSHP=X; a code above 8 is a decode error
 *            (counted as bad_cigar_op);
 *       [N9] FASTA bytes become bases through Base::try_from(u8) (edits.rs:257-261) and that conversion FOLDS CASE:
 *            'a' is Base::A.  DECIDED in round 6 (it was "upper case only, unverified" until then, and the command line
 *            refused lower case).  Grounds: (i) noodles-sam's `impl TryFrom<char> for Base` matches on
 *            `c.to_ascii_uppercase()` -- restated from the published crate source, which is not in this container;
 *            (ii) the reference's own code expects lower-case FASTA bytes to reach it: src/generate/utils.rs:96-106
 *            complements 'a' 'c' 'g' 't' next to 'A' 'C' 'G' 'T' on sequences read with the same noodles-fasta reader;
 *            (iii) the genome the command's argument is named after, the GRCh38 no-alt analysis set, is soft-masked --
 *            about half of its bases are lower case -- so the other reading makes `ngs qc -r` fail on line 2 of chr1 of the
 *            one FASTA it is documented for.  Any other byte (not one of "=ACMGRSVTWYHKDBN" in either case) is a
 *            TryFromCharError for the record whose slice start..start+span holds it, and for no other record: the
 *            conversion runs per record over its slice.  orc_fasta_base_code restates the conversion; a refused byte is
 *            kept in ref_bases as a code above 15 (tests/golden/hand_softmasked.json: worked by hand);
 *      [N10] a CIGAR of more than 65535 operations (SAM specification 4.2.2): the BAM record
 *            holds the placeholder <l_seq>S<reference span>N and the real operations in a CG:B,I
 *            tag, which noodles-bam resolves while decoding the record (as far as can be told
 *            without its source: the crate gained this in its 0.1x releases).  The facets then
 *            see the real CIGAR (general.rs:103-121 tallies its operations, coverage.rs:159-160
 *            its span, edits.rs walks it), and so does this restatement: the readers of
 *            ngsq_bam.h hand the tag's operations on (the batch's n_cigar column is 16 bits wide
 *            and saturates at 65535; the count of such a record is in cigar_off, which is what
 *            orc_view_record uses whenever a batch has offsets).  The placeholder WITHOUT a tag
 *            is an alignment like any other.  tests/golden/hand_longcigar.bam: expectations
 *            worked by hand.  Aux tags are otherwise skipped unread: no facet looks at them.
 *     The hand goldens and the synthetic workloads stay away from the corner cases
 *     of [N4] and [N5] (span 0, start beyond L) except where a test names them.
 *   - A SECOND READING (round 6).  tests/literal_model.py restates the same facets once more, from the .rs files
 *     alone, in the reference's own shape (a record at a time, dicts for its HashMaps, a Histogram class with its
 *     methods), and tests/test_literal_model.py holds the two restatements against each other: the same Results
 *     document on random records (all seven facets, FASTA lengths that differ from LN, soft-masked and refused bytes,
 *     roles that share names), and "the model stops at a record" <=> "this oracle counts an error".  It does not lift
 *     "parity unpinned" -- both are readings of one source, and [N1]..[N10] are shared assumptions -- but a slip in
 *     either shows.  Its first 800 random settings found one in THIS file: every read that ended beyond @SQ LN inside a
 *     longer FASTA sequence was counted as a run the reference aborts, where edits.rs only panics for an `M` base there
 *     (increment().unwrap(), :283-291); bases beyond LN under D, N, = or X go through.  Oracle, kernels and loader now
 *     keep the exact rule (tests/test_reference.py::test_reads_beyond_ln_inside_a_longer_fasta_sequence).
 */
#ifndef ORC_ORACLE_H
#define ORC_ORACLE_H

#include "../include/ngsq.h"
#include "histogram.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_ctx orc_ctx;

/* a decoded record as the facets see it (noodles sam::alignment::Record accessors) */
typedef struct orc_record {
    uint16_t flag;
    uint8_t mapq;
    int32_t ref_id;
    int32_t pos;
    int32_t mate_ref_id;
    int32_t tlen;
    uint32_t l_seq;
    const uint8_t *seq;  /* packed 4-bit */
    uint32_t n_qual;     /* l_seq, or 0 when qualities are missing */
    const uint8_t *qual;
    int qual_fixed_row;  /* row of a fixed-stride batch: 0xFF bytes are absent scores */
    uint32_t n_cigar;
    const uint32_t *cigar;
    uint64_t index;      /* index of the record in the file (GC offset fn) */
} orc_record;

orc_ctx *orc_create(const ngsq_config *cfg);
void orc_destroy(orc_ctx *ctx);
const char *orc_last_error(const orc_ctx *ctx);

/* command.rs:305-316 (pass 1) and :356-397 (pass 2) over one SoA batch */
int orc_process_batch(orc_ctx *ctx, const ngsq_batch *batch, uint32_t pass_mask);
uint32_t orc_quality_rows(const orc_ctx *ctx); /* 1 + the last cycle any read reached */
/* summarize (command.rs:328-330), teardown per sequence (:392-396), aggregate (:406-414) */
int orc_finalize(orc_ctx *ctx);

/* GenomicFeaturesFacet::try_from's interval stores (features.rs:300-343) */
int orc_set_features(orc_ctx *ctx, const ngsq_features *features);
int orc_get_features(const orc_ctx *ctx, ngsq_features_metrics *out);
int orc_get_error_counts(const orc_ctx *ctx, ngsq_error_counts *out);
int orc_get_general(const orc_ctx *ctx, ngsq_general_metrics *out);
int orc_get_template_length(const orc_ctx *ctx, uint64_t *histogram, size_t n_bins,
                            uint64_t *processed, uint64_t *ignored);
int orc_get_gc_content(const orc_ctx *ctx, ngsq_gc_metrics *out);
int orc_get_quality_scores(const orc_ctx *ctx, uint64_t *scores, size_t n_rows);
uint64_t orc_coverage_n_bins(const orc_ctx *ctx, uint32_t ref);
int orc_get_coverage_sequence(const orc_ctx *ctx, uint32_t ref, int *seen, uint64_t *histogram,
                              size_t n_hist_bins, uint64_t *ignored, double *bin_means,
                              size_t n_bins);
int orc_get_coverage_nonsensical(const orc_ctx *ctx, uint64_t *nonsensical_records);
int orc_get_edits(const orc_ctx *ctx, uint64_t *read_one_edits, uint64_t *read_two_edits,
                  size_t n_edit_bins, uint64_t *vaf_histogram, size_t n_vaf_bins);
int64_t orc_results_json(const orc_ctx *ctx, const char *const *ref_names, char *buf, size_t cap);

/* alignment.rs:48-107 + :113-125: edits between a reference slice and a record
 * (bases as 4-bit codes, one per byte).  Returns 0 and *edits, or
 * 1 = "...consume a reference base...", 2 = "...consume a record base...",
 * 3 = "reference sequence was not fully consumed",
 * 4 = "record sequence was not fully consumed". */
int orc_stepthrough_edits(const uint8_t *reference, size_t n_reference, const uint8_t *record,
                          size_t n_record, const uint32_t *cigar, size_t n_cigar,
                          uint64_t *edits);
const char *orc_stepthrough_error_message(int code);

/* [N9] Base::try_from(u8) on a FASTA byte: the 4-bit BAM code, or -1 (the byte is refused) */
int orc_fasta_base_code(uint8_t byte);

/* timing helper for bench.py's cpu_baseline leg: seconds of CPU work spent in
 * orc_process_batch + orc_finalize since orc_create */
double orc_elapsed_seconds(const orc_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif
