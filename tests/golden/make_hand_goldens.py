"""Writes tests/golden/hand_six_records.json.

HAND-DERIVED golden: six records whose expected facet results were
worked out on paper from the reference's source (general.rs:31-153,
template_length.rs:79-100, gc_content.rs:38-122, quality_scores.rs:37-49,
coverage.rs:148-287).  No reference binary exists in this environment (Rust,
no cargo), so these are NOT outputs of the reference; they pin the oracle to an
independent reading of the code.  The expected values are literals below, not
computed by any implementation in this repository.
"""
import json
import os

records = [
    # A: paired, proper, mate reverse, read 1
    dict(name="A", flag=0x063, mapq=60, ref_id=0, pos=99, mate_ref_id=0, tlen=350, cigar="150M",
         seq="C" * 150, qual=[30] * 150),
    # B: paired, proper, reverse, read 2, soft clip
    dict(name="B", flag=0x093, mapq=60, ref_id=0, pos=199, mate_ref_id=0, tlen=-350, cigar="100M50S",
         seq="A" * 150, qual=[20] * 150),
    # C: paired, unmapped, mate unmapped, read 1, unplaced, missing qualities
    dict(name="C", flag=0x04D, mapq=0, ref_id=-1, pos=-1, mate_ref_id=-1, tlen=0, cigar="*",
         seq="ACGT" * 37 + "AC", qual=None),
    # D: paired, read 1, duplicate, mate on another sequence, low mapq, runs past the end of chr1
    dict(name="D", flag=0x441, mapq=3, ref_id=0, pos=899, mate_ref_id=1, tlen=0, cigar="150M",
         seq="G" * 150, qual=[40] * 150),
    # E: secondary, unpaired, missing mapq
    dict(name="E", flag=0x100, mapq=255, ref_id=0, pos=99, mate_ref_id=-1, tlen=0, cigar="150M",
         seq="T" * 150, qual=[10] * 150),
    # F: supplementary, paired, read 1, hard clip, exactly 100 bases of N
    dict(name="F", flag=0x841, mapq=60, ref_id=0, pos=299, mate_ref_id=0, tlen=0, cigar="50H100M",
         seq="N" * 100, qual=[93] * 100),
]

scores = {}
for cycle in range(1, 151):
    v = [0] * 94
    v[30] = v[20] = v[40] = v[10] = 1
    if cycle <= 100:
        v[93] = 1
    scores[str(cycle)] = {"values": v, "range_start": 0, "range_stop": 93}

tlen_hist = [0] * 1025
tlen_hist[0] = 4
tlen_hist[350] = 1
gc_hist = [0] * 101
gc_hist[0] = 2     # B (all A), F (all N)
gc_hist[50] = 1    # C (ACGT repeats: any 100-base window holds 50 G/C)
gc_hist[100] = 1   # A (all C)
cov_dist = [0] * 2049
cov_dist[0], cov_dist[1], cov_dist[2], cov_dist[3] = 600, 251, 100, 50

expected = {
    "general": {
        "records": {
            "total": 6, "unmapped": 1, "duplicate": 1,
            "designation": {"primary": 4, "secondary": 1, "supplementary": 1},
            "primary_mapped": 3, "primary_duplicate": 1, "paired": 4, "read_1": 3, "read_2": 1,
            "proper_pair": 2, "singleton": 0, "mate_mapped": 3,
            "mate_reference_sequence_id_mismatch": 1, "mate_reference_sequence_id_mismatch_hq": 0,
        },
        "cigar": {"read_one_cigar_ops": {"M": 3, "H": 1}, "read_two_cigar_ops": {"M": 2, "S": 1}},
        "summary": {
            "duplication_pct": 16.666666666666664, "mapped_pct": 83.33333333333334,
            "mate_reference_sequence_id_mismatch_pct": 16.666666666666664,
            "mate_reference_sequence_id_mismatch_hq_pct": 0.0,
        },
    },
    "features": None,
    "gc_content": {
        "histogram": {"values": gc_hist, "range_start": 0, "range_stop": 100},
        "nucleobases": {"total_gc_count": 150, "total_at_count": 150, "total_other_count": 100},
        "records": {"processed": 4, "ignored_flags": 2, "ignored_too_short": 0},
        "summary": {"gc_content_pct": 37.5, "ignored_flags_pct": 33.33333333333333,
                    "ignored_too_short_pct": 0.0},
    },
    "template_length": {
        "histogram": {"values": tlen_hist, "range_start": 0, "range_stop": 1024},
        "records": {"processed": 5, "ignored": 1},
        "summary": {"template_length_unknown_pct": 66.66666666666666,
                    "template_length_out_of_range_pct": 16.666666666666664},
    },
    "quality_scores": {"scores": scores},
    # chr1 L=1000, chr2 L=500, bin size 400.  Depth over chr1: 100..199 -> 2 (A,E),
    # 200..249 -> 3 (A,E,B), 250..299 -> 1 (B), 300..399 -> 1 (F), 900..1000 -> 1 (D);
    # D's positions 1001..1049 are 49 nonsensical increments.  1001 positions (0..=1000).
    "coverage": {
        "mean_coverage": {"chr1": 601 / 1001},
        "mean_coverage_per_bin": {"chr1": [0.0, 1.25, 0.0, 0.505]},
        "median_coverage": {"chr1": 0.0},
        "median_over_mean_coverage": {"chr1": 0.0},
        "ignored": {"nonsensical_records": 49, "pileup_too_large_positions": {"chr1": 0}},
        "coverage_distribution": {"values": cov_dist, "range_start": 0, "range_stop": 2048},
        "genome_covered_by": {"10x": 0.0, "20x": 0.0, "30x": 0.0, "40x": 0.0, "50x": 0.0, "60x": 0.0},
    },
    "edits": None,
}

doc = {
    "description": "hand-derived golden, see make_hand_goldens.py",
    "config": {"ref_names": ["chr1", "chr2"], "ref_len": [1000, 500], "ref_is_primary": [1, 1],
               "bin_size": 400, "max_read_len": 150, "facets": 0x1F},
    "records": records,
    "expected": expected,
}
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hand_six_records.json")
with open(out, "w") as f:
    json.dump(doc, f, indent=1)
print(out)
