"""Writes tests/golden/hand_edits_multiseq.json.

HAND-DERIVED golden, second case: the Edits walk over every CIGAR operation, the Coverage
teardown over several sequences, and the f32 / f64 roundings, worked out on paper from the reference's source
(edits.rs:217-353, utils/alignment.rs:48-107, utils/cigar.rs:6-23, coverage.rs:148-287, histogram.rs:258-337,
general.rs:31-153).  No reference binary exists in this environment (Rust, no cargo): these are NOT outputs of
the reference; they pin the oracle -- and through tests/test_parity_gpu.py the HIP path -- to an independent
reading of the code.  The expected values are literals below, not computed by any implementation in this
repository; the arithmetic behind each is in the comments.

Sequences (header order): chrA L=60 primary, chrB L=40 primary, chrN L=30 NOT primary, chrE L=50 primary, no records.
Reference bases: every sequence is "ACGT" repeated, i.e. position p (1-based) holds "ACGT"[(p-1) % 4].
Coverage bin size 25; everything else at the reference's constants (cov_cap 2048, tlen 1024 ...).

Records
  R1  chrA start 1, read 1 of a proper pair (0x43), CIGAR 1H 2S 4M 1I 2M 2D 2= 1X 3N 2M 1P 1M, 15 bases
        S  NN                       (no reference)
        4M ACGA  vs  ACGT  pos 1-4  -> refs 1,2,3; ALT 4
        1I T
        2M AC    vs  AC    pos 5-6  -> refs 5,6
        2D               pos 7-8    (no record base; nothing counted)
        2= TT    vs  AC    pos 9-10 -> NOT compared: only Kind::Match is (edits.rs:277), although the bases differ
        1X G     vs  G     pos 11   -> not compared either
        3N               pos 12-14
        2M GA    vs  GT    pos 15-16-> ref 15; ALT 16
        1P
        1M C     vs  A     pos 17   -> ALT 17
      3 edits, first segment -> read_one_edits[3].  Reference span 4+2+2+2+1+3+2+1 = 17: Coverage 1..=17 (D and N count).
  R2  chrA start 3, read 2 of a proper pair (0x83), 6M GTACGT = reference 3..8 -> refs 3..8, 0 edits -> read_two_edits[0]
  R3  chrA start 4, unpaired (0), 1M A vs T -> ALT 4, 1 edit; not a first segment -> read_two_edits[1]
  2100 x  chrB start 10, unpaired, 1M C vs C -> refs[10] = 2100 (depth 2100 > 2048: a pileup too large)
  47 x    chrB start 12, unpaired, 1M T vs T;  53 x the same with G -> refs[12] = 47, alts[12] = 53
  D1  chrB start 20, duplicate (0x400), 3M: Coverage counts it (no flag filter, coverage.rs:148-180), Edits skips it
  N1  chrN start 5, unpaired, 4M ACGT = reference 5..8: Edits processes every sequence (edits.rs:173-175),
      Coverage only primary ones (coverage.rs:133-138)
"""
import json
import os

one = lambda **kw: dict(dict(flag=0, mapq=60, mate_ref_id=-1, tlen=0), **kw)  # noqa: E731
records = [
    one(name="R1", flag=0x43, ref_id=0, pos=0, mate_ref_id=0, cigar="1H2S4M1I2M2D2=1X3N2M1P1M", seq="NNACGATACTTGGAC", qual=[30] * 15),
    one(name="R2", flag=0x83, ref_id=0, pos=2, mate_ref_id=0, cigar="6M", seq="GTACGT", qual=[30] * 6),
    one(name="R3", ref_id=0, pos=3, cigar="1M", seq="A", qual=[30]),
]
records += [one(ref_id=1, pos=9, cigar="1M", seq="C", qual=[30]) for _ in range(2100)]
records += [one(ref_id=1, pos=11, cigar="1M", seq="T", qual=[30]) for _ in range(47)]
records += [one(ref_id=1, pos=11, cigar="1M", seq="G", qual=[30]) for _ in range(53)]
records += [one(name="D1", flag=0x400, ref_id=1, pos=19, cigar="3M", seq="TTT", qual=[30] * 3)]
records += [one(name="N1", ref_id=2, pos=4, cigar="4M", seq="ACGT", qual=[30] * 4)]


def hist(n, **bins):
    v = [0] * n
    for k, c in bins.items():
        v[int(k[1:])] = c
    return v


# ---- General (general.rs:31-153).  2205 records, all primary and mapped; one duplicate; R1 and R2 are the only paired ones.
general = {
    "records": {
        "total": 2205, "unmapped": 0, "duplicate": 1,
        "designation": {"primary": 2205, "secondary": 0, "supplementary": 0},
        "primary_mapped": 2205, "primary_duplicate": 1, "paired": 2, "read_1": 1, "read_2": 1,
        "proper_pair": 2, "singleton": 0, "mate_mapped": 2,
        "mate_reference_sequence_id_mismatch": 0, "mate_reference_sequence_id_mismatch_hq": 0,
    },
    # R1 carries every operation once, and M four times; all other records count as "read two" (not first segments):
    # R2, R3, 2200 one-base reads, D1, N1 -> 2204 M
    "cigar": {"read_one_cigar_ops": {"H": 1, "S": 1, "M": 4, "I": 1, "D": 1, "=": 1, "X": 1, "N": 1, "P": 1},
              "read_two_cigar_ops": {"M": 2204}},
    "summary": {"duplication_pct": 0.045351473922902494,   # 1 / 2205 * 100 in f64
                "mapped_pct": 100.0,
                "mate_reference_sequence_id_mismatch_pct": 0.0,
                "mate_reference_sequence_id_mismatch_hq_pct": 0.0},
}

# ---- Coverage (coverage.rs:148-287), bin size 25
# chrA depth: 1,2 -> 1; 3 -> 2 (R1,R2); 4 -> 3 (R1,R2,R3); 5..8 -> 2; 9..17 -> 1; position 0 and 18..60 -> 0
#   61 positions: depth 0 x 44, 1 x 11, 2 x 5, 3 x 1;  mean = (11 + 10 + 3) / 61;  median: 44 > 30.5 in bin 0 -> 0.0
#   bins: [0.0 (position 0 alone), (1+1+2+3+2+2+2+2+9)/25 = 24/25, 0/25, tail 60 % 25 = 10 positions: 0/10]
# chrB depth: 10 -> 2100 (too large: ignored), 12 -> 100, 20..22 -> 1, the other 36 of 41 positions -> 0
#   coverages: 0 x 36, 1 x 3, 100 x 1 (40 values);  mean = 103 / 40;  median: 36 > 20 in bin 0 -> 0.0
#   bins: [0.0, (2100 + 100 + 3)/25 = 88.12 (the too-large position still counts here), tail 40 % 25 = 15: 0/15]
# chrN is not primary, chrE has no record: neither appears anywhere
# distribution: 0 x 80, 1 x 14, 2 x 5, 3 x 1, 100 x 1 = 101;  total positions = 101 + 1 too large = 102
# genome_covered_by (f32): one position (depth 100) reaches 10x..60x: 1f32 / 102f32 * 100f32 = 0.9803922
coverage = {
    "mean_coverage": {"chrA": 0.39344262295081966, "chrB": 2.575},
    "mean_coverage_per_bin": {"chrA": [0.0, 0.96, 0.0, 0.0], "chrB": [0.0, 88.12, 0.0]},
    "median_coverage": {"chrA": 0.0, "chrB": 0.0},
    "median_over_mean_coverage": {"chrA": 0.0, "chrB": 0.0},
    "ignored": {"nonsensical_records": 0, "pileup_too_large_positions": {"chrA": 0, "chrB": 1}},
    "coverage_distribution": {"values": hist(2049, b0=80, b1=14, b2=5, b3=1, b100=1), "range_start": 0, "range_stop": 2048},
    "genome_covered_by": {k: 0.9803922 for k in ("10x", "20x", "30x", "40x", "50x", "60x")},
}

# ---- Edits (edits.rs:217-353)
# read_one_edits: R1 -> [3];  read_two_edits: [0] = R2 + 2100 + 47 + N1 = 2149, [1] = R3 + 53 = 54  (D1 is a duplicate: skipped)
# VAF per position, (alts as f32 / total as f32 * 100.0) as usize:
#   chrA 1,2,3,5,6,7,8,15: no alts -> bin 0 (8 positions);  4: 2 alts of 3 -> 66.66667 -> 66;  16, 17: 1 of 1 -> 100
#        (9,10,11 under = / X and 12..14 under N hold neither refs nor alts: not visited)
#   chrB 10: 0 of 2100 -> 0;  12: 53 of 100 -> 0.53f32 * 100 = 52.999996 -> bin 52 (NOT 53)
#   chrN 5..8: bin 0 (4 positions)
edits = {
    "read_one_edits": {"values": hist(513, b3=1), "range_start": 0, "range_stop": 512},
    "read_two_edits": {"values": hist(513, b0=2149, b1=54), "range_start": 0, "range_stop": 512},
    "vaf_histogram": {"values": hist(101, b0=13, b52=1, b66=1, b100=2), "range_start": 0, "range_stop": 100},
    "summary": {"mean_edits_read_one": 3.0, "mean_edits_read_two": 0.02451202905129369},   # 54 / 2203
}

expected = {"general": general, "features": None, "gc_content": None, "template_length": None, "quality_scores": None,
            "coverage": coverage, "edits": edits}
doc = {
    "description": "hand-derived golden (Edits over every CIGAR operation, multi-sequence Coverage), see make_hand_goldens_edits.py",
    "config": {"ref_names": ["chrA", "chrB", "chrN", "chrE"], "ref_len": [60, 40, 30, 50], "ref_is_primary": [1, 1, 0, 1],
               "bin_size": 25, "max_read_len": 32, "facets": 0x31,
               "ref_bases": [("ACGT" * 15)[:n] for n in (60, 40, 30, 50)]},
    "records": records,
    "expected": expected,
}
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hand_edits_multiseq.json")
with open(out, "w") as f:
    json.dump(doc, f, separators=(",", ":"))
print(out)
