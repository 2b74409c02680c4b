"""Oracle vs the hand-derived goldens in tests/golden/ (SURVEY.md 8c), plus the
coverage binning rule on a tiny sequence (L = 10, bin size 4)."""
import json
import os

import numpy as np

from ngs_amd import ffi
from tests.util import batch_from_records, json_equal

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hand_six_records.json")


def load_gold():
    with open(GOLD) as f:
        return json.load(f)


def test_oracle_matches_hand_golden(oracle_mod):
    g = load_gold()
    cfg = g["config"]
    hb = batch_from_records(g["records"])
    o = oracle_mod.Oracle(cfg["ref_len"], cfg["ref_is_primary"], facets=cfg["facets"],
                          bin_size=cfg["bin_size"], max_read_len=cfg["max_read_len"])
    o.process_batch(hb)
    o.finalize()
    got = o.results(cfg["ref_names"])
    json_equal(got, g["expected"])


BASE_CODE = {"A": 1, "C": 2, "G": 4, "T": 8, "N": 15}  # BAM 4-bit codes, one per byte (ngsq_config.ref_bases)


def load_gold_edits():
    with open(os.path.join(os.path.dirname(GOLD), "hand_edits_multiseq.json")) as f:
        g = json.load(f)
    g["config"]["ref_bases"] = [np.array([BASE_CODE[c] for c in s], dtype=np.uint8) for s in g["config"]["ref_bases"]]
    return g


def test_oracle_matches_hand_golden_edits_multiseq(oracle_mod):
    """Second hand-derived case: the Edits walk over M I D N S H P = X with a mismatch under `=`, the read-1 / read-2
    split, a VAF on the f32 truncation edge (53/100 -> bin 52), two covered sequences + a non-primary one + one
    without records, a position deeper than the coverage capacity (tests/golden/make_hand_goldens_edits.py)."""
    g = load_gold_edits()
    cfg = g["config"]
    for cut in (None, 1000):   # one batch, and the same records in two (state carried across batches)
        o = oracle_mod.Oracle(cfg["ref_len"], cfg["ref_is_primary"], facets=cfg["facets"], bin_size=cfg["bin_size"],
                              max_read_len=cfg["max_read_len"], ref_bases=cfg["ref_bases"])
        hb = batch_from_records(g["records"])
        if cut is None:
            o.process_batch(hb)
        else:
            o.process_batch(hb.slice(0, cut))
            o.process_batch(hb.slice(cut, hb.n))
        o.finalize()
        json_equal(o.results(cfg["ref_names"]), g["expected"])
    text = o.results_json(cfg["ref_names"])
    assert '"10x": 0.9803922,' in text and '"chrB": 88.12' not in text and "88.12" in text


def test_json_text_layout(oracle_mod):
    """serde_json pretty layout: two-space indent, `"k": v`, no trailing newline,
    shortest round-trip floats with a trailing .0 on integers (results.rs:55)."""
    g = load_gold()
    cfg = g["config"]
    o = oracle_mod.Oracle(cfg["ref_len"], cfg["ref_is_primary"], facets=cfg["facets"],
                          bin_size=cfg["bin_size"], max_read_len=cfg["max_read_len"])
    o.process_batch(batch_from_records(g["records"]))
    o.finalize()
    text = o.results_json(cfg["ref_names"])
    assert text.startswith('{\n  "general": {\n    "records": {\n      "total": 6,')
    assert not text.endswith("\n")
    assert '"duplication_pct": 16.666666666666664,' in text
    assert '"mate_reference_sequence_id_mismatch_hq_pct": 0.0\n' in text
    assert '"features": null,' in text and text.rstrip().endswith('"edits": null\n}')
    assert '"10x": 0.0,' in text


def test_empty_input_gives_null_summaries(oracle_mod):
    """total = 0: every percentage is 0/0 = NaN, which serde_json writes as null."""
    o = oracle_mod.Oracle([1000], facets=ffi.FACETS_DEFAULT, max_read_len=150)
    o.finalize()
    r = o.results(["chr1"])
    assert r["general"]["summary"] == {"duplication_pct": None, "mapped_pct": None,
                                       "mate_reference_sequence_id_mismatch_pct": None,
                                       "mate_reference_sequence_id_mismatch_hq_pct": None}
    assert r["gc_content"]["summary"]["gc_content_pct"] is None
    assert r["template_length"]["summary"]["template_length_unknown_pct"] is None
    assert r["quality_scores"] == {"scores": {}}
    assert r["coverage"]["mean_coverage"] == {}
    assert r["coverage"]["genome_covered_by"]["10x"] is None
    assert r["general"]["cigar"] == {"read_one_cigar_ops": {}, "read_two_cigar_ops": {}}


def test_coverage_binning_rule(oracle_mod):
    """L = 10, bin 4: bins = [0.0, mean(p1..4), mean(p5..8), mean(p9..10)]; the
    distribution counts L+1 = 11 positions (coverage.rs:206-230)."""
    recs = [dict(flag=0, ref_id=0, pos=1, cigar="4M", seq="ACGT", qual=[1] * 4),   # covers 2..5
            dict(flag=0, ref_id=0, pos=4, cigar="5M", seq="ACGTA", qual=[1] * 5),  # covers 5..9
            dict(flag=0, ref_id=0, pos=8, cigar="2M", seq="AC", qual=[1] * 2)]     # covers 9..10
    o = oracle_mod.Oracle([10], facets=ffi.FACET_COVERAGE, bin_size=4)
    o.process_batch(batch_from_records(recs))
    o.finalize()
    r = o.results(["s"])["coverage"]
    # depth: p1 0, p2 1, p3 1, p4 1 | p5 2, p6 1, p7 1, p8 1 | p9 2, p10 1
    assert r["mean_coverage_per_bin"]["s"] == [0.0, 3 / 4, 5 / 4, 3 / 2]
    dist = r["coverage_distribution"]["values"]
    assert sum(dist) == 11 and dist[0] == 2 and dist[1] == 7 and dist[2] == 2
    assert r["mean_coverage"]["s"] == 11 / 11
    assert r["median_coverage"]["s"] == 1.0


def test_f32_genome_covered_by(oracle_mod):
    """coverage.rs:282 is f32 arithmetic: 3/7 -> 42.857143, 1/3 -> 33.333336."""
    # L = 6 -> 7 positions; 3 of them at depth 10
    recs = [dict(flag=0, ref_id=0, pos=0, cigar="3M", seq="ACG", qual=[1] * 3)] * 10
    o = oracle_mod.Oracle([6], facets=ffi.FACET_COVERAGE, bin_size=50000)
    o.process_batch(batch_from_records(recs))
    o.finalize()
    text = o.results_json(["s"])
    assert '"10x": 42.857143,' in text
    assert '"20x": 0.0,' in text
    recs = [dict(flag=0, ref_id=0, pos=0, cigar="1M", seq="A", qual=[1])] * 10
    o = oracle_mod.Oracle([2], facets=ffi.FACET_COVERAGE)
    o.process_batch(batch_from_records(recs))
    o.finalize()
    assert '"10x": 33.333336,' in o.results_json(["s"])


def test_vaf_f32_truncation(oracle_mod):
    """edits.rs:331-335: (alts as f32 / total as f32 * 100.0) as usize; 53/100 -> bin 52,
    59/100 -> 58."""
    for alts, want in ((53, 52), (59, 58), (50, 50), (100, 100), (0, 0)):
        ref = np.array([1], dtype=np.uint8)  # 'A'
        recs = [dict(flag=0, ref_id=0, pos=0, cigar="1M", seq="C", qual=[1])] * alts + \
               [dict(flag=0, ref_id=0, pos=0, cigar="1M", seq="A", qual=[1])] * (100 - alts)
        o = oracle_mod.Oracle([1], facets=ffi.FACET_EDITS, ref_bases=[ref])
        o.process_batch(batch_from_records(recs))
        o.finalize()
        vaf = o.edits()[2]
        assert vaf[want] == 1 and vaf.sum() == 1, (alts, np.nonzero(vaf))


def test_genomic_features_hand_golden(oracle_mod):
    """Hand-derived from features.rs:115-262 (incl. its two boundary quirks: a feature's last base is
    outside its rust_lapper interval, and the lookup reaches one base past the read's end)."""
    NAMES = dict(five_prime_UTR=0, three_prime_UTR=1, CDS=2, exon=3, gene=4)
    model = [("gene", 1000, 5000), ("exon", 1000, 1200), ("exon", 3000, 3300), ("five_prime_UTR", 1000, 1050),
             ("CDS", 1051, 1200), ("CDS", 3000, 3200), ("three_prime_UTR", 3201, 3300),
             ("gene", 10, 500)]  # the last one sits on the non-primary sequence below
    ref = [0] * 7 + [1]
    rec = lambda **kw: dict(dict(flag=0, mapq=60, ref_id=0, cigar="50M", seq="A" * 50, qual=[30] * 50), **kw)  # noqa: E731
    records = [
        rec(pos=999),                                       # A [1000,1051): gene+exon, 5'UTR (CDS starts at 1051: no)
        rec(pos=1999, cigar="100M", seq="A" * 100, qual=[30] * 100),   # B intronic
        rec(pos=5999, cigar="100M", seq="A" * 100, qual=[30] * 100),   # C intergenic
        rec(pos=0, flag=4),                                 # D unmapped
        rec(pos=99, ref_id=1),                              # E non-primary sequence
        rec(pos=3149, cigar="100M", seq="A" * 100, qual=[30] * 100),   # F CDS + 3'UTR, exonic
        rec(pos=1199, cigar="1M", seq="A", qual=[30]),      # G exon/CDS end at 1200 = exclusive: intronic
        rec(pos=949),                                       # H [950,1001) touches base 1000: gene, exon, 5'UTR
    ]
    from tests.util import batch_from_records
    orc = oracle_mod.Oracle([10_000, 1_000], [1, 0], facets=ffi.FACET_FEATURES)
    orc.set_features(ref, [NAMES[n] for n, _, _ in model], [s for _, s, _ in model], [e for _, _, e in model])
    orc.process_batch(batch_from_records(records))
    orc.finalize()
    got = orc.results(["chr1", "chrUn"])["features"]
    assert got == {
        "exonic_translation_regions": {"utr_five_prime_count": 2, "utr_three_prime_count": 1, "coding_sequence_count": 1},
        "gene_regions": {"intergenic_count": 1, "exonic_count": 3, "intronic_count": 2},
        "records": {"processed": 6, "ignored_flags": 1, "ignored_nonprimary_chromosome": 1},
        "summary": {"ignored_flags_pct": 12.5, "ignored_nonprimary_chromosome_pct": 12.5},
    }
    assert orc.results(["chr1", "chrUn"])["general"] is None
