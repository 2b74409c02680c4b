"""Pins the oracle against every known-answer test the reference holds for the
hot path:

  src/utils/histogram.rs:405-523   (10 unit tests)  + doc examples :41-108, :241-248
  src/utils/alignment.rs:134-202   (5 unit tests)
  src/qc/record_based/gc_content.rs:145-151
  src/qc.rs:238-271                (facet names / default set, as constants of the ABI)

The values below are the reference tests' own inputs and expected outputs
(data, transcribed); no reference source is copied.
"""
import ctypes as C

import numpy as np
import pytest

from ngs_amd import ffi
from tests.util import BASE_CODES, parse_cigar


class H:
    def __init__(self, mod, capacity=None):
        self.lib = mod.load()
        self.h = mod.Hist()
        if capacity is None:
            assert self.lib.orc_hist_init_default(C.byref(self.h)) == 0
        else:
            assert self.lib.orc_hist_init(C.byref(self.h), capacity) == 0

    def inc(self, b, v=None):
        if v is None:
            return self.lib.orc_hist_increment(C.byref(self.h), b)
        return self.lib.orc_hist_increment_by(C.byref(self.h), b, v)

    def get(self, b):
        return self.lib.orc_hist_get(C.byref(self.h), b)

    def q(self, name):
        some, out = C.c_int(), C.c_double()
        rc = getattr(self.lib, "orc_hist_" + name)(C.byref(self.h), C.byref(some), C.byref(out))
        assert rc == 0
        return out.value if some.value else None

    def values(self):
        n = self.h.range_stop + 1
        return [self.h.values[i] for i in range(n)]


def test_initialization(oracle_mod):  # histogram.rs:406-411
    s = H(oracle_mod, 100)
    assert s.lib.orc_hist_range_len(C.byref(s.h)) == 101
    assert s.h.range_start == 0 and s.h.range_stop == 100


def test_valid_increments_and_mean_median(oracle_mod):  # histogram.rs:414-431
    s = H(oracle_mod, 100)
    assert s.inc(25) == 0 and s.inc(50) == 0 and s.inc(75, 3) == 0 and s.inc(100, 5) == 0
    assert (s.get(25), s.get(50), s.get(75), s.get(100)) == (1, 1, 3, 5)
    assert s.lib.orc_hist_mean(C.byref(s.h)) == 80.0
    assert s.q("first_quartile") == 75.0
    assert s.q("median") == 87.5
    assert s.q("third_quartile") == 100.0
    assert s.q("interquartile_range") == 25.0


def test_median_on_empty_histogram(oracle_mod):  # histogram.rs:434-437
    assert H(oracle_mod, 5000).q("median") is None


def test_median_extensively(oracle_mod):  # histogram.rs:440-463
    s = H(oracle_mod, 5000)
    for b, v in ((0, 2500), (10, 2500), (100, 2500), (5000, 5000)):
        assert s.inc(b, v) == 0
    assert s.q("median") == 100.0
    s.inc(200, 2500)
    assert s.q("median") == 150.0
    s.inc(200)
    assert s.q("median") == 200.0


def test_invalid_increments(oracle_mod):  # histogram.rs:466-469
    assert H(oracle_mod, 100).inc(101) == 1  # BinOutOfBoundsError


def test_default_is_512_zero_based(oracle_mod):  # histogram.rs:472-481
    d = H(oracle_mod)
    assert (d.h.range_start, d.h.range_stop) == (0, 512)
    assert d.lib.orc_hist_range_len(C.byref(d.h)) == 513


def test_values(oracle_mod):  # histogram.rs:484-490
    h = H(oracle_mod, 3)
    h.inc(1), h.inc(2), h.inc(3, 3)
    assert h.values() == [0, 1, 1, 3]


def test_values_normalized(oracle_mod):  # histogram.rs:493-499
    h = H(oracle_mod, 3)
    h.inc(1), h.inc(2), h.inc(3, 3)
    out = (C.c_double * 4)()
    h.lib.orc_hist_values_normalized(C.byref(h.h), out)
    assert list(out) == [0.0, 0.2, 0.2, 0.6]


def test_count_values_from_bottom(oracle_mod):  # histogram.rs:502-511
    h = H(oracle_mod, 3)
    h.inc(0, 5), h.inc(1, 3), h.inc(2, 6)
    f = h.lib.orc_hist_count_from_bottom_until
    assert [f(C.byref(h.h), b) for b in (0, 1, 2, 3)] == [5, 8, 14, 14]


def test_count_values_from_top(oracle_mod):  # histogram.rs:514-523
    h = H(oracle_mod, 3)
    h.inc(0, 5), h.inc(1, 3), h.inc(2, 6)
    f = h.lib.orc_hist_count_from_top_until
    assert [f(C.byref(h.h), b) for b in (3, 2, 1, 0)] == [0, 6, 9, 14]


def test_doc_examples(oracle_mod):  # histogram.rs:41-108, :241-248
    h = H(oracle_mod, 10)
    assert h.inc(0) == 0 and h.inc(1, 42) == 0
    assert (h.get(0), h.get(1)) == (1, 42)
    assert h.lib.orc_hist_range_len(C.byref(h.h)) == 11
    ir = h.lib.orc_hist_in_range
    assert ir(C.byref(h.h), 0) and ir(C.byref(h.h), 5) and ir(C.byref(h.h), 10) and not ir(C.byref(h.h), 11)
    assert h.inc(11) == 1
    g = H(oracle_mod, 3)
    g.inc(0, 2), g.inc(1), g.inc(2)
    assert g.values() == [2, 1, 1, 0]
    out = (C.c_double * 4)()
    g.lib.orc_hist_values_normalized(C.byref(g.h), out)
    assert list(out) == [0.5, 0.25, 0.25, 0.0]
    k = H(oracle_mod, 100)
    assert ir(C.byref(k.h), 0) and ir(C.byref(k.h), 100) and not ir(C.byref(k.h), 101)


def test_gc_default_histogram(oracle_mod):  # gc_content.rs:145-151
    o = oracle_mod.Oracle([1000], facets=ffi.FACET_GC_CONTENT)
    o.finalize()
    assert len(o.gc_content()["histogram"]) == 101
    assert ffi.GC_BINS == 101


def _edits(mod, ref, rec, cigar):
    lib = mod.load()
    r = np.array([BASE_CODES[c] for c in ref], dtype=np.uint8)
    q = np.array([BASE_CODES[c] for c in rec], dtype=np.uint8)
    c = np.array(parse_cigar(cigar), dtype=np.uint32)
    e = C.c_uint64()
    rc = lib.orc_stepthrough_edits(r.ctypes.data_as(ffi.u8p), len(r), q.ctypes.data_as(ffi.u8p), len(q),
                                   c.ctypes.data_as(ffi.u32p), len(c), C.byref(e))
    return rc, e.value, lib.orc_stepthrough_error_message(rc).decode()


def test_alignment_zero_edits(oracle_mod):  # alignment.rs:135-143
    assert _edits(oracle_mod, "ACTG", "ACTG", "4M")[:2] == (0, 0)


def test_alignment_one_edit(oracle_mod):  # alignment.rs:146-154
    assert _edits(oracle_mod, "AATG", "ACTG", "4M")[:2] == (0, 1)


def test_alignment_softclips(oracle_mod):  # alignment.rs:157-165
    assert _edits(oracle_mod, "ACTG", "ACTGACTG", "4M4S")[:2] == (0, 0)


def test_alignment_too_few_record_bases(oracle_mod):  # alignment.rs:168-184
    rc, _, msg = _edits(oracle_mod, "ACTG", "ACTGACTG", "4M5S")
    assert rc != 0
    assert msg == ("malformed record: record specifies that we should be able to consume a record base, "
                   "but no such base was found")


def test_alignment_too_few_reference_bases(oracle_mod):  # alignment.rs:187-202
    rc, _, msg = _edits(oracle_mod, "ACTG", "ACT", "3M2D")
    assert rc != 0
    assert msg == ("malformed record: record specifies that we should be able to consume a reference base, "
                   "but no such base was found")


def test_facet_names_and_default_set():  # qc.rs:238-271 + each facet's name()
    assert bin(ffi.FACETS_DEFAULT & ffi.FACETS_RECORD_BASED).count("1") == 4  # 4 record-based by default
    assert bin(ffi.FACETS_DEFAULT & ffi.FACETS_SEQUENCE_BASED).count("1") == 1  # + Coverage


def test_facet_names_from_library(lib):
    names = {b: lib.ngsq_facet_name(b).decode() for b in (1, 2, 4, 8, 16, 32, 64)}
    assert names == {1: "General", 2: "Template Length", 4: "GC Content", 8: "Quality Score",
                     16: "Coverage", 32: "Edits", 64: "Genomic Features"}   # qc.rs:138-139 names, features.rs:107
    assert lib.ngsq_facet_name(128) is None


def test_genome_table_known_answers():  # utils/genome/ncbi/grch38_no_alt.rs:287-354 (the reference's own tests)
    """The sequence table the CLI and tools/qc_sharded.py load (ngs_amd/data/GRCh38_no_alt_AnalysisSet.tsv) against
    every known-answer test the reference holds for this genome: 22 autosomes, 2 sex chromosomes, chrM and chrEBV
    present, 42 unlocalized and 127 unplaced sequences, no alternative contigs / decoys / others, and 193 sequences in
    the primary assembly (autosomes + sex + alt + unlocalized + unplaced: utils/genome.rs:59-83) -- the set Coverage
    supports (coverage.rs:133-138)."""
    import collections
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ngs_amd", "data",
                        "GRCh38_no_alt_AnalysisSet.tsv")
    groups = collections.Counter()
    names = set()
    for line in open(path):
        if line.startswith("#") or not line.strip():
            continue
        name, group = line.rstrip("\n").split("\t")
        assert name not in names
        names.add(name)
        groups[group] += 1
    assert groups["autosome"] == 22 and groups["sex"] == 2
    assert groups["mitochondrion"] == 1 and "chrM" in names
    assert groups["ebv"] == 1 and "chrEBV" in names
    assert groups["unlocalized"] == 42 and groups["unplaced"] == 127
    assert groups["alt"] == 0 and groups["decoy"] == 0 and groups["other"] == 0
    primary = sum(groups[g] for g in ("autosome", "sex", "alt", "unlocalized", "unplaced"))
    assert primary == 193
