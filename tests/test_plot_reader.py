"""SURVEY.md 8(f) rank 4, second half: a results file of this build must be readable by `ngs plot sample`
(src/plot/sample.rs:41-94, Results::read at src/qc/results.rs:63-67).  oracle/plot_reader.py restates what the three
sample plots read and compute; here (CPU) it is pinned on a hand-computed document and on the reference's messages,
and (GPU) the text ngsq_results_json emits goes through it and must give the numbers the oracle's document gives."""
import json
import os

import numpy as np
import pytest

from ngs_amd import ffi, host
from tests.test_results_schema import SCHEMA, conforms, workload


@pytest.fixture(scope="module")
def reader(oracle_mod):
    from oracle import plot_reader
    return plot_reader


def hist(values):
    return {"values": list(values), "range_start": 0, "range_stop": len(values) - 1}


def test_hand_computed_document(reader):
    """Numbers worked out by hand from utils/histogram.rs:272-352 and the plot sources."""
    gc = [0] * 101
    gc[40], gc[41], gc[60] = 1, 2, 1
    q1 = [0] * 94
    q1[10], q1[20], q1[30] = 1, 2, 1          # n = 4: p25 -> needed 1.0, collected == needed at bin 10 -> midpoint(10, 20) = 15
    q2 = [0] * 94
    q2[37] = 5                                # every percentile 37
    vaf = [0] * 101
    vaf[0], vaf[50] = 3, 1
    doc = {"general": None, "features": None,
           "gc_content": {"histogram": hist(gc), "nucleobases": {}, "records": {}, "summary": {}},
           "template_length": None, "quality_scores": {"scores": {"2": hist(q2), "1": hist(q1)}}, "coverage": None,
           "edits": {"read_one_edits": hist([0] * 513), "read_two_edits": hist([0] * 513), "vaf_histogram": hist(vaf), "summary": {}}}
    out = reader.plot_sample(doc, "/data/SAMPLE.bam.results.json")
    assert list(out) == ["quality-score-distribution.sample.html", "gc-content-distribution.sample.html", "vaf-distribution.sample.html"]
    g = out["gc-content-distribution.sample.html"]
    assert g["name"] == "SAMPLE.bam" and g["x"] == list(range(101))
    assert (g["y"][40], g["y"][41], g["y"][60], sum(g["y"])) == (0.25, 0.5, 0.25, 1.0)
    q = out["quality-score-distribution.sample.html"]
    assert q["x"] == [1, 2]                                   # integer keys, sorted numerically ("10" would come after "2")
    # position 1: median: needed 2.0, bins 10 -> 1, 20 -> 3 > 2 => 20; Q3: needed 3.0 == collected at 20 => midpoint(20, 30) = 25
    assert q["y"] == [20.0, 37.0] and q["error_plus"] == [5.0, 0.0] and q["error_minus"] == [5.0, 0.0] and q["y_lim"] == 37.0
    v = out["vaf-distribution.sample.html"]
    assert (v["y"][0], v["y"][50], len(v["y"])) == (0.75, 0.25, 101)
    assert list(reader.plot_sample(doc, "x.results.json", only="gc content distribution")) == ["gc-content-distribution.sample.html"]


def test_reference_messages(reader):
    empty = {"gc_content": {"histogram": hist([0] * 101)}, "quality_scores": None, "edits": None}
    with pytest.raises(reader.PlotError, match="has GC content information, but it's empty!"):
        reader.gc_content_distribution(empty, "f.results.json")
    with pytest.raises(reader.PlotError, match="File f.results.json has no quality score information!"):
        reader.quality_score_distribution(empty, "f.results.json")
    with pytest.raises(reader.PlotError, match="File f.results.json has no Edits information!"):
        reader.vaf_distribution(empty, "f.results.json")
    with pytest.raises(reader.PlotError, match="has no GC content information!"):
        reader.gc_content_distribution({"gc_content": None}, "f.results.json")
    with pytest.raises(reader.PlotError, match="No plots matched the specified `--only` flag: nope"):
        reader.select_plots("nope")


def test_oracle_document_is_plottable(oracle_mod, reader):
    ref_len, bases, hb, feats = workload()
    orc = oracle_mod.Oracle(ref_len, facets=ffi.FACETS_DEFAULT | ffi.FACET_EDITS, bin_size=1000, max_read_len=128, ref_bases=bases)
    orc.process_batch(hb)
    orc.finalize(allow_malformed=True)
    doc = json.loads(orc.results_json(["chr1", "chr2"]))
    conforms(SCHEMA, doc)
    out = reader.plot_sample(doc, "s.bam.results.json")
    q = out["quality-score-distribution.sample.html"]
    scores = orc.quality_scores()
    touched = [i + 1 for i in range(scores.shape[0]) if scores[i].sum()]
    assert q["x"] == touched and len(q["y"]) == len(touched)
    assert abs(sum(out["gc-content-distribution.sample.html"]["y"]) - 1.0) < 1e-12
    assert abs(sum(out["vaf-distribution.sample.html"]["y"]) - 1.0) < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("sorted_input", [False, True])
def test_gpu_results_file_is_plottable(gpu_lib, oracle_mod, reader, tmp_path, sorted_input):
    from tests.util import coordinate_sorted
    ref_len, bases, hb, feats = workload()
    hb = coordinate_sorted(hb)
    facets = ffi.FACETS_DEFAULT | ffi.FACET_EDITS
    with host.QcContext(ref_len, facets=facets, bin_size=1000, max_read_len=128, ref_bases=bases, lib=gpu_lib,
                        sorted_input=sorted_input) as gpu:
        gpu.process_batch(hb)
        gpu.finalize(allow_malformed=True)
        text = gpu.results_json(["chr1", "chr2"])
    path = str(tmp_path / "SAMPLE.bam.results.json")          # the file `ngs qc` writes (results.rs:50-60)
    with open(path, "w") as f:
        f.write(text)
    doc = json.load(open(path))
    conforms(SCHEMA, doc)                                       # Results::read
    got = reader.plot_sample(doc, path)
    orc = oracle_mod.Oracle(ref_len, facets=facets, bin_size=1000, max_read_len=128, ref_bases=bases)
    orc.process_batch(hb)
    orc.finalize(allow_malformed=True)
    want = reader.plot_sample(json.loads(orc.results_json(["chr1", "chr2"])), path)
    assert got == want
    assert got["gc-content-distribution.sample.html"]["name"] == "SAMPLE.bam"
    # a run without Edits cannot draw the VAF plot, and says so like the reference
    with host.QcContext(ref_len, facets=ffi.FACETS_DEFAULT, bin_size=1000, max_read_len=128, lib=gpu_lib) as gpu:
        gpu.process_batch(hb)
        gpu.finalize(allow_malformed=True)
        doc2 = json.loads(gpu.results_json(["chr1", "chr2"]))
    with pytest.raises(reader.PlotError, match="has no Edits information!"):
        reader.plot_sample(doc2, path)
