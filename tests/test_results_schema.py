"""The `Results` document against the reference's own struct definitions: tests/golden/results_schema.json is
derived by tests/golden/make_results_schema.py from src/qc/results.rs:23-45 and every struct below it (serde, no
rename/skip attributes: field names in declaration order).  Both the oracle's document (CPU) and the text
ngsq_results_json emits on the GPU path must deserialize into exactly that shape -- what `Results::read`
(results.rs:63-67) and `ngs plot` rely on."""
import json
import os

import numpy as np
import pytest

from ngs_amd import ffi, host
from tests.util import random_batch, random_ref_bases, make_edit_friendly

SCHEMA = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "results_schema.json")))["schema"]
INT_KINDS = {"usize", "u64", "u32", "i32", "i64"}


def conforms(node, value, path="results"):
    kind = node["kind"]
    if kind == "option":
        if value is not None:
            conforms(node["of"], value, path)
    elif kind == "struct":
        assert isinstance(value, dict), f"{path}: {node['name']} must be an object"
        want = [f["name"] for f in node["fields"]]
        assert list(value.keys()) == want, f"{path}: fields {list(value.keys())} != {want} (names and declaration order)"
        for f in node["fields"]:
            conforms(f["schema"], value[f["name"]], f"{path}.{f['name']}")
    elif kind == "map":
        assert isinstance(value, dict), f"{path}: a map"
        for k, v in value.items():
            assert isinstance(k, str) and (node["key"] == "String" or k.isdigit()), f"{path}: key {k!r}"   # serde_json: integer keys as strings
            conforms(node["of"], v, f"{path}[{k}]")
    elif kind == "vec":
        assert isinstance(value, list), f"{path}: a list"
        for i, v in enumerate(value):
            conforms(node["of"], v, f"{path}[{i}]")
    elif kind in INT_KINDS:
        assert isinstance(value, int) and not isinstance(value, bool) and (value >= 0 or kind[0] == "i"), f"{path}: {value!r} is not {kind}"
    elif kind in ("f64", "f32"):
        assert value is None or isinstance(value, float), f"{path}: {value!r} is not a float (NaN / inf serialize as null)"
    else:
        raise AssertionError(f"{path}: unhandled kind {kind}")


def test_schema_fixture_is_the_documented_document():
    """SURVEY appendix A / results.rs:23-45: the seven facets in this order, and the leaf structs the plots read."""
    top = [f["name"] for f in SCHEMA["fields"]]
    assert top == ["general", "features", "gc_content", "template_length", "quality_scores", "coverage", "edits"]
    gc = SCHEMA["fields"][2]["schema"]["of"]
    assert [f["name"] for f in gc["fields"]] == ["histogram", "nucleobases", "records", "summary"]
    hist = gc["fields"][0]["schema"]
    assert [f["name"] for f in hist["fields"]] == ["values", "range_start", "range_stop"]


def workload(seed=3):
    rng = np.random.default_rng(seed)
    ref_len = [6000, 900]
    bases = random_ref_bases(rng, ref_len)
    hb = make_edit_friendly(random_batch(rng, 3000, ref_len, max_len=120, weird=False), rng, bases, ref_len)
    feats = (np.array([0, 0, 1], np.uint32), np.array([0, 1, 2], np.uint32), np.array([10, 500, 5], np.uint32),
             np.array([400, 900, 700], np.uint32), (0, 1, 2, 3, 4))
    return ref_len, bases, hb, feats


@pytest.mark.parametrize("facets", [ffi.FACETS_DEFAULT, ffi.FACETS_DEFAULT | ffi.FACET_EDITS | ffi.FACET_FEATURES, ffi.FACET_GC_CONTENT, 0])
def test_oracle_document_has_the_reference_shape(oracle_mod, facets):
    ref_len, bases, hb, feats = workload()
    orc = oracle_mod.Oracle(ref_len, facets=facets, bin_size=1000, max_read_len=128,
                            ref_bases=bases if facets & ffi.FACET_EDITS else None)
    if facets & ffi.FACET_FEATURES:
        orc.set_features(*feats)
    orc.process_batch(hb)
    orc.finalize(allow_malformed=True)
    doc = orc.results(["chr1", "chr2"])
    conforms(SCHEMA, doc)
    on = {"general": ffi.FACET_GENERAL, "features": ffi.FACET_FEATURES, "gc_content": ffi.FACET_GC_CONTENT,
          "template_length": ffi.FACET_TEMPLATE_LENGTH, "quality_scores": ffi.FACET_QUALITY_SCORE,
          "coverage": ffi.FACET_COVERAGE, "edits": ffi.FACET_EDITS}
    for key, bit in on.items():
        assert (doc[key] is not None) == bool(facets & bit), key       # absent facet => null (results.rs)


@pytest.mark.gpu
@pytest.mark.parametrize("sorted_input", [False, True])
def test_gpu_document_has_the_reference_shape(gpu_lib, sorted_input):
    from tests.util import coordinate_sorted
    ref_len, bases, hb, feats = workload()
    hb = coordinate_sorted(hb)
    facets = ffi.FACETS_DEFAULT | ffi.FACET_EDITS | ffi.FACET_FEATURES
    with host.QcContext(ref_len, facets=facets, bin_size=1000, max_read_len=128, ref_bases=bases, lib=gpu_lib,
                        sorted_input=sorted_input) as gpu:
        gpu.set_features(*feats)
        gpu.process_batch(hb)
        gpu.finalize(allow_malformed=True)
        conforms(SCHEMA, gpu.results(["chr1", "chr2"]))
    with host.QcContext(ref_len, facets=ffi.FACET_QUALITY_SCORE, max_read_len=128, lib=gpu_lib) as gpu:   # no records at all
        gpu.finalize()
        doc = gpu.results(["chr1", "chr2"])
        conforms(SCHEMA, doc)
        assert doc["quality_scores"] == {"scores": {}} and doc["general"] is None
