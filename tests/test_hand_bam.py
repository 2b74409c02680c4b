"""BAM bytes assembled from the SAM/BAM specification by tests/golden/make_hand_bam.py -- not by this repository's writers
(until now every BAM a test read was written by the code under test's siblings).  The committed files
exercise: aux tags of every type behind the qualities, absent qualities, every CIGAR operation, a BGZF member with two
dynamic DEFLATE blocks and an (empty) stored block, a header in a stored block, an empty member in the middle of the
file, fixed Huffman codes, records straddling members, an unplaced record -- and the long-CIGAR convention.

CPU: the host reader against the table the generator wrote the records from, and the oracle on those records against
numbers worked out by hand in the generator.  GPU: the device reader (inflate + parse kernels) byte for byte against the
host reader, and the command line against the oracle."""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

from ngs_amd import ffi, host
from tests.test_bam_ingest import read_all, records_of

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
BAM = os.path.join(GOLD, "hand_spec.bam")
LONG = os.path.join(GOLD, "hand_longcigar.bam")
OPS = "MIDNSHP=X"


def expected():
    with open(os.path.join(GOLD, "hand_spec_expected.json")) as f:
        return json.load(f)


def test_the_committed_files_are_what_the_generator_writes(tmp_path):
    """(the fixtures are data; the script that made them is committed next to them and reproduces them bit for bit)"""
    import shutil
    import sys
    d = tmp_path / "golden"
    d.mkdir()
    shutil.copy(os.path.join(GOLD, "make_hand_bam.py"), d)
    subprocess.run([sys.executable, str(d / "make_hand_bam.py")], check=True, capture_output=True)
    for name in ("hand_spec.bam", "hand_spec.bam.bai", "hand_longcigar.bam", "hand_spec_expected.json"):
        assert open(d / name, "rb").read() == open(os.path.join(GOLD, name), "rb").read(), name


@pytest.mark.parametrize("max_records", [1 << 20, 3, 1])
def test_host_reader_on_hand_assembled_bytes(lib, max_records):
    exp = expected()
    refs, batches, n = read_all(lib, BAM, max_records)
    assert refs == [tuple(r) for r in exp["references"]] and n == len(exp["records"])
    got = [r for b in batches for r in records_of(b)]
    ids = np.concatenate([b.cols["record_id"] for b in batches])
    assert [int(v) for v in ids] == exp["virtual_offsets"]
    for (fixed, seq, qual, cigar), want in zip(got, exp["records"]):
        assert fixed == (want["flag"], want["mapq"], want["ref_id"], want["pos"], want["mate_ref_id"], want["tlen"], want["l_seq"],
                         len(want["cigar"]))
        codes = [(seq[i // 2] >> (4 if i % 2 == 0 else 0)) & 15 for i in range(want["l_seq"])]
        assert codes == want["seq_codes"]
        assert list(qual) == (want["qual"] if want["qual"] is not None else [])     # 0xFF-filled: no scores
        assert list(cigar) == want["cigar"]


def test_oracle_on_the_hand_assembled_records(lib, oracle_mod):
    """The numbers in hand_spec_expected.json were worked out by hand from the record table (general.rs:31-124,
    template_length.rs:79-87, quality_scores.rs:37-49, coverage.rs:148-180): the oracle, fed the host reader's batches, must
    arrive at them."""
    exp = expected()
    _, batches, _ = read_all(lib, BAM, 5)
    lens = [r[1] for r in exp["references"]]
    o = oracle_mod.Oracle(lens, facets=ffi.FACETS_DEFAULT, bin_size=1, max_read_len=64)
    for b in batches:
        o.process_batch(b)
    o.finalize()
    doc = o.results(["chr1", "chr2"])
    g = exp["general"]
    rec = doc["general"]["records"]
    for k in ("total", "unmapped", "duplicate", "primary_mapped", "primary_duplicate", "paired", "read_1", "read_2", "proper_pair",
              "singleton", "mate_mapped", "mate_reference_sequence_id_mismatch", "mate_reference_sequence_id_mismatch_hq"):
        assert rec[k] == g[k], k
    assert rec["designation"] == {"primary": g["primary"], "secondary": g["secondary"], "supplementary": g["supplementary"]}
    assert doc["general"]["cigar"]["read_one_cigar_ops"] == g["read_one_cigar_ops"]
    assert doc["general"]["cigar"]["read_two_cigar_ops"] == g["read_two_cigar_ops"]
    t = doc["template_length"]
    assert t["records"]["ignored"] == exp["template_length"]["ignored"]
    assert t["histogram"]["values"][251] == exp["template_length"]["251"] and t["histogram"]["values"][0] == exp["template_length"]["0"]
    scores = doc["quality_scores"]["scores"]
    for cycle, count in exp["quality_rows"].items():
        assert (sum(scores[cycle]["values"]) if cycle in scores else 0) == count, cycle
    assert scores["1"]["values"][93] == 1 and scores["1"]["values"][0] == 1      # r4's 93s, r5's zeros
    depth = doc["coverage"]["mean_coverage_per_bin"]["chr1"]                    # bins of one position: the depth itself
    for p, dep in exp["depth_chr1"].items():
        assert depth[int(p)] == float(dep), p
    assert doc["coverage"]["mean_coverage_per_bin"]["chr2"][11] == 1.0 and doc["coverage"]["mean_coverage_per_bin"]["chr2"][131] == 0.0


# hand_longcigar.bam (make_hand_bam.py): ONE record, flag 0, chr1 pos 1000 (0-based), 30 bases "ACGTAC" x 5, CIGAR field
# "30S35N" = the long-CIGAR placeholder, real operations 10M5D20M in its CG:B,I tag (SAM specification 4.2.2; three operations
# stand in for "more than 65535").  What the facets must see, worked by hand from the reference's source:
#   General (general.rs:103-121): not first segment -> read_two_cigar_ops {M: 2, D: 1}  (the placeholder would say {S: 1, N: 1})
#   Coverage (coverage.rs:159-180): span 10 + 5 + 20 = 35 -> positions 1001..1035 (1-based, as the per-position vectors are indexed) at depth 1, 1000 and 1036 at 0
#   Edits (edits.rs:276-291) against a reference of all A: read bases 0..9 under the first M, 10..29 under the second;
#         "ACGTAC" x 5 has A at indices 0, 4 (mod 6): of the 30 compared bases 10 are A -> 20 edits; the five deleted
#         positions 1011..1015 are compared with nothing (refs + alts = 0 there)
LONG_EXPECT = {"read_two": {"M": 2, "D": 1}, "depth": {1000: 0, 1001: 1, 1010: 1, 1013: 1, 1035: 1, 1036: 0}, "edits": 20}


def _check_long_cigar(doc):
    assert doc["general"]["cigar"]["read_two_cigar_ops"] == LONG_EXPECT["read_two"] and doc["general"]["cigar"]["read_one_cigar_ops"] == {}
    depth = doc["coverage"]["mean_coverage_per_bin"]["chr1"]   # (bins of one position)
    for p, dep in LONG_EXPECT["depth"].items():
        assert depth[p] == float(dep), p
    if doc.get("edits"):
        assert doc["edits"]["read_two_edits"]["values"][LONG_EXPECT["edits"]] == 1 and sum(doc["edits"]["read_two_edits"]["values"]) == 1
        assert sum(doc["edits"]["vaf_histogram"]["values"]) == 30     # thirty compared positions, none of the deleted ones


def _long_cigar_context(lib_or_oracle, make):
    ref_len = [100_000, 5_000]
    bases = [np.full(L, 1, dtype=np.uint8) for L in ref_len]   # all A
    return make(ref_len, facets=ffi.FACETS_DEFAULT | ffi.FACET_EDITS, bin_size=1, max_read_len=64, ref_bases=bases), ref_len


def test_long_cigar_is_resolved_from_the_cg_tag(lib, oracle_mod):
    """[N10] (oracle/oracle.h): noodles hands the facets the operations of the CG tag, not the placeholder.  Host reader ->
    oracle against numbers worked by hand (above)."""
    _, batches, n = read_all(lib, LONG, 10)
    assert n == 1
    hb = batches[0]
    assert int(hb.cols["n_cigar"][0]) == 3 and hb.cols["cigar_off"] is not None
    assert [int(x) for x in hb.cols["cigar"][:3]] == [10 << 4 | 0, 5 << 4 | 2, 20 << 4 | 0]
    orc, _ = _long_cigar_context(None, lambda rl, **kw: oracle_mod.Oracle(rl, **kw))
    orc.process_batch(hb)
    orc.finalize()
    _check_long_cigar(orc.results(["chr1", "chr2"]))


@pytest.mark.gpu
def test_long_cigar_on_the_device(gpu_lib, oracle_mod, tmp_path):
    """The device reader finds the tag too (k_rec_fixed), the kernels take the count from the offsets, and the facets arrive at
    the hand-worked numbers."""
    from tests.test_device_ingest_gpu import read_all_device, same_batches
    with host.QcContext([100_000, 5_000], lib=gpu_lib) as ctx:
        got, n = read_all_device(gpu_lib, ctx, LONG, 10)
        _, want, _ = read_all(gpu_lib, LONG, 10)
        assert n == 1 and read_all_device.last_stats["long_cigar_records"] == 1
        same_batches(got, want)
    _, batches, _ = read_all(gpu_lib, LONG, 10)
    gpu, _ = _long_cigar_context(None, lambda rl, **kw: host.QcContext(rl, lib=gpu_lib, **kw))
    gpu.process_batch(batches[0])
    gpu.finalize()
    _check_long_cigar(gpu.results(["chr1", "chr2"]))
    gpu.close()


def _seventy_thousand_ops(lib, tmp_path):
    """More operations than the 16-bit field counts, for real: 35 000 x (1M 1I) between ordinary records.  The test writer
    stores it as the specification says (placeholder + CG tag)."""
    from tests import bamio
    from tests.util import batch_from_records
    big = "1M1I" * 35_000
    recs = [dict(flag=0x41 if k % 2 else 0, mapq=30, ref_id=0, pos=500 + 7 * k, mate_ref_id=0, tlen=0, cigar=big if k == 3 else "40M",
                 seq=("ACGT" * 17_500) if k == 3 else "ACGT" * 10, qual=[20] * (70_000 if k == 3 else 40)) for k in range(8)]
    p = str(tmp_path / "big.bam")
    bamio.write_bam(p, batch_from_records(recs), ["chr1"], [200_000])
    _, hbatches, n = read_all(lib, p, 100)
    assert n == 8 and int(hbatches[0].cols["n_cigar"][3]) == 65535                  # saturated (include/ngsq.h, ABI 5)
    off = hbatches[0].cols["cigar_off"]
    assert int(off[4] - off[3]) == 70_000 and int(off[3] - off[2]) == 1            # the offsets say how many
    assert np.array_equal(hbatches[0].cols["cigar"][int(off[3]):int(off[4])], np.tile(np.array([16, 17], dtype=np.uint32), 35_000))
    return p, hbatches


def test_a_cigar_of_seventy_thousand_operations_host_reader(lib, oracle_mod, tmp_path):
    _, hbatches = _seventy_thousand_ops(lib, tmp_path)
    o = oracle_mod.Oracle([200_000], facets=ffi.FACETS_DEFAULT, bin_size=1000, max_read_len=128)
    o.process_batch(hbatches[0])
    o.finalize(allow_malformed=True)
    g = o.general()
    assert g["read_one_cigar_ops"][0] + g["read_two_cigar_ops"][0] == 35_000 + 7 and g["read_one_cigar_ops"][1] + g["read_two_cigar_ops"][1] == 35_000


@pytest.mark.gpu
def test_a_cigar_of_seventy_thousand_operations(gpu_lib, oracle_mod, tmp_path):
    """Both readers resolve it; n_cigar says 65535 and the offsets 70 000; every facet kernel agrees with the oracle."""
    from tests.test_device_ingest_gpu import read_all_device, same_batches
    from tests.util import compare_contexts, json_equal
    p, hbatches = _seventy_thousand_ops(gpu_lib, tmp_path)
    with host.QcContext([200_000], lib=gpu_lib) as ctx:
        dbatches, dn = read_all_device(gpu_lib, ctx, p, 100)
        assert dn == 8 and read_all_device.last_stats["long_cigar_records"] == 1
        same_batches(dbatches, hbatches)
    ref_len = [200_000]
    bases = [np.tile(np.array([1, 2, 4, 8], dtype=np.uint8), 50_000)]
    kw = dict(facets=ffi.FACETS_DEFAULT | ffi.FACET_EDITS, bin_size=1000, max_read_len=128, ref_bases=bases, gc_seed=3)
    orc = oracle_mod.Oracle(ref_len, **kw)
    gpu = host.QcContext(ref_len, lib=gpu_lib, **kw)
    for c in (orc, gpu):
        c.process_batch(hbatches[0])
    assert orc.finalize(allow_malformed=True) == gpu.finalize(allow_malformed=True)
    compare_contexts(gpu, orc, 1, kw["facets"], kw["bin_size"], ref_len)
    json_equal(gpu.results(["chr1"]), orc.results(["chr1"]))
    gpu.close()


@pytest.mark.gpu
@pytest.mark.parametrize("raw_mb", [None, "1"])
def test_device_reader_on_hand_assembled_bytes(gpu_lib, monkeypatch, raw_mb):
    from tests.test_device_ingest_gpu import read_all_device, same_batches
    if raw_mb:
        monkeypatch.setenv("NGSQ_INGEST_RAW_MB", raw_mb)
    with host.QcContext([100000, 5000], lib=gpu_lib) as ctx:
        for max_records in (1 << 20, 3):
            _, want, n = read_all(gpu_lib, BAM, max_records)
            got, n2 = read_all_device(gpu_lib, ctx, BAM, max_records)
            assert n2 == n
            same_batches(got, want)
        _, want, _ = read_all(gpu_lib, LONG, 10)      # the long-CIGAR convention, resolved by both (ABI 5)
        got, n2 = read_all_device(gpu_lib, ctx, LONG, 10)
        assert n2 == 1
        same_batches(got, want)


@pytest.mark.gpu
def test_cli_on_hand_assembled_bytes(gpu_lib, oracle_mod, tmp_path):
    from ngs_amd import build
    from tests.util import json_equal
    exp = expected()
    _, batches, _ = read_all(gpu_lib, BAM, 1 << 20)
    o = oracle_mod.Oracle([r[1] for r in exp["references"]], facets=ffi.FACETS_DEFAULT, max_read_len=256, gc_seed=0x4E4753)
    for b in batches:
        o.process_batch(b)
    o.finalize()
    want = o.results(["chr1", "chr2"])
    ngs = build.build_cli(verbose=False)
    for extra in (["--ingest", "device"], ["--ingest", "host"], ["--gpus", "2", "--same-device"]):
        out = tmp_path / "_".join(extra).replace("-", "")
        r = subprocess.run([ngs, "-q", "qc", BAM, "GRCh38_no_alt_AnalysisSet", "-o", str(out)] + extra, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        json_equal(json.load(open(out / "hand_spec.bam.results.json")), want)


def _placeholder_without_tag(tmp_path):
    """<l_seq>S<n>N as a record's real CIGAR, no CG tag: odd, but an alignment like any other."""
    from tests import bamio
    from tests.util import batch_from_records
    recs = [dict(flag=0, mapq=30, ref_id=0, pos=100 + 10 * k, mate_ref_id=-1, tlen=0, cigar="30S35N" if k == 3 else "30M",
                 seq="ACGTAC" * 5, qual=[25] * 30) for k in range(8)]
    hb = batch_from_records(recs)
    p = str(tmp_path / "ph.bam")
    bamio.write_bam(p, hb, ["chr1"], [100000])
    return p


def test_the_placeholder_alone_is_an_alignment(lib, tmp_path):
    _, batches, n = read_all(lib, _placeholder_without_tag(tmp_path), 100)
    assert n == 8 and [int(x) for x in batches[0].cols["n_cigar"]] == [1, 1, 1, 2, 1, 1, 1, 1]


@pytest.mark.gpu
def test_the_placeholder_alone_is_an_alignment_on_the_device(gpu_lib, tmp_path):
    from tests.test_device_ingest_gpu import read_all_device, same_batches
    p = _placeholder_without_tag(tmp_path)
    with host.QcContext([100000], lib=gpu_lib) as ctx:
        got, n = read_all_device(gpu_lib, ctx, p, 100)
        _, want, _ = read_all(gpu_lib, p, 100)
        assert n == 8
        same_batches(got, want)
