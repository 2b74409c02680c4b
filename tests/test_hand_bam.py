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


def test_long_cigar_placeholder_is_refused_by_name(lib):
    """[N10] (oracle/oracle.h): the real CIGAR of the record is in its CG tag; scanning the placeholder would be a wrong answer."""
    h = C.c_void_p()
    assert lib.ngsq_bam_open(LONG.encode(), 1, C.byref(h)) == 0
    b = ffi.Batch()
    assert lib.ngsq_bam_next_batch(h, 10, C.byref(b)) == ffi.ERR_UNSUPPORTED
    assert b"CIGAR of more than 65535 operations" in lib.ngsq_bam_last_error()
    lib.ngsq_bam_close(h)


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
        with pytest.raises(RuntimeError, match="CIGAR of more than 65535 operations"):
            read_all_device(gpu_lib, ctx, LONG, 10)


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
