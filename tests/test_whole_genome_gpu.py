"""All seven facets on the header `ngs qc` meets in practice: the 195 @SQ lines of the GRCh38 no-alt analysis set
(193 primary; chrM and chrEBV are not), records on the chromosomes, chrM, chrEBV and the unplaced contigs, reads that
straddle sequence ends, empty sequences between covered ones.  The reference requires every @SQ to be in the named genome
and loops over all of them (src/qc/command.rs:258-272,356; src/utils/genome/ncbi/grch38_no_alt.rs:17-285, test :308-311);
until round 6 no test here had more than four sequences.

Every path against the oracle's document: host batches on the depth arrays and streamed, device ingest from a BAM, the
command line with -r FASTA and -f GFF as one process and as `--gpus 3 --same-device`."""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

from ngs_amd import build, ffi, host
from tests import bamio, genome_util as gu
from tests.util import compare_contexts, json_equal

pytestmark = pytest.mark.gpu

FACETS_ALL = ffi.FACETS_DEFAULT | ffi.FACET_EDITS | ffi.FACET_FEATURES
KW = dict(facets=FACETS_ALL, max_read_len=1024, gc_seed=0x4E4753)     # the command line's context (bin 50 000, caps of the reference)


class World:
    pass


@pytest.fixture(scope="module")
def world(oracle_mod, tmp_path_factory):
    w = World()
    w.names, w.lens, w.primary = gu.header(64)
    assert len(w.names) == 195 and sum(w.primary) == 193
    w.bases, w.lower = gu.reference_bases(11, w.lens, soft_masked=True)
    w.rows, w.model = gu.gene_model(12, w.names, w.lens, w.primary, 6000)
    w.features = gu.model_arrays(w.model)
    w.clean = gu.sorted_records(gu.genome_records(13, 60_000, w.names, w.lens, w.bases, clean=True))
    w.dir = tmp_path_factory.mktemp("genome")
    w.bam = str(w.dir / "g.bam")
    w.clean = bamio.with_ids(w.clean, bamio.write_bam(w.bam, w.clean, w.names, w.lens, block_payload=30_000))
    o = oracle_mod.Oracle(w.lens, w.primary, ref_bases=w.bases, **KW)
    o.set_features(*w.features)
    o.process_batch(w.clean)
    assert o.finalize() == 0
    w.oracle = o
    w.doc = o.results(w.names)
    return w


def test_the_fixture_has_the_shape(world):
    """what the tests below rely on: records on chromosomes, chrM, chrEBV and contigs; empty sequences; overruns"""
    c = world.clean.cols
    seen = set(int(r) for r in np.unique(c["ref_id"]))
    dead = set(gu.empty_sequences(world.names))
    assert not (seen & dead) and len(seen) > 150
    for name in ("chr1", "chrX", "chrY", "chrM", "chrEBV", "chrUn_KI270302v1"):
        assert world.names.index(name) in seen or world.names.index(name) in dead, name
    cov = world.doc["coverage"]
    assert "chrM" not in cov["mean_coverage"] and "chrEBV" not in cov["mean_coverage"]            # not primary: coverage.rs:133-138
    assert "chr5" not in cov["mean_coverage"] and "chr1" in cov["mean_coverage"]                  # no record: coverage.rs:187-193
    assert len(cov["mean_coverage"]) == len([r for r in seen if world.primary[r]])
    assert cov["ignored"]["nonsensical_records"] > 1000                                            # reads past sequence ends
    assert sum(world.doc["edits"]["vaf_histogram"]["values"]) > 100_000
    assert world.doc["features"]["records"]["processed"] > 10_000


@pytest.mark.parametrize("sorted_input", [False, True])
def test_host_batches_all_seven_facets(gpu_lib, world, sorted_input):
    hb = world.clean
    with host.QcContext(world.lens, world.primary, ref_bases=world.bases, sorted_input=sorted_input, lib=gpu_lib, **KW) as gpu:
        gpu.set_features(*world.features)
        for lo in range(0, hb.n, 17_001):
            gpu.process_batch(hb.slice(lo, min(hb.n, lo + 17_001)))
        assert gpu.finalize() == 0
        compare_contexts(gpu, world.oracle, 195, FACETS_ALL, 50_000, world.lens)
        json_equal(gpu.results(world.names), world.doc)
        # a second file on the same context: the reset behind a finalize clears 195 sequences' worth of state
        gpu.reset()
        for lo in range(0, hb.n, 23_456):
            gpu.process_batch(hb.slice(lo, min(hb.n, lo + 23_456)))
        assert gpu.finalize() == 0
        json_equal(gpu.results(world.names), world.doc)


def test_raw_records_count_the_same_errors(gpu_lib, oracle_mod, world):
    """random_batch's raw records over the 195 sequences (ids of -1, positions beyond the end, any CIGAR): both sides count
    the same aborts and the same everything else, in file order and shuffled (depth arrays: no order needed)."""
    hb = gu.genome_records(14, 40_000, world.names, world.lens, world.bases, clean=False)
    o = oracle_mod.Oracle(world.lens, world.primary, ref_bases=world.bases, **KW)
    o.set_features(*world.features)
    o.process_batch(hb)
    rc_o = o.finalize(allow_malformed=True)
    with host.QcContext(world.lens, world.primary, ref_bases=world.bases, lib=gpu_lib, **KW) as gpu:
        gpu.set_features(*world.features)
        gpu.process_batch(hb.slice(0, 9_999))
        gpu.process_batch(hb.slice(9_999, hb.n))
        assert gpu.finalize(allow_malformed=True) == rc_o
        compare_contexts(gpu, o, 195, FACETS_ALL, 50_000, world.lens)
        json_equal(gpu.results(world.names), o.results(world.names))


def test_device_ingest_from_the_bam(gpu_lib, world):
    """BAM (195 @SQ lines: the header alone spans BGZF blocks) -> HIP inflate + parse -> all seven facets, streamed Coverage"""
    with host.QcContext(world.lens, world.primary, ref_bases=world.bases, sorted_input=True, lib=gpu_lib, **KW) as gpu:
        gpu.set_features(*world.features)
        h = C.c_void_p()
        assert gpu_lib.ngsq_bam_open(world.bam.encode(), 2, C.byref(h)) == 0, gpu_lib.ngsq_bam_last_error()
        assert gpu_lib.ngsq_bam_n_refs(h) == 195
        n = 0
        while True:
            b = ffi.Batch()
            assert gpu_lib.ngsq_bam_next_batch_device(h, gpu._ctx, 7_777, C.byref(b)) == 0, gpu_lib.ngsq_bam_last_error()
            if b.n_records == 0:
                break
            n += b.n_records
            assert gpu_lib.ngsq_process_batch(gpu._ctx, C.byref(b), ffi.PASS_BOTH) == 0, gpu_lib.ngsq_last_error(gpu._ctx)
        gpu_lib.ngsq_bam_close(h)
        assert n == world.clean.n
        assert gpu.finalize() == 0
        json_equal(gpu.results(world.names), world.doc)


@pytest.fixture(scope="module")
def ngs(lib):
    return build.build_cli(verbose=False)


def run(ngs, *args, env=None):
    return subprocess.run([ngs, *args], capture_output=True, text=True, env=env)


@pytest.fixture(scope="module")
def files(world):
    """the FASTA as the analysis set comes: soft-masked (about half of it lower case), its own sequence order, sequences the
    BAM does not have; the GFF gzipped"""
    fa = str(world.dir / "ref.fa")
    order = list(range(25, 195)) + list(range(25))      # contigs first: edits.rs:185-205 finds a sequence by name
    gu.write_fasta(fa, world.names, world.bases, world.lower, order=order, extra=[("chrUn_decoy_not_in_the_bam", "ACGTNNNNacgtn" * 40)])
    gff = str(world.dir / "model.gff3.gz")
    gu.write_gff(gff, world.rows + ["##FASTA", ">junk", "ACGT"])
    return fa, gff


@pytest.mark.parametrize("gpus", [1, 3])
def test_command_line_all_facets(ngs, gpu_lib, world, files, gpus, tmp_path):
    fa, gff = files
    out = tmp_path / "o"
    extra = ["--gpus", str(gpus), "--same-device"] if gpus > 1 else []
    r = run(ngs, "qc", world.bam, gu.GENOME, "-r", fa, "-f", gff, "-o", str(out), "--batch-records", "9001", *extra)
    assert r.returncode == 0, r.stderr
    got = json.load(open(out / "g.bam.results.json"))
    json_equal(got, world.doc)
    assert "  [*] Edits, Heavy" in r.stderr and "  [*] Genomic Features, Moderate" in r.stderr


def test_command_line_array_coverage_and_host_ingest(ngs, gpu_lib, world, files, tmp_path):
    fa, gff = files
    for k, extra in enumerate((["--coverage", "array"], ["--ingest", "host"])):
        out = tmp_path / f"o{k}"
        r = run(ngs, "-q", "qc", world.bam, gu.GENOME, "-r", fa, "-f", gff, "-o", str(out), *extra)
        assert r.returncode == 0, r.stderr
        json_equal(json.load(open(out / "g.bam.results.json")), world.doc)


def test_workers_load_the_sequences_their_byte_range_reaches(ngs, gpu_lib, world, files, tmp_path):
    """`ngs qc --gpus 3 -r` on a file with a REAL index: a worker asks the loader only for the sequences whose records its byte
    range of the file can hold (first virtual offsets from the BAI) -- N workers do not upload the FASTA N times -- and the
    document is still the oracle's.  A sequence left out by mistake would fail the run (`edits_bad_reference`), never compare
    with nothing."""
    fa, gff = files
    bam = str(tmp_path / "indexed.bam")
    bamio.write_bam(bam, world.clean, world.names, world.lens, block_payload=30_000, real_index=True)
    out = tmp_path / "o"
    r = run(ngs, "-v", "qc", bam, gu.GENOME, "-r", fa, "-f", gff, "-o", str(out), "--gpus", "3", "--same-device",
            env=dict(os.environ, NGSQ_REF_MARGIN_BYTES="65536"))
    assert r.returncode == 0, r.stderr
    got = json.load(open(out / "indexed.bam.results.json"))
    json_equal(got, world.doc)          # (the same BAM bytes as world.bam -- only the index differs -- so the same record ids)
    import re
    loaded = [int(m.group(1)) for m in re.finditer(r"the bases of (\d+) of the 195 sequences", r.stderr)]
    assert len(loaded) >= 1 and min(loaded) < 195, r.stderr[-2000:]      # (rank 0 narrates; at least it loaded a part only)
