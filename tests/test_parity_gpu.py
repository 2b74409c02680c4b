"""Parity tests proper: the HIP path through the C ABI vs the CPU oracle on the same
seeded inputs.  Bit-exact on every integer array; parsed-JSON equal on the
`Results` document (the reference's HashMap order is random).
Run on the GPU box with `pytest -m gpu`.
"""
import ctypes as C
import json
import os

import numpy as np
import pytest

from ngs_amd import ffi, host
from tests.util import (batch_from_records, compare_contexts, json_equal, make_edit_friendly, random_batch,
                        random_ref_bases, to_fixed_stride)

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hand_six_records.json")


def run_both(oracle_mod, lib, batches, ref_len, primary=None, facets=ffi.FACETS_DEFAULT, bin_size=1000,
             max_read_len=320, gc_seed=7, ref_bases=None, on_device=False, allow_malformed=True, names=None,
             features=None):
    kw = dict(facets=facets, bin_size=bin_size, max_read_len=max_read_len, gc_seed=gc_seed, ref_bases=ref_bases)
    orc = oracle_mod.Oracle(ref_len, primary, **kw)
    gpu = host.QcContext(ref_len, primary, lib=lib, **kw)
    if features is not None:
        orc.set_features(*features)
        gpu.set_features(*features)
    for hb in batches:
        orc.process_batch(hb)
        if on_device:
            db = gpu.upload(hb)
            gpu.process_batch(db)
        else:
            gpu.process_batch(hb)
    rc_o = orc.finalize(allow_malformed=allow_malformed)
    rc_g = gpu.finalize(allow_malformed=allow_malformed)
    assert rc_o == rc_g
    compare_contexts(gpu, orc, len(ref_len), facets, bin_size, ref_len)
    names = names or [f"chr{i + 1}" for i in range(len(ref_len))]
    json_equal(gpu.results(names), orc.results(names))
    return gpu, orc


def test_hand_golden(gpu_lib, oracle_mod):
    g = json.load(open(GOLD))
    cfg = g["config"]
    hb = batch_from_records(g["records"])
    gpu = host.QcContext(cfg["ref_len"], cfg["ref_is_primary"], facets=cfg["facets"], bin_size=cfg["bin_size"],
                         max_read_len=cfg["max_read_len"], lib=gpu_lib)
    gpu.process_batch(hb)
    gpu.finalize()
    json_equal(gpu.results(cfg["ref_names"]), g["expected"])


@pytest.mark.parametrize("sorted_input", [False, True])
def test_hand_golden_edits_multiseq(gpu_lib, sorted_input):
    """The second hand-derived case through the HIP path: Edits over every CIGAR operation, multi-sequence Coverage
    (arrays and streamed), a pileup deeper than the capacity, the f32 VAF edge (make_hand_goldens_edits.py)."""
    from tests.test_oracle_golden import load_gold_edits
    g = load_gold_edits()
    cfg = g["config"]
    hb = batch_from_records(g["records"])
    gpu = host.QcContext(cfg["ref_len"], cfg["ref_is_primary"], facets=cfg["facets"], bin_size=cfg["bin_size"],
                         max_read_len=cfg["max_read_len"], ref_bases=cfg["ref_bases"], sorted_input=sorted_input, lib=gpu_lib)
    gpu.process_batch(hb.slice(0, 1000))
    gpu.process_batch(hb.slice(1000, hb.n))
    gpu.finalize()
    json_equal(gpu.results(cfg["ref_names"]), g["expected"])
    gpu.close()


@pytest.mark.parametrize("on_device", [False, True])
def test_synthetic_fixed_150bp(gpu_lib, oracle_mod, on_device):
    """configs[1]/[2] shape at a size the oracle finishes in seconds."""
    n, L = 200_000, 500_000
    cfg = host.synth_config(n, ref_len=L, n_refs=2)
    hb = host.synth_host_batch(cfg, 0, n, gpu_lib)
    run_both(oracle_mod, gpu_lib, [hb], [L, 300_000], bin_size=50_000, max_read_len=150, on_device=on_device,
             allow_malformed=False)


def test_synthetic_mixed_lengths_and_cigars(gpu_lib, oracle_mod):
    """configs[4] shape: 50-300 bp, soft clips / indels / skips."""
    n, L = 100_000, 400_000
    cfg = host.synth_config(n, mode=ffi.SYNTH_MIXED, ref_len=L, n_refs=2)
    hb = host.synth_host_batch(cfg, 0, n, gpu_lib)
    gpu, _ = run_both(oracle_mod, gpu_lib, [hb], [L, 1000], bin_size=50_000, max_read_len=300,
                      allow_malformed=False)
    ops = gpu.general()["read_one_cigar_ops"]
    assert ops[0] and ops[1] and ops[2] and ops[3] and ops[4]  # M I D N S all present


def test_device_generator_matches_host_generator(gpu_lib):
    for mode in (ffi.SYNTH_FIXED, ffi.SYNTH_MIXED):
        cfg = host.synth_config(50_000, mode=mode, ref_len=1_000_000)
        hb = host.synth_host_batch(cfg, 10_000, 20_000, gpu_lib)
        with host.QcContext([1_000_000, 1000], lib=gpu_lib) as gpu:
            db = gpu.synth_device_batch(cfg, 10_000, 20_000)
            for name, a in hb.cols.items():
                if a is None:
                    continue
                got = gpu.download_column(db, name, a.size)
                np.testing.assert_array_equal(got, a, err_msg=f"mode {mode} column {name}")


@pytest.mark.parametrize("mode", ["fixed", "mixed"])
def test_genome_mode_generator_and_parity(gpu_lib, oracle_mod, mode):
    """GENOME mode of the synthetic generator (include/ngsq_shared.h; bench.py whole_genome and the realistic file): the records are
    spread over the 195 sequences of the GRCh38 header (chromosomes scaled down here), coordinate-sorted, never across a sequence's
    end, mates of 1 % of the pairs on the next sequence; the device generator equals the host's byte for byte, and all default
    facets + Edits on them equal the oracle's (the reads are sampled from the synthetic reference of their OWN sequence)."""
    from ngs_amd.genome_shape import grch38_no_alt
    names, lens, primary = grch38_no_alt(256)
    n = 120_000
    cfg = host.synth_config(n, mode=ffi.SYNTH_MIXED if mode == "mixed" else ffi.SYNTH_FIXED, genome=lens, lib=gpu_lib,
                            seq_model=ffi.SYNTH_SEQ_FROM_REFERENCE, file_style=ffi.SYNTH_FILE_CIGAR_MIX if mode == "fixed" else 0)
    hb = host.synth_host_batch(cfg, 0, n, gpu_lib)
    c = hb.cols
    key = (c["ref_id"].astype(np.int64) << 32) | c["pos"]
    assert (np.diff(key) >= 0).all() and len(np.unique(c["ref_id"])) > 150
    span = np.zeros(n, dtype=np.int64)
    co = c["cigar_off"] if c.get("cigar_off") is not None else np.arange(n + 1, dtype=np.uint64) * hb.cigar_stride
    for i in range(0, n, 37):      # (a sample: a Python loop)
        ops = c["cigar"][int(co[i]):int(co[i]) + int(c["n_cigar"][i])]
        span[i] = sum(int(x) >> 4 for x in ops if (int(x) & 15) in (0, 2, 3, 7, 8))
    assert (c["pos"][::37] + span[::37] <= np.array(lens)[c["ref_id"][::37]]).all()      # no read crosses the end of its sequence
    bases = [host.synth_reference(cfg, r, L, gpu_lib) for r, L in enumerate(lens)]
    kw = dict(facets=ffi.FACETS_DEFAULT | ffi.FACET_EDITS, max_read_len=320, gc_seed=3)
    orc = oracle_mod.Oracle(lens, primary, ref_bases=bases, **kw)
    orc.process_batch(hb)
    assert orc.finalize() == 0
    r1, r2, _ = orc.edits()
    assert int(r1[:6].sum() + r2[:6].sum()) > 0.98 * int(r1.sum() + r2.sum())          # sampled from the right sequence: few edits
    with host.QcContext(lens, primary, ref_bases=bases, sorted_input=True, lib=gpu_lib, **kw) as gpu:
        db = gpu.synth_device_batch(cfg, 0, n)
        for name_, a in hb.cols.items():
            if a is not None:
                np.testing.assert_array_equal(gpu.download_column(db, name_, a.size), a, err_msg=f"{mode} column {name_}")
        gpu.process_batch(db)
        assert gpu.finalize() == 0
        compare_contexts(gpu, orc, len(lens), kw["facets"], 50_000, lens)
        json_equal(gpu.results(names), orc.results(names))


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_edge_cases(gpu_lib, oracle_mod, seed):
    """Ragged, empty, unplaced, out-of-range, every flag bit, every CIGAR op."""
    rng = np.random.default_rng(seed)
    ref_len = [5000, 1, 777, 30_000]
    primary = [1, 1, 0, 1]
    hb = random_batch(rng, 20_000, ref_len)
    run_both(oracle_mod, gpu_lib, [hb], ref_len, primary, bin_size=[1000, 7, 50_000][seed - 1], max_read_len=320)


@pytest.mark.parametrize("max_len", [16, 31, 37, 100, 149, 150, 151, 255, 300])
@pytest.mark.parametrize("on_device", [False, True])
def test_fixed_pitch_rows_with_padding(gpu_lib, oracle_mod, max_len, on_device):
    """The fast (dense byte-stream) kernels: ragged reads in fixed-pitch rows, 0xFF
    quality padding, all-0xFF rows (missing qualities), odd pitches, and the same
    records through the offsets layout must give identical results."""
    rng = np.random.default_rng(max_len)
    ref_len = [9000, 2000]
    var = random_batch(rng, 7001, ref_len, max_len=max_len, min_len=0, weird=max_len >= 101)
    fixed = to_fixed_stride(var, min_len=max_len)
    g_fixed, _ = run_both(oracle_mod, gpu_lib, [fixed], ref_len, bin_size=333, max_read_len=320,
                          on_device=on_device)
    g_var, _ = run_both(oracle_mod, gpu_lib, [var], ref_len, bin_size=333, max_read_len=320, on_device=on_device)
    json_equal(g_fixed.results(["a", "b"]), g_var.results(["a", "b"]))


def test_quality_table_grows_with_the_reads(gpu_lib, oracle_mod):
    """quality_scores.rs:18 keeps a map per position: any read length works.  The table starts with max_read_len rows
    and grows to the longest read of each batch -- fixed-pitch rows say it by their pitch, host batches are looked at,
    the readers of ngsq_bam.h announce it (ngsq_batch.max_l_seq).  Only a DEVICE batch in the offsets layout that does
    not announce its longest read must keep to the rows there are: that is the one limit left (NGSQ_ERR_LIMIT)."""
    rng = np.random.default_rng(77)
    var = random_batch(rng, 500, [5000], max_len=60, weird=False)
    fixed = to_fixed_stride(var, min_len=80)
    run_both(oracle_mod, gpu_lib, [fixed], [5000], max_read_len=64, facets=ffi.FACET_QUALITY_SCORE)
    var2 = random_batch(rng, 500, [5000], max_len=80, min_len=70, weird=False)
    fixed2 = to_fixed_stride(var2)
    for hb in (fixed2, var2):
        g, _ = run_both(oracle_mod, gpu_lib, [hb], [5000], max_read_len=64, facets=ffi.FACET_QUALITY_SCORE)
        assert g.error_counts()["read_too_long"] == 0 and g.quality_scores().shape[0] == 128    # 80 cycles -> the next multiple of 64
        assert int(g.quality_scores()[64:80].sum()) > 0
    # a device batch in the offsets layout: announced -> grows; not announced -> the limit, reported as what it is
    for announce in (True, False):
        with host.QcContext([5000], lib=gpu_lib, max_read_len=64, facets=ffi.FACET_QUALITY_SCORE) as c:
            db = c.upload(var2)
            st = db.struct()
            st.max_l_seq = 80 if announce else 0
            assert gpu_lib.ngsq_process_batch(c._ctx, C.byref(st), ffi.PASS_BOTH) == 0
            if announce:
                c.finalize()
                o = oracle_mod.Oracle([5000], max_read_len=64, facets=ffi.FACET_QUALITY_SCORE)
                o.process_batch(var2)
                o.finalize()
                assert np.array_equal(c.quality_scores()[:80], o.quality_scores(80))
            else:
                with pytest.raises(host.NgsqError) as ei:
                    c.finalize()
                assert ei.value.code == ffi.ERR_LIMIT and "implementation limit: 500 read(s) longer than the quality table's 64 cycles" in ei.value.message


@pytest.mark.parametrize("length", [1025, 5000, 40_000])
def test_long_reads(gpu_lib, oracle_mod, length, tmp_path):
    """reads longer than 1024 bases used to stop `ngs qc` with NGSQ_ERR_LIMIT.  Reads of 1 025, 5 000 and 40 000
    bases (and short ones between them) through the offsets layout on the host and on the device, through fixed-pitch
    rows, and as a BAM file through the host reader, the device reader and the command line: the oracle's document."""
    import json, subprocess
    from ngs_amd import build
    from tests import bamio
    rng = np.random.default_rng(length)
    n = 600 if length <= 5000 else 60
    L = [max(3 * length, 50_000), 7_000]
    hb = random_batch(rng, n, L, max_len=length, min_len=length - 3, weird=False)
    short = random_batch(rng, n, L, max_len=200, min_len=30, weird=False)
    for on_device in (False, True):
        g, o = run_both(oracle_mod, gpu_lib, [short, hb, short], L, max_read_len=256, bin_size=5000, on_device=on_device)
        assert g.error_counts()["read_too_long"] == 0 and int(g.quality_scores()[length - 4].sum()) == n
    if length <= 5000:   # (a fixed-pitch batch of 40 kb rows is not what any reader would build)
        run_both(oracle_mod, gpu_lib, [to_fixed_stride(hb)], L, max_read_len=256, bin_size=5000)
    # as a file: coordinate-sorted, through both readers and the command line
    from tests.util import coordinate_sorted
    both = coordinate_sorted(bamio_concat(short, hb))
    both.cols["flag"] &= np.uint16(0xFFFF ^ 0x1)
    bam = str(tmp_path / "long.bam")
    names = ["chr1", "chr2"]
    both = bamio.with_ids(both, bamio.write_bam(bam, both, names, L, block_payload=40_000))
    o = oracle_mod.Oracle(L, max_read_len=256, gc_seed=0x4E4753, bin_size=50_000)
    o.process_batch(both)
    o.finalize()
    want = o.results(names)
    assert str(length) in want["quality_scores"]["scores"] and str(length + 1) not in want["quality_scores"]["scores"]
    ngs = build.build_cli(verbose=False)
    for ingest in ("device", "host"):
        out = tmp_path / ingest
        r = subprocess.run([ngs, "-q", "qc", bam, "GRCh38_no_alt_AnalysisSet", "-o", str(out), "--ingest", ingest, "--batch-records", "257"],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        json_equal(json.load(open(out / "long.bam.results.json")), want)


def bamio_concat(a, b):
    """Two offsets-layout batches as one."""
    cols = {}
    for k in host.FIXED_COLUMNS:
        cols[k] = np.concatenate([a.cols[k], b.cols[k]])
    for data, off in (("seq", "seq_off"), ("qual", "qual_off"), ("cigar", "cigar_off")):
        cols[data] = np.concatenate([a.cols[data], b.cols[data]])
        cols[off] = np.concatenate([a.cols[off], b.cols[off][1:] + a.cols[off][-1]]).astype(np.uint64)
    return host.HostBatch(a.n + b.n, cols, 0, 0, 0, 0)


def test_many_batches_equal_one_batch(gpu_lib, oracle_mod):
    rng = np.random.default_rng(11)
    ref_len = [20_000, 9000]
    hb = random_batch(rng, 30_000, ref_len, weird=True)
    cuts = [0, 1, 2, 4097, 4098, 12_345, 29_999, 30_000]
    parts = [hb.slice(a, b) for a, b in zip(cuts[:-1], cuts[1:])]
    gpu_parts, orc = run_both(oracle_mod, gpu_lib, parts, ref_len, bin_size=512)
    gpu_one, _ = run_both(oracle_mod, gpu_lib, [hb], ref_len, bin_size=512)
    json_equal(gpu_parts.results(["a", "b"]), gpu_one.results(["a", "b"]))


def test_empty_and_tiny_batches(gpu_lib, oracle_mod):
    ref_len = [1000]
    empty = batch_from_records([])
    one = batch_from_records([dict(flag=0, mapq=9, ref_id=0, pos=0, cigar="3M", seq="ACG", qual=[0, 93, 5])])
    run_both(oracle_mod, gpu_lib, [empty], ref_len)
    run_both(oracle_mod, gpu_lib, [empty, one, empty], ref_len)


def test_no_records_gives_null_summaries(gpu_lib, oracle_mod):
    gpu, _ = run_both(oracle_mod, gpu_lib, [], [1000, 2000])
    r = gpu.results(["a", "b"])
    assert r["general"]["summary"]["mapped_pct"] is None and r["quality_scores"] == {"scores": {}}


def test_device_batch_of_reads_without_bases(gpu_lib, oracle_mod):
    """Every read of an uploaded offsets-layout batch has l_seq 0: seq_bytes is 0 because the column is empty, not because the
    caller left it out (the parity sweep's seed 554 of round 6: refused as "needs seq_bytes" until then)."""
    recs = [dict(flag=f, mapq=30, ref_id=0, pos=p, mate_ref_id=-1, tlen=0, cigar=c, seq="", qual=None)
            for f, p, c in ((0, 5, "10M"), (0x10, 7, "*"), (0x4, -1, "*"), (0x400, 9, "3M2D3M"))]
    hb = batch_from_records(recs)
    orc = oracle_mod.Oracle([500], bin_size=100, max_read_len=320)
    orc.process_batch(hb)
    orc.finalize()
    with host.QcContext([500], bin_size=100, max_read_len=320, lib=gpu_lib) as gpu:
        gpu.process_batch(gpu.upload(hb))
        gpu.finalize()
        compare_contexts(gpu, orc, 1, ffi.FACETS_DEFAULT, 100, [500])
        assert gpu.gc_content()["ignored_too_short"] == 3 and gpu.gc_content()["ignored_flags"] == 1


def test_facet_subsets(gpu_lib, oracle_mod):
    """`--only FACET` (qc.rs:101-123): every other top-level key is null."""
    rng = np.random.default_rng(5)
    ref_len = [8000]
    hb = random_batch(rng, 5000, ref_len)
    for facets in (ffi.FACET_GENERAL, ffi.FACET_TEMPLATE_LENGTH, ffi.FACET_GC_CONTENT, ffi.FACET_QUALITY_SCORE,
                   ffi.FACET_COVERAGE, ffi.FACET_QUALITY_SCORE | ffi.FACET_GC_CONTENT | ffi.FACET_TEMPLATE_LENGTH):
        gpu, _ = run_both(oracle_mod, gpu_lib, [hb], ref_len, facets=facets)
        r = gpu.results(["s"])
        for key, bit in (("general", 1), ("template_length", 2), ("gc_content", 4), ("quality_scores", 8),
                         ("coverage", 16), ("edits", 32)):
            assert (r[key] is None) == (not facets & bit), (facets, key)


def test_pass_masks(gpu_lib, oracle_mod):
    """The two passes of the driver can be fed different record subsets (-n semantics)."""
    rng = np.random.default_rng(8)
    ref_len = [6000]
    hb = random_batch(rng, 4000, ref_len)
    orc = oracle_mod.Oracle(ref_len, bin_size=100, max_read_len=320)
    gpu = host.QcContext(ref_len, bin_size=100, max_read_len=320, lib=gpu_lib)
    a, b = hb.slice(0, 1000), hb.slice(500, 4000)
    for c in (orc, gpu):
        c.process_batch(a, ffi.PASS_RECORD)
        c.process_batch(b, ffi.PASS_SEQUENCE)
        c.finalize(allow_malformed=True)
    compare_contexts(gpu, orc, 1, ffi.FACETS_DEFAULT, 100, ref_len)


def test_malformed_records_are_reported_not_hidden(gpu_lib, oracle_mod):
    recs = [dict(flag=0x1, mapq=9, ref_id=-1, pos=5, mate_ref_id=0, cigar="3M", seq="ACG", qual=[1, 2, 3]),  # unwrap on None
            dict(flag=0, mapq=9, ref_id=0, pos=5, cigar="3M", seq="ACG", qual=[1, 94, 255])]               # score > 93
    hb = batch_from_records(recs)
    gpu = host.QcContext([100], lib=gpu_lib)
    gpu.process_batch(hb)
    with pytest.raises(host.NgsqError) as ei:
        gpu.finalize()
    assert ei.value.code == ffi.ERR_MALFORMED_RECORD
    assert gpu.error_counts()["missing_reference_id"] == 1 and gpu.error_counts()["bad_quality_score"] == 2
    run_both(oracle_mod, gpu_lib, [hb], [100])


def test_reset_makes_the_context_reusable(gpu_lib, oracle_mod):
    rng = np.random.default_rng(21)
    ref_len = [10_000, 400]
    hb1, hb2 = random_batch(rng, 6000, ref_len), random_batch(rng, 3000, ref_len)
    gpu = host.QcContext(ref_len, bin_size=300, max_read_len=320, lib=gpu_lib)
    gpu.process_batch(hb1)
    gpu.finalize(allow_malformed=True)
    first = gpu.results(["a", "b"])
    gpu.reset()
    gpu.process_batch(hb2)
    gpu.finalize(allow_malformed=True)
    orc = oracle_mod.Oracle(ref_len, bin_size=300, max_read_len=320)
    orc.process_batch(hb2)
    orc.finalize(allow_malformed=True)
    json_equal(gpu.results(["a", "b"]), orc.results(["a", "b"]))
    gpu.reset()
    gpu.process_batch(hb1)
    gpu.finalize(allow_malformed=True)
    json_equal(gpu.results(["a", "b"]), first)  # idempotent


def test_pileup_too_large_and_large_depth(gpu_lib, oracle_mod):
    """depth > cov_cap is `ignored` (coverage.rs:211-213) but still in the bin means."""
    recs = [dict(flag=0, ref_id=0, pos=10, cigar="20M", seq="A" * 20, qual=[5] * 20)] * 70 + \
           [dict(flag=0, ref_id=0, pos=15, cigar="2M", seq="AC", qual=[5] * 2)] * 10
    hb = batch_from_records(recs)
    gpu, _ = run_both(oracle_mod, gpu_lib, [hb], [100], bin_size=16, facets=ffi.FACET_COVERAGE)
    orc = oracle_mod.Oracle([100], facets=ffi.FACET_COVERAGE, bin_size=16, cov_cap=64)
    g2 = host.QcContext([100], facets=ffi.FACET_COVERAGE, bin_size=16, cov_cap=64, lib=gpu_lib)
    for c in (orc, g2):
        c.process_batch(hb)
        c.finalize()
    assert g2.coverage_sequence(0)[2] == 20  # 20 positions at depth 70/80 > 64
    json_equal(g2.results(["s"]), orc.results(["s"]))


@pytest.mark.parametrize("seed", [4, 5])
def test_edits_facet(gpu_lib, oracle_mod, seed):
    rng = np.random.default_rng(seed)
    ref_len = [6000, 2500]
    bases = random_ref_bases(rng, ref_len)
    hb = make_edit_friendly(random_batch(rng, 8000, ref_len, weird=False, min_len=1), rng, bases, ref_len)
    gpu, _ = run_both(oracle_mod, gpu_lib, [hb], ref_len, facets=ffi.FACETS_DEFAULT | ffi.FACET_EDITS,
                      ref_bases=bases, bin_size=500)
    r1, r2, vaf = gpu.edits()
    assert r1.sum() + r2.sum() > 1000 and vaf.sum() > 1000
    # malformed walks (random CIGARs) must be counted identically
    hb2 = random_batch(rng, 4000, ref_len)
    run_both(oracle_mod, gpu_lib, [hb2], ref_len, facets=ffi.FACET_EDITS, ref_bases=bases)
    # sequence without bases in the FASTA
    run_both(oracle_mod, gpu_lib, [hb], ref_len, facets=ffi.FACET_EDITS, ref_bases=[bases[0], None])


def brute_force_refs_alts(hb, ref_len, bases):
    """refs_per_position / alts_per_position (edits.rs:276-291) by a literal walk in Python: independent of the oracle."""
    refs = [np.zeros(L + 1, dtype=np.uint32) for L in ref_len]
    alts = [np.zeros(L + 1, dtype=np.uint32) for L in ref_len]
    c = hb.cols
    for i in range(hb.n):
        f, r, pos, l = int(c["flag"][i]), int(c["ref_id"][i]), int(c["pos"][i]), int(c["l_seq"][i])
        if r < 0 or pos < 0 or (f & 0x404) or bases[r] is None:
            continue
        cig = [int(x) for x in (c["cigar"][int(c["cigar_off"][i]):int(c["cigar_off"][i + 1])] if c["cigar_off"] is not None
                                 else c["cigar"][i * hb.cigar_stride:i * hb.cigar_stride + int(c["n_cigar"][i])])]
        span = sum(x >> 4 for x in cig if (x & 15) in (0, 2, 3, 7, 8))
        s = pos + 1
        if s + span - 1 == 0 or s > ref_len[r] or s + span - 1 > ref_len[r]:
            continue
        sq = (c["seq"][int(c["seq_off"][i]):int(c["seq_off"][i + 1])] if c["seq_off"] is not None
              else c["seq"][i * hb.seq_stride:i * hb.seq_stride + (l + 1) // 2])
        read = np.empty(2 * len(sq), dtype=np.uint8)
        read[0::2], read[1::2] = sq >> 4, sq & 15
        qp = rp = 0
        for x in cig:
            op, ln = x & 15, x >> 4
            if op == 0:
                m = min(ln, l - qp)
                same = read[qp:qp + m] == bases[r][pos + rp:pos + rp + m]
                np.add.at(refs[r], s + rp + np.nonzero(same)[0], 1)
                np.add.at(alts[r], s + rp + np.nonzero(~same)[0], 1)
                if m < ln:
                    break
                qp, rp = qp + ln, rp + ln
            else:
                if op in (1, 4, 7, 8):
                    if qp + ln > l:
                        break
                    qp += ln
                if op in (2, 3, 7, 8):
                    rp += ln
    return refs, alts


@pytest.mark.parametrize("mode", ["fixed", "aligner", "mixed", "iid", "random"])
def test_edits_positions_against_a_literal_walk(gpu_lib, oracle_mod, mode):
    """Round 4's Edits kernel keeps the `M` cover as a difference array and only the mismatches per position; refs = cover -
    alts appears at the teardown.  ngsq_get_edits_positions against a literal Python walk of every record, and the histograms
    against the oracle, on reads SAMPLED FROM the reference (0.5 % substitutions: the sparse case the kernel is built for,
    fixed 150 bases, the same with an aligner's CIGARs -- 9 % soft-clipped, 3 % an insertion, 3 % a deletion: fixed-pitch rows
    whose marked records the walk kernel lists and takes 64 at a time -- and the 50-300 base CIGAR mix), on independent bases (three in four mismatch: every dword is revisited)
    and on random records (every CIGAR shape, unsorted, reads that run out of bases)."""
    rng = np.random.default_rng(77)
    if mode == "random":
        ref_len = [6000, 2500]
        bases = random_ref_bases(rng, ref_len)
        hb = make_edit_friendly(random_batch(rng, 6000, ref_len, weird=False, min_len=1), rng, bases, ref_len)
    else:
        ref_len = [150_000, 20_000]
        cfg = host.synth_config(30_000, mode=ffi.SYNTH_MIXED if mode == "mixed" else ffi.SYNTH_FIXED, ref_len=ref_len[0], n_refs=2,
                                seq_model=ffi.SYNTH_SEQ_IID if mode == "iid" else ffi.SYNTH_SEQ_FROM_REFERENCE,
                                file_style=ffi.SYNTH_FILE_CIGAR_MIX if mode == "aligner" else 0)
        bases = [host.synth_reference(cfg, r, L, gpu_lib) for r, L in enumerate(ref_len)]
        hb = host.synth_host_batch(cfg, 0, 30_000, gpu_lib)
    gpu, orc = run_both(oracle_mod, gpu_lib, [hb], ref_len, facets=ffi.FACET_EDITS, ref_bases=bases)
    want_refs, want_alts = brute_force_refs_alts(hb, ref_len, bases)
    for r in range(len(ref_len)):
        refs, alts = gpu.edits_positions(r)
        assert np.array_equal(alts, want_alts[r]) and np.array_equal(refs, want_refs[r]), (mode, r)
    r1, r2, vaf = gpu.edits()
    if mode == "aligner":
        assert (hb.cols["n_cigar"] == 3).mean() > 0.04 and hb.cigar_stride == 3
    if mode in ("fixed", "aligner", "mixed"):      # ~0.75 substitutions per 150 bases: most reads have none or one
        assert r1[0] + r2[0] > 0.2 * hb.n and (r1[:4].sum() + r2[:4].sum()) > 0.9 * (r1.sum() + r2.sum())
    if mode == "iid":
        assert r1[:50].sum() + r2[:50].sum() == 0


@pytest.mark.parametrize("layout", ["rows", "rows_gc", "offsets"])
@pytest.mark.parametrize("spacing", [12, 40, 300, 2000])
def test_edits_on_sparse_sorted_reads(gpu_lib, oracle_mod, layout, spacing):
    """Sorted reads further apart than a wave's LDS window holds 64 of (low-depth data: 64 x 40 positions against 1408 entries): the
    reads of a pass beyond the window's end stay on the fast path -- their cover and their mismatches go straight to the arrays
    (round 6) -- up to the 15 k positions the descriptor can say; at 300 and 2000 apart the later ones of a pass are the walk's.
    Positions against the literal walk, histograms against the oracle; substitutions in every read, second M's, clips."""
    rng = np.random.default_rng(1000 + spacing)
    n = 3000
    L = [n * spacing + 5000, 5000]
    bases = random_ref_bases(rng, L)
    letters = "=ACMGRSVTWYHKDBN"
    recs = []
    for i in range(n):
        pos = i * spacing + int(rng.integers(0, max(1, spacing // 2)))
        shape = rng.random()
        l = 150 if layout != "offsets" else int(rng.integers(40, 300))
        if shape < 0.6:
            cig, ref_at = f"{l}M", [(pos, 0, l)]
        elif shape < 0.75:
            a = int(rng.integers(1, 20))
            cig, ref_at = f"{a}S{l - a}M", [(pos, a, l)]
        elif shape < 0.9:
            a, g = int(rng.integers(5, l - 5)), int(rng.integers(1, 3000))
            cig, ref_at = f"{a}M{g}{'DN'[int(rng.integers(0, 2))]}{l - a}M", [(pos, 0, a), (pos + a + g, a, l)]
        else:
            a, g = int(rng.integers(5, l - 12)), int(rng.integers(1, 9))
            cig, ref_at = f"{a}M{g}I{l - a - g}M", [(pos, 0, a), (pos + a, a + g, l)]
        codes = rng.choice(np.array([1, 2, 4, 8], dtype=np.uint8), l)
        for p0, q0, q1 in ref_at:                              # the reference's bases under the M's ...
            chunk = bases[0][p0:p0 + (q1 - q0)]
            codes[q0:q0 + len(chunk)] = chunk
        for k in rng.integers(0, l, int(rng.integers(0, 4))):  # ... with up to three substitutions
            codes[k] = 15
        recs.append(dict(flag=int(rng.choice([0, 0x10, 0x40, 0x80])), mapq=60, ref_id=0, pos=pos, mate_ref_id=-1, tlen=0, cigar=cig,
                         seq="".join(letters[c] for c in codes), qual=[30] * l))
    hb = batch_from_records(recs)
    if layout != "offsets":
        hb = to_fixed_stride(hb)
    facets = ffi.FACET_EDITS | (ffi.FACET_GC_CONTENT if layout == "rows_gc" else 0)
    gpu, orc = run_both(oracle_mod, gpu_lib, [hb], L, facets=facets, ref_bases=bases)
    assert not any(orc.error_counts().values())
    want_refs, want_alts = brute_force_refs_alts(batch_from_records(recs), L, bases)
    refs, alts = gpu.edits_positions(0)
    assert np.array_equal(alts, want_alts[0]) and np.array_equal(refs, want_refs[0])
    assert int(alts.sum()) > n // 2


@pytest.mark.parametrize("mode", ["fixed", "aligner", "subst25", "ragged_rows", "ids"])
def test_gc_content_tallied_by_the_edits_kernel(gpu_lib, oracle_mod, mode):
    """With GC Content and Edits both enabled, fixed-pitch rows of up to 160 bases are scanned ONCE: k_edits_rows tallies the GC
    window from the sequence bytes it compares and k_gc is not launched (round 5).  Every GC counter and the histogram against
    the oracle -- 150-base reads (plain, an aligner's CIGARs, a quarter of the bases substituted), rows of 60-160 bases padded to
    one pitch (reads below the 100-base window, windows at every offset and parity, duplicates / secondaries), records with
    identities of their own (the offset is drawn from them) -- and the launch counts that say which kernel did it."""
    from tests.util import to_fixed_stride
    rng = np.random.default_rng(91)
    facets = ffi.FACETS_DEFAULT | ffi.FACET_EDITS
    if mode in ("ragged_rows", "ids"):
        ref_len = [9000, 2500]
        bases = random_ref_bases(rng, ref_len)
        hb = to_fixed_stride(make_edit_friendly(random_batch(rng, 9000, ref_len, weird=False, min_len=60, max_len=160), rng, bases, ref_len))
        assert hb.seq_stride == 80 and hb.cols["seq_off"] is None
        if mode == "ids":
            hb.cols["record_id"] = rng.integers(0, 1 << 62, hb.n).astype(np.uint64)
    else:
        ref_len = [150_000, 20_000]
        cfg = host.synth_config(30_000, ref_len=ref_len[0], n_refs=2, file_style=ffi.SYNTH_FILE_CIGAR_MIX if mode == "aligner" else 0,
                                seq_model=ffi.synth_seq_subst(0.25) if mode == "subst25" else ffi.SYNTH_SEQ_FROM_REFERENCE)
        bases = [host.synth_reference(cfg, r, L, gpu_lib) for r, L in enumerate(ref_len)]
        hb = host.synth_host_batch(cfg, 0, 30_000, gpu_lib)
    for on_device in (False, True):
        gpu, orc = run_both(oracle_mod, gpu_lib, [hb.slice(0, hb.n // 3), hb.slice(hb.n // 3, hb.n)], ref_len, facets=facets, ref_bases=bases,
                            on_device=on_device)
        t = gpu.kernel_timing()
        assert t["gc"]["launches"] == 0 and t["edits"]["launches"] == 2, t     # the GC window was tallied by the Edits launch
        g = gpu.gc_content()
        assert g["processed"] > (500 if mode in ("ragged_rows", "ids") else 0.5 * hb.n) and (mode != "ragged_rows" or g["ignored_too_short"] > 100)
    # the same records through the offsets layout keep k_gc (and give the same document: run_both compares with the oracle)
    if mode == "fixed":
        ragged = host.synth_host_batch(host.synth_config(30_000, mode=ffi.SYNTH_MIXED, ref_len=ref_len[0], n_refs=2, seq_model=ffi.SYNTH_SEQ_FROM_REFERENCE),
                                       0, 30_000, gpu_lib)
        gpu, _ = run_both(oracle_mod, gpu_lib, [ragged], ref_len, facets=facets, ref_bases=bases)
        assert gpu.kernel_timing()["gc"]["launches"] == 1


@pytest.mark.parametrize("sorted_rows,cigar_offsets", [(True, False), (True, True), (False, False), (True, "offsets layout")])
def test_edits_second_segment_edge_cases(gpu_lib, oracle_mod, sorted_rows, cigar_offsets):
    """k_edits_rows compares the second M of `M (I|D) M` in a step of its own (edits_kernel.hip 2c), against the reference
    del - ins bases further on.  Every place that arithmetic can go wrong, one record each on 150-base fixed-pitch rows, the
    literal Python walk and the oracle as judges: the indel at a window boundary (32, 64, 96, 128 bases), one base before and
    behind it, first M of one base, second M of one base, insertions and deletions of 1, 2, 7, 8 and 15 bases (the shift odd and
    even: both packed copies of the reference), 16 bases (the walk kernel's until the gap got a word of its own), positions 0, 15 (walk) and 16,
    odd and even; skips that end inside the wave's window and kilobases beyond it."""
    from tests.util import to_fixed_stride
    rng = np.random.default_rng(4242)
    ref_len = [40_000, 3_000]
    bases = random_ref_bases(rng, ref_len)
    L, recs, pos = 150, [], 16
    for a in (1, 2, 31, 32, 33, 63, 64, 65, 96, 127, 128, 129, 140, 148):
        for g, op in ((1, "I"), (2, "I"), (7, "I"), (8, "I"), (15, "I"), (16, "I"), (1, "D"), (2, "D"), (7, "D"), (8, "D"), (15, "D"), (16, "D")):
            m2 = L - a - (g if op == "I" else 0)
            if m2 < 1:
                continue
            for p in (pos, pos + 1):
                # the read: the reference under both M segments, random bases in the insertion, a substitution in each segment
                ref0 = p
                seg1 = bases[0][ref0:ref0 + a].copy()
                seg2 = bases[0][ref0 + a + (g if op == "D" else 0):ref0 + a + (g if op == "D" else 0) + m2].copy()
                seg1[a // 2] = 1 if seg1[a // 2] != 1 else 2
                seg2[m2 // 2] = 4 if seg2[m2 // 2] != 4 else 8
                codes = np.concatenate([seg1, rng.choice(np.array([1, 2, 4, 8], dtype=np.uint8), g if op == "I" else 0), seg2])
                recs.append(dict(flag=0x40 if len(recs) % 2 else 0, mapq=60, ref_id=0, pos=p, mate_ref_id=-1, tlen=0,
                                 cigar=f"{a}M{g}{op}{m2}M", seq="".join("=ACMGRSVTWYHKDBN"[c] for c in codes), qual=[30] * L))
            pos += 5
    # a skip (N) between two M: the second M is compared kilobases further on (its cover and its mismatches go straight to the arrays
    # when it lies beyond the wave's window)
    for skip in (100, 1300, 5000):
        for p in (pos, pos + 1):
            a1 = 70
            seg1, seg2 = bases[0][p:p + a1].copy(), bases[0][p + a1 + skip:p + a1 + skip + L - a1].copy()
            seg1[3] = 1 if seg1[3] != 1 else 2
            seg2[-1] = 4 if seg2[-1] != 4 else 8
            recs.append(dict(flag=0x40, mapq=60, ref_id=0, pos=p, mate_ref_id=-1, tlen=0, cigar=f"{a1}M{skip}N{L - a1}M",
                             seq="".join("=ACMGRSVTWYHKDBN"[c] for c in np.concatenate([seg1, seg2])), qual=[30] * L))
        pos += 7
    for p in (0, 15):   # too close to the sequence's start for the shifted reference: the walk kernel's
        recs.append(dict(flag=0, mapq=60, ref_id=0, pos=p, mate_ref_id=-1, tlen=0, cigar="70M3I77M",
                         seq="".join("=ACMGRSVTWYHKDBN"[c] for c in np.concatenate([bases[0][p:p + 70], [1, 1, 1], bases[0][p + 70:p + 147]])), qual=[30] * L))
    if sorted_rows:
        recs.sort(key=lambda r: r["pos"])
    hv = batch_from_records(recs)
    hb = to_fixed_stride(hv)
    assert hb.seq_stride == 75 and hb.cigar_stride == 3
    if cigar_offsets == "offsets layout":   # everything through offsets: the RAGGED variant of the window lanes
        hb = hv
    elif cigar_offsets:   # fixed-pitch SEQ / QUAL rows with the CIGARs through offsets: what the device reader makes of an aligner's file
        cols = dict(hb.cols)
        cols["cigar"], cols["cigar_off"] = hv.cols["cigar"], hv.cols["cigar_off"]
        hb = host.HostBatch(hb.n, cols, hb.seq_stride, hb.qual_stride, 0, 0)
    gpu, orc = run_both(oracle_mod, gpu_lib, [hb], ref_len, facets=ffi.FACET_EDITS, ref_bases=bases)
    want_refs, want_alts = brute_force_refs_alts(hb, ref_len, bases)
    refs, alts = gpu.edits_positions(0)
    assert np.array_equal(alts, want_alts[0]) and np.array_equal(refs, want_refs[0])
    r1, r2, _ = gpu.edits()
    assert int(r1.sum() + r2.sum()) == len(recs) and int(r1[2] + r2[2]) >= len(recs) - 2   # two substitutions per read


@pytest.mark.parametrize("order", ["sorted", "unsorted", "one position"])
@pytest.mark.parametrize("subst", [0.0, 0.05, 0.6])
def test_edits_ragged_window_edge_cases(gpu_lib, oracle_mod, order, subst):
    """The offsets layout through the window lanes (k_edits_rows<.., RAGGED>, round 5): a record has ceil(bytes / 16) windows of its
    own and a byte map in LDS says which record a window of the pass belongs to.  Reads of every length around the window size
    (1..49 bases), around 100, 150, 256 and 300, around the variant's limit (255 bytes: 509..512 bases, the longer ones are the walk
    kernel's) and far beyond it, with no bases at all, soft clips on either end, an insertion or a deletion, the first and the last
    base substituted -- sorted (the windows a pass needs fit the LDS window), unsorted (most records are left to the walk) and all
    on one position (a pass's cover on two entries); the literal Python walk and the oracle as judges."""
    rng = np.random.default_rng(991)
    ref_len = [60_000, 4_000]
    bases = random_ref_bases(rng, ref_len)
    lens = list(range(1, 50)) + [63, 64, 65, 95, 96, 97, 100, 127, 128, 129, 149, 150, 151, 255, 256, 257, 299, 300, 301,
                                 479, 480, 481, 508, 509, 510, 511, 512, 513, 600, 1000, 2200]
    recs, pos = [], 20
    code = "=ACMGRSVTWYHKDBN"

    def sample(r, p, m):
        x = bases[r][p:p + m].copy()
        if m:
            x[0] = 1 if x[0] != 1 else 2          # the first and the last compared base
            x[-1] = 4 if x[-1] != 4 else 8
        if subst:
            hit = rng.random(m) < subst
            x[hit] = rng.choice(np.array([1, 2, 4, 8, 15], dtype=np.uint8), int(hit.sum()))
        return x
    for rep in range(3):
        for l in lens:
            for shape in ("M", "SM", "MS", "SMS", "MIM", "MDM", "MNM"):
                r = 1 if (l < 300 and rep == 2 and shape == "M") else 0
                p = pos if order != "one position" else 500
                if p + l + 40 > ref_len[r]:
                    p = int(rng.integers(16, ref_len[r] - l - 40))
                a = z = 0
                if "S" in shape and l >= 3:
                    a = int(rng.integers(1, max(2, l // 3))) if shape[0] == "S" else 0
                    z = int(rng.integers(1, max(2, l // 3))) if shape[-1] == "S" else 0
                m = l - a - z
                if shape in ("MIM", "MDM", "MNM") and l >= 8:
                    # (a deletion of up to 40 bases, a skip that ends inside the wave's window of ~1100 positions, at its end, or far beyond it)
                    g = int(rng.integers(1, 6)) if shape == "MIM" else int(rng.integers(1, 41)) if shape == "MDM" else int(rng.choice([1, 90, 700, 1100, 1160, 9000]))
                    g = min(g, l - 3) if shape == "MIM" else g
                    m1 = int(rng.integers(1, l - (g if shape == "MIM" else 0) - 1))
                    if p + l + g + 40 > ref_len[r]:
                        p = int(rng.integers(16, ref_len[r] - l - g - 40))
                    if shape == "MIM":
                        m2 = l - m1 - g
                        x = np.concatenate([sample(r, p, m1), rng.choice(np.array([1, 2, 4, 8], dtype=np.uint8), g), sample(r, p + m1, m2)])
                        cigar = f"{m1}M{g}I{m2}M"
                    else:
                        m2 = l - m1
                        x = np.concatenate([sample(r, p, m1), sample(r, p + m1 + g, m2)])
                        cigar = f"{m1}M{g}{'D' if shape == 'MDM' else 'N'}{m2}M"
                else:
                    x = np.concatenate([rng.choice(np.array([1, 2, 4, 8], dtype=np.uint8), a), sample(r, p, m),
                                        rng.choice(np.array([1, 2, 4, 8], dtype=np.uint8), z)])
                    cigar = (f"{a}S" if a else "") + f"{m}M" + (f"{z}S" if z else "")
                recs.append(dict(flag=(0x40 if len(recs) % 3 else 0x80) | (0x10 if len(recs) % 5 == 0 else 0), mapq=60, ref_id=r, pos=p,
                                 mate_ref_id=-1, tlen=0, cigar=cigar, seq="".join(code[c] for c in x), qual=[30] * l))
                pos += int(rng.integers(0, 4))
    # records with nothing for the window lanes: no bases (the reference aborts on them: counted), unmapped, no CIGAR
    recs.append(dict(flag=0, mapq=60, ref_id=0, pos=700, mate_ref_id=-1, tlen=0, cigar="10M", seq="", qual=[]))
    recs.append(dict(flag=4, mapq=0, ref_id=-1, pos=-1, mate_ref_id=-1, tlen=0, cigar="*", seq="ACGT" * 30, qual=[30] * 120))
    recs.append(dict(flag=0, mapq=60, ref_id=0, pos=800, mate_ref_id=-1, tlen=0, cigar="*", seq="ACGT" * 30, qual=[30] * 120))
    if order == "sorted":
        recs.sort(key=lambda q: (q["ref_id"] if q["ref_id"] >= 0 else 99, q["pos"]))
    elif order == "unsorted":
        rng.shuffle(recs)
    hb = batch_from_records(recs)
    assert hb.cols["seq_off"] is not None
    gpu, orc = run_both(oracle_mod, gpu_lib, [hb], ref_len, facets=ffi.FACET_EDITS, ref_bases=bases)
    want_refs, want_alts = brute_force_refs_alts(hb, ref_len, bases)
    for r in range(2):
        refs, alts = gpu.edits_positions(r)
        assert np.array_equal(alts, want_alts[r]) and np.array_equal(refs, want_refs[r]), r
    r1, r2, _ = gpu.edits()
    assert int(r1.sum() + r2.sum()) >= 0.9 * len(recs)     # (the longest reads at 60 % have more than 512 edits: counted as such)


def test_lane_per_record_edits_kernel_still_agrees(gpu_lib):
    """`k_edits` (a lane per record) is what the offsets layout ran on until round 5 and what is left for references beyond
    4 Gbases (the window lanes address the two packed copies with 32-bit offsets): NGSQ_EDITS_PER_RECORD=1 sends every batch through
    it.  The Edits tests of this file once more in a child process with that set."""
    import subprocess
    import sys
    if os.environ.get("NGSQ_EDITS_PER_RECORD"):
        pytest.skip("already the child")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider",
                        "-k", "test_edits_facet or test_edits_positions_against_a_literal_walk or test_edits_second_segment_edge_cases "
                              "or (test_edits_ragged_window_edge_cases and sorted)"],
                       capture_output=True, text=True, timeout=1500, env=dict(os.environ, NGSQ_EDITS_PER_RECORD="1"),
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout


def test_reference_bases_must_be_4_bit_codes(gpu_lib):
    bad = np.full(100, 1, dtype=np.uint8)
    bad[57] = 65   # an ASCII letter instead of a code
    with pytest.raises(Exception, match="4-bit"):
        host.QcContext([100], facets=ffi.FACET_EDITS, ref_bases=[bad], lib=gpu_lib)


def test_vaf_f32_rounding_exhaustive_small_totals(gpu_lib, oracle_mod):
    """Every (alts, total) with total <= 128 through both VAF paths (f32 divide, multiply, truncate)."""
    T = 128
    L = T * (T + 1) // 2 + T
    ref = np.full(L, 1, dtype=np.uint8)
    recs, p = [], 0
    for total in range(1, T + 1):
        for alts in range(0, total + 1, max(1, total // 16)):
            recs += [dict(flag=0, ref_id=0, pos=p, cigar="1M", seq="C", qual=[1])] * alts
            recs += [dict(flag=0, ref_id=0, pos=p, cigar="1M", seq="A", qual=[1])] * (total - alts)
            p += 1
    assert p <= L
    run_both(oracle_mod, gpu_lib, [batch_from_records(recs)], [L], facets=ffi.FACET_EDITS, ref_bases=[ref])


def test_sharded_state_sums_to_single_context(gpu_lib, oracle_mod):
    """shard the records over contexts, add the exchange blocks, finalize once."""
    rng = np.random.default_rng(31)
    ref_len = [40_000, 3000]
    hb = random_batch(rng, 24_000, ref_len, weird=True)
    kw = dict(bin_size=700, max_read_len=320, gc_seed=3)
    shards = [hb.slice(0, 9000), hb.slice(9000, 9001), hb.slice(9001, 24_000)]
    blocks = None
    for sh in shards:
        with host.QcContext(ref_len, lib=gpu_lib, **kw) as c:
            c.process_batch(sh)
            c.synchronize()
            got = [c.state_download(w) for w in (0, 1)]
        blocks = got if blocks is None else [a + b for a, b in zip(blocks, got)]
    total = host.QcContext(ref_len, lib=gpu_lib, **kw)
    total.state_upload(0, blocks[0])
    total.state_upload(1, blocks[1])
    total.finalize(allow_malformed=True)
    orc = oracle_mod.Oracle(ref_len, **kw)
    orc.process_batch(hb)
    orc.finalize(allow_malformed=True)
    compare_contexts(total, orc, 2, ffi.FACETS_DEFAULT, 700, ref_len)
    json_equal(total.results(["a", "b"]), orc.results(["a", "b"]))


def test_edits_block_uploaded_alone_is_torn_down_and_reset(gpu_lib, oracle_mod):
    """A host that restores or merges ONLY the edits block (ngsq_state_upload which = 2): the teardown, the per-position
    getter and the reset behind the finalize act on the sequences whose `Edits wrote here` word is set -- an upload sets them
    (it did not; the VAF histogram came back empty and the next run inherited the uploaded data)."""
    rng = np.random.default_rng(77)
    ref_len = [5000, 1800, 900]
    bases = random_ref_bases(rng, ref_len)
    hb = make_edit_friendly(random_batch(rng, 5000, ref_len, weird=False, min_len=1), rng, bases, ref_len)
    kw = dict(facets=ffi.FACET_EDITS, ref_bases=bases)
    with host.QcContext(ref_len, lib=gpu_lib, **kw) as src:
        src.process_batch(hb)
        src.synchronize()
        block = src.state_download(2)
        src.finalize(allow_malformed=True)
        want_vaf = src.edits()[2].copy()
        want_pos = [src.edits_positions(r) for r in range(3)]
    assert want_vaf.sum() > 500
    with host.QcContext(ref_len, lib=gpu_lib, **kw) as dst:
        dst.state_upload(2, block)          # nothing else: no counters, no records
        dst.finalize(allow_malformed=True)
        np.testing.assert_array_equal(dst.edits()[2], want_vaf)
        for r in range(3):
            for a, b in zip(dst.edits_positions(r), want_pos[r]):
                np.testing.assert_array_equal(a, b)
        # ... and the next run starts from zero
        dst.reset()
        hb2 = make_edit_friendly(random_batch(rng, 700, ref_len, weird=False, min_len=1), rng, bases, ref_len)
        dst.process_batch(hb2)
        dst.finalize(allow_malformed=True)
        orc = oracle_mod.Oracle(ref_len, **kw)
        orc.process_batch(hb2)
        orc.finalize(allow_malformed=True)
        compare_contexts(dst, orc, 3, ffi.FACET_EDITS, 50_000, ref_len)
        # the upload before the counters block (whose words are zero) still counts
        dst.reset()
        dst.state_upload(2, block)
        dst.state_upload(0, np.zeros_like(dst.state_download(0)))
        dst.finalize(allow_malformed=True)
        np.testing.assert_array_equal(dst.edits()[2], want_vaf)


@pytest.mark.parametrize("n,sorted_input", [(4_000_000, False), (100_000_000, True), (100_000_000, False)])
def test_full_size_properties(gpu_lib, n, sorted_input):
    """At sizes the oracle cannot reach quickly: size-independent invariants on records generated in HBM over a chr1-sized
    sequence -- 4 M, and BASELINE configs[2]'s full 100 M (26 GB of columns) with Coverage streamed and on the difference arrays."""
    L = 248_956_422
    cfg = host.synth_config(n, ref_len=L, n_refs=2)
    with host.QcContext([L, 242_193_529], max_read_len=150, timing=True, sorted_input=sorted_input, lib=gpu_lib) as gpu:
        db = gpu.synth_device_batch(cfg, 0, n)
        gpu.process_batch(db)
        gpu.finalize()
        g = gpu.general()
        assert g["total"] == n and g["primary"] + g["secondary"] + g["supplementary"] == n
        q = gpu.quality_scores()
        assert (q.sum(axis=1) == n).all()  # every record reaches every cycle exactly once
        h, processed, ignored = gpu.template_length()
        assert processed + ignored == n and h.sum() == processed
        gc = gpu.gc_content()
        assert gc["processed"] + gc["ignored_flags"] + gc["ignored_too_short"] == n
        assert gc["histogram"].sum() == gc["processed"]
        assert gc["total_gc_count"] + gc["total_at_count"] + gc["total_other_count"] == 100 * gc["processed"]
        seen, hist, ign, bins = gpu.coverage_sequence(0)
        assert seen and hist.sum() + ign == L + 1
        mapped = n - g["unmapped"]
        assert bins.sum() == 150 * mapped  # every mapped read adds 150 to the depth total
        assert gpu.coverage_nonsensical() == 0
        assert not gpu.coverage_sequence(1)[0]
        first = gpu.results(["chr1", "chr2"])
        gpu.reset()
        gpu.process_batch(db)
        gpu.finalize()
        json_equal(gpu.results(["chr1", "chr2"]), first)  # deterministic


def test_full_size_properties_mixed(gpu_lib):
    """BASELINE configs[4]'s shape on one GPU at its full 100 M records (50-300 bp reads, clips / insertions / deletions / skips,
    ragged columns with offsets): the quality table against the read lengths themselves (row c = the reads longer than c -- the
    16-bit LDS counters of k_qual_ragged and their flushes), the CIGAR tallies against the operations column (the offsets-layout
    k_fields), the depth total against the reference-consuming operations, the depth histogram over L + 1 positions, determinism."""
    n, L = 100_000_000, 248_956_422
    cfg = host.synth_config(n, mode=ffi.SYNTH_MIXED, max_len=300, ref_len=L, n_refs=2)
    with host.QcContext([L, 242_193_529], max_read_len=300, sorted_input=True, lib=gpu_lib) as gpu:
        db = gpu.synth_device_batch(cfg, 0, n)
        gpu.process_batch(db)
        gpu.finalize()
        g = gpu.general()
        assert g["total"] == n and g["primary"] + g["secondary"] + g["supplementary"] == n
        l_seq = gpu.download_column(db, "l_seq", n)
        assert int(l_seq.min()) >= 50 and int(l_seq.max()) == 300
        longer = n - np.cumsum(np.bincount(l_seq, minlength=301))   # longer[c] = reads with more than c bases
        q = gpu.quality_scores()
        rows = q.sum(axis=1)
        assert rows.shape[0] >= 300 and (rows[:300] == longer[:300]).all() and not rows[300:].any()
        assert int(rows.sum()) == db.qual_bytes == int(l_seq.sum(dtype=np.int64))
        del l_seq
        cigar = gpu.download_column(db, "cigar", db.cigar_ops)
        ops, lens = cigar & 15, (cigar >> 4).astype(np.int64)
        per_kind = np.bincount(ops, minlength=9)
        one, two = np.asarray(g["read_one_cigar_ops"]), np.asarray(g["read_two_cigar_ops"])
        assert ((one + two) == per_kind[:9]).all() and int((one + two).sum()) == db.cigar_ops
        consumed = int(lens[(ops == 0) | (ops == 2) | (ops == 3) | (ops == 7) | (ops == 8)].sum())
        del cigar, ops, lens
        h, processed, ignored = gpu.template_length()
        assert processed + ignored == n and h.sum() == processed
        gc = gpu.gc_content()
        assert gc["processed"] + gc["ignored_flags"] + gc["ignored_too_short"] == n and gc["ignored_too_short"] > 0
        assert gc["histogram"].sum() == gc["processed"]
        assert gc["total_gc_count"] + gc["total_at_count"] + gc["total_other_count"] == 100 * gc["processed"]
        seen, hist, ign, bins = gpu.coverage_sequence(0)
        assert seen and hist.sum() + ign == L + 1
        assert int(bins.sum()) == consumed      # every M / D / N / = / X base of a placed read is one unit of depth (coverage.rs:159-176)
        assert gpu.coverage_nonsensical() == 0 and not gpu.coverage_sequence(1)[0]
        first = gpu.results(["chr1", "chr2"])
        gpu.reset()
        gpu.process_batch(db)
        gpu.finalize()
        json_equal(gpu.results(["chr1", "chr2"]), first)


# ---------------------------------------------------------------------------------------------
# N > 1 with REAL contexts: three processes share the one GPU of the box and run ngsq_exchange (the C++
# protocol of ngs_amd/csrc/exchange.cpp) over a host transport -- the library's shared-memory transport, or
# callbacks over torch.distributed/gloo; the device blocks are staged through the host.  RCCL refuses two
# ranks on one device: its entry points are exercised with one rank below, and the library's RCCL transport itself
# (no staging: collectives and grouped sends / receives on device buffers) with three ranks over tests/rccl_double.
# ---------------------------------------------------------------------------------------------
def _rank_worker(rank, world, port, q, mode, transport):
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from ngs_amd import ffi as F, host as H, shard
        from oracle import oracle_py
        from tests.test_shard_gloo import _make_comm
        from tests.util import json_equal as jeq

        comm, done = _make_comm(transport, rank, world, port)
        lib = F.load_library()
        # (eight ranks -- the node this is for: 200 k records and 200 Coverage chunks per rank)
        n, L = (90_000, 700_000) if world <= 3 else (200_000 * world, 800_000 * world)
        ref_len = [L, 50_000]
        stream = "stream" in mode   # sorted_input contexts: Coverage streamed, only the seams are exchanged
        scfg = H.synth_config(n, mode=F.SYNTH_MIXED if "mixed" in mode else F.SYNTH_FIXED, ref_len=L, n_refs=2)
        whole = H.synth_host_batch(scfg, 0, n, lib)
        first, cnt = shard.shard_range(n, rank, world)
        kw = dict(facets=F.FACETS_DEFAULT, bin_size=50_000, max_read_len=300, gc_seed=5)
        ref_bases = None
        if mode == "edits":   # Edits: refs/alts all-reduced, VAF split over the ranks
            rng = np.random.default_rng(77)
            ref_bases = [rng.choice(np.array([1, 2, 4, 8], dtype=np.uint8), size=x) for x in ref_len]
            kw["facets"] = F.FACETS_DEFAULT | F.FACET_EDITS
            scfg = H.synth_config(n, mode=F.SYNTH_FIXED, ref_len=L, n_refs=2)
        ctx = H.QcContext(ref_len, device=0, lib=lib, sorted_input=stream, ref_bases=ref_bases,
                          cov_head_guard=8192 if stream and rank else 0, **kw)
        names = ["chr1", "chr2"]
        if mode == "stream-overlap":  # every rank scans the same records: the exchange must refuse, on every rank
            ctx.process_batch(whole.slice(0, cnt))
            try:
                comm.exchange(ctx)
                q.put((rank, "FAIL overlapping sorted_input shards were accepted"))
            except shard.CommError as e:
                q.put((rank, "ok" if e.code == F.ERR_UNSORTED and "overlap" in str(e) else "FAIL " + str(e)))
            ctx.close()
            done()
            return
        want = None
        if rank == 0:
            orc = oracle_py.Oracle(ref_len, ref_bases=ref_bases, **kw)
            orc.process_batch(whole)
            orc.finalize()
            want = orc.results(names)
        for step in range(2):  # the second pass checks reset after a partial teardown
            ctx.process_batch(whole.slice(first, first + cnt))
            rep = comm.exchange(ctx)
            assert rep["mode"] == "owner", rep
            if stream:
                assert int(ctx.state_download(4).sum()) > 30   # most of this shard's chunks never touched the array
            ctx.finalize()
            got = ctx.results(names)
            if rank == 0:
                jeq(got, want)
            import hashlib
            import json as _json
            digest = np.frombuffer(hashlib.sha256(_json.dumps(got, sort_keys=True).encode()).digest(), dtype=np.uint64)
            assert (comm.allgather(digest) == digest).all()  # every rank holds the whole-file result
            ctx.reset()
        ctx.close()
        done()
        q.put((rank, "ok"))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, "FAIL " + traceback.format_exc()))


@pytest.mark.parametrize("mode,transport", [("fixed", "shm"), ("mixed", "gloo"), ("fixed-stream", "gloo"), ("mixed-stream", "shm"),
                                            ("stream-overlap", "shm"), ("edits", "shm"),
                                            # the RCCL transport's own code path (device buffers, no host staging) with three
                                            # ranks, librccl replaced by tests/rccl_double
                                            ("fixed", "rccl-double"), ("mixed-stream", "rccl-double"), ("edits", "rccl-double"),
                                            ("stream-overlap", "rccl-double")])
def test_three_ranks_owner_computes_teardown(gpu_lib, oracle_mod, mode, transport):
    from tests.test_shard_gloo import _run_ranks
    _run_ranks(_rank_worker, 3, mode, transport)


@pytest.mark.parametrize("mode,transport", [("fixed", "shm"), ("mixed-stream", "shm"), ("edits", "rccl-double"), ("fixed-stream", "rccl-double"),
                                            ("stream-overlap", "rccl-double")])
def test_eight_ranks_owner_computes_teardown(gpu_lib, oracle_mod, mode, transport):
    """The world the library is for -- eight ranks, here sharing this box's one GPU (no more than three had ever
    run together): the exchange plan cut eight ways, halos between seven seams, the VAF teardown split in eight, over the
    shared-memory transport and over the RCCL transport's own code path (device buffers; librccl replaced by tests/rccl_double)."""
    from tests.test_shard_gloo import _run_ranks
    _run_ranks(_rank_worker, 8, mode, transport)


def test_rccl_transport_host_collectives_three_ranks(gpu_lib):
    """RcclComm's host-buffer collectives (own stream + device scratch) and multi-round sendrecv with three ranks,
    librccl replaced by tests/rccl_double: the transport checks of tests/test_shard_gloo.py, on the GPU."""
    from tests.test_shard_gloo import _run_ranks, _transport_worker
    _run_ranks(_transport_worker, 3, "rccl-double")


def test_rccl_entry_points_with_one_rank(gpu_lib, oracle_mod):
    """ncclGetUniqueId / ncclCommInitRank / the collectives of one exchange on the context's stream, world = 1
    (a one-GPU box cannot hold more RCCL ranks): the result must equal the plain finalize."""
    from ngs_amd import shard
    n, L = 60_000, 500_000
    ref_len = [L, 40_000]
    scfg = host.synth_config(n, mode=ffi.SYNTH_MIXED, ref_len=L, n_refs=2)
    hb = host.synth_host_batch(scfg, 0, n, gpu_lib)
    comm = shard.Comm.rccl(0, 1, shard.unique_id(gpu_lib), 0, gpu_lib)
    assert comm.kind == "rccl" and comm.world == 1
    assert comm.allgather_ints([5, 6]) == [[5, 6]]
    assert (comm.allreduce(np.arange(10, dtype=np.uint64)) == np.arange(10, dtype=np.uint64)).all()
    comm.barrier()
    kw = dict(bin_size=50_000, max_read_len=300, gc_seed=5)
    for sorted_input in (False, True):
        with host.QcContext(ref_len, lib=gpu_lib, sorted_input=sorted_input, **kw) as plain, \
                host.QcContext(ref_len, lib=gpu_lib, sorted_input=sorted_input, **kw) as ex:
            plain.process_batch(hb)
            plain.finalize()
            for _ in range(2):
                ex.process_batch(hb)
                rep = comm.exchange(ex)
                assert rep["mode"] == "owner" and rep["halo_bytes"] == 0 and rep["host_syncs"] <= 3, rep
                ex.finalize()
                json_equal(ex.results(["a", "b"]), plain.results(["a", "b"]))
                ex.reset()
    comm.destroy()


# ---- Genomic Features facet (features.rs) ------------------------------------------------------
def random_gene_model(rng, ref_len, n, max_span=4000):
    """(ref_id, name id, start, end) of n random GFF features incl. 1-bp ones (start == end)."""
    ref = rng.integers(0, len(ref_len), n).astype(np.uint32)
    start = np.array([rng.integers(1, ref_len[r] + 1) for r in ref], dtype=np.uint32)
    span = np.where(rng.random(n) < 0.1, 0, rng.integers(0, max_span, n)).astype(np.uint32)
    name = rng.choice(5, n, p=[.1, .1, .25, .4, .15]).astype(np.uint32)
    return ref, name, start, start + span


@pytest.mark.parametrize("roles", [(0, 1, 2, 3, 4), (0, 0, 2, 3, 4), (0, 0, 0, 3, 4), (0, 1, 2, 3, 3), (0, 1, 2, 2, 4),
                                   (0, 1, 2, 3, 0)])
def test_genomic_features_matches_oracle(gpu_lib, oracle_mod, roles):
    """Random gene models and records; feature roles that share a name (the reference compares name
    strings) exercise the order-dependent if/else-if chains of features.rs:186-238."""
    rng = np.random.default_rng(sum(roles) + 31)
    ref_len = [60_000, 9_000, 25_000]
    primary = [1, 0, 1]
    ref, name, start, stop = random_gene_model(rng, ref_len, 600)
    name = np.array([roles[k] for k in name], dtype=np.uint32)  # intervals carry name ids that are in use
    hb = random_batch(rng, 30_000, ref_len, weird=True)
    hb.cols["ref_id"][hb.cols["ref_id"] < 0] = 0  # no mapped record without a sequence here (tested below)
    hb.cols["pos"][hb.cols["pos"] < 0] = 5
    facets = ffi.FACET_FEATURES | ffi.FACET_GENERAL
    gpu, orc = run_both(oracle_mod, gpu_lib, [hb.slice(0, 11_111), hb.slice(11_111, hb.n)], ref_len, primary,
                        facets=facets, features=(ref, name, start, stop, roles))
    f = gpu.features()
    assert f["processed"] + f["ignored_flags"] + f["ignored_nonprimary_chromosome"] == hb.n
    assert f["intergenic_count"] + f["exonic_count"] + f["intronic_count"] == f["processed"]
    assert f["processed"] > 10_000 and f["utr_five_prime_count"] > 0
    # a gene / exon name equal to a UTR/CDS name lands in the other store, and an exon name equal to the gene
    # name never reaches the `has_exon` branch
    gene_ok = roles[4] not in roles[:3]
    assert (f["exonic_count"] > 0) == (gene_ok and roles[3] not in (roles[0], roles[1], roles[2], roles[4]))
    assert (f["intergenic_count"] == f["processed"]) == (not gene_ok)
    assert gpu.results(["a", "b", "c"])["features"]["records"]["processed"] == f["processed"]


def test_genomic_features_errors_and_state(gpu_lib, oracle_mod):
    ref_len = [5000]
    feats = (np.array([0]), np.array([4]), np.array([100]), np.array([900]), (0, 1, 2, 3, 4))
    recs = [dict(flag=0, mapq=9, ref_id=0, pos=150, cigar="50M", seq="A" * 50, qual=[30] * 50),
            dict(flag=0, mapq=9, ref_id=-1, pos=150, cigar="50M", seq="A" * 50, qual=[30] * 50),   # features.rs:132-140
            dict(flag=0, mapq=9, ref_id=0, pos=-1, cigar="50M", seq="A" * 50, qual=[30] * 50)]     # features.rs:171-174
    gpu, orc = run_both(oracle_mod, gpu_lib, [batch_from_records(recs)], ref_len, facets=ffi.FACET_FEATURES,
                        features=feats)
    e = gpu.error_counts()
    assert e["features_missing_reference_id"] == 1 and e["features_missing_position"] == 1
    assert gpu.features()["intronic_count"] == 1
    # the facet cannot FINISH without a gene model (batches may come before it: test_gene_model_after_the_first_batches)
    q = host.QcContext(ref_len, facets=ffi.FACET_FEATURES, lib=gpu_lib)
    q.process_batch(batch_from_records(recs[:1]))
    with pytest.raises(Exception, match="ngsq_set_features"):
        q.finalize()
    q.reset()                 # (drops what was kept of the batch)
    assert q.finalize() == 0 and q.features()["processed"] == 0
    q.close()


@pytest.mark.parametrize("layout", ["offsets", "rows"])
def test_gene_model_after_the_first_batches(gpu_lib, oracle_mod, layout):
    """Batches that reach the context BEFORE ngsq_set_features (the host is still reading the GFF: a gzip stream inflates on one
    thread for seconds) keep what Genomic Features needs of their records -- flag, sequence, position, span: 16 bytes each -- and are
    looked up when the model arrives; batches behind it go the usual way.  Same tallies and errors as the oracle's, every facet
    beside it untouched; a model that arrives twice (ngsq_set_features again) replaces the tables for what follows."""
    from tests.util import to_fixed_stride
    rng = np.random.default_rng(404)
    ref_len = [60_000, 9_000, 25_000]
    primary = [1, 0, 1]
    ref, name, start, stop = random_gene_model(rng, ref_len, 600)
    hb = random_batch(rng, 30_000, ref_len, weird=True, max_len=120 if layout == "rows" else 300)
    if layout == "rows":
        hb = to_fixed_stride(hb)
    facets = ffi.FACETS_DEFAULT | ffi.FACET_FEATURES
    kw = dict(facets=facets, bin_size=1000, max_read_len=320, gc_seed=7)
    orc = oracle_mod.Oracle(ref_len, primary, **kw)
    orc.set_features(ref, name, start, stop)
    orc.process_batch(hb)
    rc = orc.finalize(allow_malformed=True)
    with host.QcContext(ref_len, primary, lib=gpu_lib, **kw) as gpu:
        gpu.process_batch(hb.slice(0, 7_001))
        gpu.process_batch(gpu.upload(hb.slice(7_001, 19_000)))
        gpu.set_features(ref, name, start, stop)        # the model arrives: the 19 000 records kept so far are looked up
        gpu.process_batch(hb.slice(19_000, hb.n))
        assert gpu.finalize(allow_malformed=True) == rc
        compare_contexts(gpu, orc, 3, facets, 1000, ref_len)
        json_equal(gpu.results(["a", "b", "c"]), orc.results(["a", "b", "c"]))
        f = gpu.features()
        assert f["processed"] + f["ignored_flags"] + f["ignored_nonprimary_chromosome"] + \
            gpu.error_counts()["features_missing_reference_id"] + gpu.error_counts()["features_missing_position"] == hb.n


def test_quality_counters_of_the_offsets_layout_do_not_wrap(gpu_lib):
    """k_qual_ragged keeps 16-bit counters in LDS and flushes them every 63 x 1024 records of a block (qual_kernel.hip): 40 M
    records that all add to the SAME cells -- 78 k per block of the grid's 512 -- come out exact only if the flushes happen."""
    n, l = 40_000_000, 16
    cols = {"flag": np.zeros(n, np.uint16), "mapq": np.full(n, 30, np.uint8), "ref_id": np.zeros(n, np.int32),
            "pos": np.zeros(n, np.int32), "mate_ref_id": np.full(n, -1, np.int32), "tlen": np.zeros(n, np.int32),
            "l_seq": np.full(n, l, np.uint32), "n_cigar": np.ones(n, np.uint16),
            "seq": np.full(n * (l // 2), 0x12, np.uint8), "seq_off": None,
            "qual": np.full(n * l, 30, np.uint8), "qual_off": np.arange(n + 1, dtype=np.uint64) * l,
            "cigar": np.full(n, l << 4, np.uint32), "cigar_off": None, "record_id": None}
    cols["qual"][7::16] = 70     # a score beyond the LDS table's 64 rows: straight to the global counters
    hb = host.HostBatch(n, cols, l // 2, 0, 1, 0)
    with host.QcContext([1000], facets=ffi.FACET_QUALITY_SCORE, max_read_len=64, lib=gpu_lib) as gpu:
        gpu.process_batch(hb)
        gpu.finalize()
        q = gpu.quality_scores()
        for c in range(l):
            want = 70 if c == 7 else 30
            assert int(q[c][want]) == n and int(q[c].sum()) == n, c
        assert int(q[l:].sum()) == 0
