"""Streaming Coverage (ngsq_config.sorted_input, cov_stream.hip) against the oracle: coordinate-sorted
inputs must give the same Results as the difference-array path, bit for bit, wherever the batch, tile,
chunk and sequence boundaries fall; unsorted input must be refused."""
import numpy as np
import pytest

from ngs_amd import ffi, host
from tests.util import batch_from_records, compare_contexts, coordinate_sorted, json_equal, random_batch, take_records

pytestmark = pytest.mark.gpu


def run_sorted(oracle_mod, lib, batches, ref_len, primary=None, facets=ffi.FACETS_DEFAULT, bin_size=1000,
               max_read_len=320, on_device=False, guard=0, cov_cap=0, passes=1):
    kw = dict(facets=facets, bin_size=bin_size, max_read_len=max_read_len, gc_seed=7, cov_cap=cov_cap)
    orc = oracle_mod.Oracle(ref_len, primary, **kw)
    for hb in batches:
        orc.process_batch(hb)
    rc_o = orc.finalize(allow_malformed=True)
    names = [f"chr{i + 1}" for i in range(len(ref_len))]
    with host.QcContext(ref_len, primary, lib=lib, sorted_input=True, cov_head_guard=guard, **kw) as gpu:
        for p in range(passes):
            if p:
                gpu.reset()
            for hb in batches:
                gpu.process_batch(gpu.upload(hb) if on_device else hb)
            assert gpu.finalize(allow_malformed=True) == rc_o
            compare_contexts(gpu, orc, len(ref_len), facets, bin_size, ref_len)
            json_equal(gpu.results(names), orc.results(names))
        flags = gpu.state_download(4)
    return flags


def split(hb, cuts):
    out, lo = [], 0
    for c in list(cuts) + [hb.n]:
        out.append(take_records(hb, np.arange(lo, c)))
        out[-1].first_record_index = lo
        lo = c
    return out


@pytest.mark.parametrize("on_device", [False, True])
def test_synthetic_fixed_150bp_streams(gpu_lib, oracle_mod, on_device):
    n, L = 200_000, 500_000
    cfg = host.synth_config(n, ref_len=L, n_refs=2)
    hb = host.synth_host_batch(cfg, 0, n, gpu_lib)
    flags = run_sorted(oracle_mod, gpu_lib, [hb], [L, 300_000], bin_size=50_000, max_read_len=150,
                       on_device=on_device, passes=2)
    assert flags.sum() >= (L // 4096) - 3          # nearly every chunk of chr1 was finished while streaming


def test_synthetic_mixed_streams_in_batches(gpu_lib, oracle_mod):
    n, L = 120_000, 400_000
    cfg = host.synth_config(n, mode=ffi.SYNTH_MIXED, ref_len=L, n_refs=2)
    hbs = [host.synth_host_batch(cfg, lo, hi - lo, gpu_lib) for lo, hi in ((0, 50_001), (50_001, 50_300), (50_300, n))]
    flags = run_sorted(oracle_mod, gpu_lib, hbs, [L, 1000], bin_size=50_000, max_read_len=300)
    assert flags.sum() > 60


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_sorted_random_edge_cases(gpu_lib, oracle_mod, seed):
    """Every flag bit / CIGAR kind / unplaced / beyond-the-end record of random_batch, in coordinate order,
    over several sequences (one not covered, one of a single base), cut into batches at odd places."""
    rng = np.random.default_rng(seed)
    ref_len = [60_000, 1, 777, 30_000, 9_000]
    primary = [1, 1, 0, 1, 1]
    hb = coordinate_sorted(random_batch(rng, 30_000, ref_len))
    cuts = [] if seed == 1 else sorted(rng.integers(1, hb.n, 3).tolist()) if seed < 4 else [256, 512, 513, 20_000]
    flags = run_sorted(oracle_mod, gpu_lib, split(hb, cuts), ref_len, primary, bin_size=[1000, 7, 50_000, 4096][seed - 1],
                       on_device=(seed % 2 == 0), passes=2 if seed == 2 else 1)
    assert flags.sum() >= 4


def test_gaps_sparse_and_spliced(gpu_lib, oracle_mod):
    """Clusters with an empty stretch between them (closed-form zero run), sparse reads (several windows per
    tile), long N skips that cross many tiles and windows, a pile deeper than cov_cap."""
    rng = np.random.default_rng(11)
    L = 3_000_000
    recs = []
    def add(pos, cigar, k=1):
        for _ in range(k):
            recs.append(dict(flag=0, ref_id=0, pos=int(pos), cigar=cigar, seq="ACGT", qual=[30] * 4))
    for p in np.sort(rng.integers(5_000, 400_000, 6000)):        # dense cluster
        add(p, rng.choice(["100M", "40M2D60M", "10S90M", "50M300N50M"]))
    add(200_000, "50M20000N50M", 3)                                 # reaches across ~20 tiles
    add(200_010, "10M1500000N10M")                                  # and one across the empty stretch
    for p in np.sort(rng.integers(1_000_000, 2_500_000, 900)):    # sparse: ~1600 positions between reads
        add(p, "75M")
    add(2_600_000, "30M", 150)                                      # deeper than cov_cap = 100
    for p in np.sort(rng.integers(2_600_000, 2_990_000, 3000)):
        add(p, "120M")
    hb = coordinate_sorted(batch_from_records(recs))
    for cuts in ([], [3000, 7000]):
        flags = run_sorted(oracle_mod, gpu_lib, split(hb, cuts), [L], facets=ffi.FACET_COVERAGE | ffi.FACET_GENERAL,
                           bin_size=50_000, cov_cap=100)
        assert flags.sum() > 250


def test_head_guard_keeps_the_first_positions_on_the_array(gpu_lib, oracle_mod):
    n, L = 100_000, 400_000
    cfg = host.synth_config(n, ref_len=L, n_refs=1)
    hb = host.synth_host_batch(cfg, 0, n, gpu_lib)
    f0 = run_sorted(oracle_mod, gpu_lib, [hb], [L], bin_size=50_000, max_read_len=150)
    f1 = run_sorted(oracle_mod, gpu_lib, [hb], [L], bin_size=50_000, max_read_len=150, guard=100_000)
    assert f0[:20].sum() >= 18 and f1[:24].sum() == 0 and f1[26:].sum() == f0[26:].sum()
    # the guard holds in whichever batch the first positions come: a first batch too small to stream anything
    # (it used to be consumed there) and the rest in further batches
    cuts = [0, 40, 300, 50_000, hb.n]
    parts = [hb.slice(a, b) for a, b in zip(cuts[:-1], cuts[1:])]
    f2 = run_sorted(oracle_mod, gpu_lib, parts, [L], bin_size=50_000, max_read_len=150, guard=100_000)
    assert f2[:24].sum() == 0 and f2[27:].sum() >= f0[27:].sum() - 3


def test_unsorted_input_is_refused(gpu_lib):
    rng = np.random.default_rng(5)
    ref_len = [50_000, 20_000]
    hb = coordinate_sorted(random_batch(rng, 5000, ref_len, weird=False))
    swapped = np.arange(hb.n)
    swapped[[1000, 3000]] = [3000, 1000]
    for batches in ([take_records(hb, swapped)],                                   # inside a batch
                    [take_records(hb, np.arange(2500, 5000)), take_records(hb, np.arange(0, 2500))]):  # across batches
        with host.QcContext(ref_len, lib=gpu_lib, sorted_input=True, max_read_len=320) as gpu:
            for b in batches:
                gpu.process_batch(b)
            with pytest.raises(host.NgsqError) as ei:
                gpu.finalize()
            assert ei.value.code == ffi.ERR_UNSORTED and "coordinate order" in ei.value.message
            gpu.reset()                       # and the context is usable again
            gpu.process_batch(hb)
            gpu.finalize(allow_malformed=True)


def test_full_size_stream_equals_classic(gpu_lib):
    """BASELINE configs[2] shape at 20 M records on a chr1-sized axis: the two Coverage paths agree (no oracle
    at this size), twice through reset."""
    n, L = 20_000_000, 248_956_422
    cfg = host.synth_config(n, ref_len=L, n_refs=2)
    res = []
    for sorted_input in (False, True):
        with host.QcContext([L, 1000], lib=gpu_lib, max_read_len=150, gc_seed=3, sorted_input=sorted_input) as gpu:
            db = gpu.synth_device_batch(cfg, 0, n)
            for _ in range(2):
                gpu.reset()
                gpu.process_batch(db)
                gpu.finalize()
            res.append(gpu.results(["chr1", "chr2"]))
            gpu.free_batch(db)
    json_equal(res[0], res[1])


def test_end_of_an_earlier_record_exactly_on_a_window_boundary(gpu_lib, oracle_mod):
    """A record of tile 0 ends exactly 2048 positions (two LDS windows) into the range tile 1 owns, while a long
    record of tile 1 keeps the windows from being skipped: its -1 is the first entry of the third window."""
    recs = []
    def add(pos, cigar):
        recs.append(dict(flag=0, ref_id=0, pos=pos, cigar=cigar, seq="ACGT", qual=[30] * 4))
    S = 20_001                                   # 1-based start of tile 1's first record
    for i in range(256):
        pos0 = 8192 + i * 10
        add(pos0, f"10M{S + 2048 - (pos0 + 1) - 20}N10M" if i == 100 else "50M")
    add(S - 1, "10M5000N10M")
    for i in range(255):
        add(30_000 + i * 40, "50M")
    for i in range(300):
        add(60_000 + i * 30, "75M")
    hb = batch_from_records(recs)
    for k in (0, 1, 1023, 1024, 1025):           # and the same with the end moved around the boundary
        r2 = list(recs)
        pos0 = 8192 + 100 * 10
        r2[100] = dict(recs[100], cigar=f"10M{S + 2048 + k - 1024 - (pos0 + 1) - 20}N10M")
        run_sorted(oracle_mod, gpu_lib, [batch_from_records(r2)], [2_000_000], facets=ffi.FACET_COVERAGE, bin_size=64)
    flags = run_sorted(oracle_mod, gpu_lib, [hb], [2_000_000], facets=ffi.FACET_COVERAGE, bin_size=64)
    assert flags.sum() >= 10


@pytest.mark.parametrize("n,batch", [(1_000_000, None), (3_000_000, 1 << 20)])
def test_sparse_spliced_chr1_stream_equals_array(gpu_lib, n, batch):
    """1-4x coverage of a chr1-sized axis with skips up to 5 kb: a tile of 256 reads owns ~60 LDS windows, ends of
    earlier tiles land anywhere in them (this shape caught a lost -1 on a window boundary).  No oracle at this
    size: the two Coverage paths must agree."""
    L = 248_956_422
    cfg = host.synth_config(n, mode=ffi.SYNTH_MIXED, ref_len=L, n_refs=2)
    res = []
    for sorted_input in (False, True):
        with host.QcContext([L, 1000], lib=gpu_lib, max_read_len=300, gc_seed=3, sorted_input=sorted_input,
                            facets=ffi.FACET_COVERAGE | ffi.FACET_GENERAL) as gpu:
            step = batch or n
            for lo in range(0, n, step):
                db = gpu.synth_device_batch(cfg, lo, min(step, n - lo))
                gpu.process_batch(db)
            gpu.finalize()
            res.append(gpu.results(["chr1", "chr2"]))
    json_equal(res[0], res[1])
