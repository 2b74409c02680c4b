#!/usr/bin/env python3
"""Randomised parity sweep (GPU): many seeds / shapes through the facet kernels against the oracle, and
through file -> device ingest -> kernels against file -> host ingest -> kernels.  Not part of pytest:
    python tools/fuzz_parity.py [--seeds 40]
"""
import argparse
import ctypes as C
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from ngs_amd import ffi, host  # noqa: E402
from oracle import oracle_py  # noqa: E402
from tests import bamio  # noqa: E402
from tests.util import (batch_from_records, compare_contexts, coordinate_sorted, json_equal, random_batch,  # noqa: E402
                        take_records, to_fixed_stride)


def sorted_sweep(lib, seeds, base=0):
    """Streaming Coverage (sorted_input) against the oracle: random densities (sparse to piles deeper than the
    open-ends list and than cov_cap), span mixes with long skips, several sequences, batches cut anywhere."""
    for seed in range(base, base + seeds):
        rng = np.random.default_rng(5000 + seed)
        n_refs = int(rng.integers(1, 4))
        # one seed in three on long sequences: sparse reads, a tile of 256 then owns dozens of LDS windows
        ref_len = [int(rng.integers(5_000, 600_000 if seed % 3 else 30_000_000)) for _ in range(n_refs)]
        primary = [int(rng.random() < 0.85) for _ in range(n_refs)]
        n = int(rng.integers(300, 60_000))
        recs = []
        long_p = float(rng.choice([0.0, 0.02, 0.3]))
        for _ in range(n):
            r = int(rng.integers(0, n_refs))
            L = ref_len[r]
            if rng.random() < 0.3:   # clusters: piles of a few hundred reads on a few hundred positions
                c = int(rng.integers(0, 8)) * (L // 8)
                pos = c + int(rng.integers(0, 300))
            else:
                pos = int(rng.integers(0, L + 20))
            k = rng.random()
            if k < long_p:
                cig = f"{int(rng.integers(1, 60))}M{int(rng.integers(100, 40_000))}N{int(rng.integers(1, 60))}M"
            elif k < 0.9:
                cig = f"{int(rng.integers(1, 300))}M"
            elif k < 0.95:
                cig = f"{int(rng.integers(1, 30))}S{int(rng.integers(1, 100))}M{int(rng.integers(1, 9))}D{int(rng.integers(1, 50))}M"
            else:
                cig = "*"
            recs.append(dict(flag=int(rng.integers(0, 4096)), ref_id=r if rng.random() > 0.01 else -1, pos=pos,
                             mate_ref_id=r, cigar=cig, seq="ACGT", qual=[20] * 4))
        hb = coordinate_sorted(batch_from_records(recs))
        kw = dict(facets=ffi.FACET_COVERAGE | ffi.FACET_GENERAL, bin_size=int(rng.choice([7, 1000, 50_000])),
                  max_read_len=320, gc_seed=seed, cov_cap=int(rng.choice([0, 64])))
        orc = oracle_py.Oracle(ref_len, primary, **kw)
        gpu = host.QcContext(ref_len, primary, lib=lib, sorted_input=True,
                             cov_head_guard=int(rng.choice([0, 0, 5000])), **kw)
        cuts = sorted(set([0, hb.n] + [int(x) for x in rng.integers(0, hb.n + 1, int(rng.integers(0, 4)))]))
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            part = take_records(hb, np.arange(lo, hi))
            orc.process_batch(part)
            gpu.process_batch(gpu.upload(part) if rng.random() < 0.5 else part)
        assert orc.finalize(allow_malformed=True) == gpu.finalize(allow_malformed=True)
        compare_contexts(gpu, orc, n_refs, kw["facets"], kw["bin_size"], ref_len)
        names = [f"s{i}" for i in range(n_refs)]
        json_equal(gpu.results(names), orc.results(names))
        flagged = int(gpu.state_download(4).sum())
        gpu.close()
        print(f"sorted seed {seed}: n={hb.n} refs={ref_len} long={long_p} batches={len(cuts) - 1} streamed_chunks={flagged} ok", flush=True)
    print("sorted sweep ok")


def extra_sweep(lib, seeds, base=0):
    """Edits (LDS window per tile of sorted reads, eight bases per step) and Genomic Features (bracketed searches per
    tile) against the oracle: sorted and unsorted batches, long skips that leave the window, several sequences with
    and without reference bases / intervals, roles that share names."""
    for seed in range(base, base + seeds):
        rng = np.random.default_rng(9000 + seed)
        n_refs = int(rng.integers(1, 4))
        ref_len = [int(rng.integers(300, 60_000)) for _ in range(n_refs)]
        primary = [int(rng.random() < 0.8) for _ in range(n_refs)]
        bases = [rng.choice(np.array([1, 2, 4, 8, 15], dtype=np.uint8), size=L, p=[.24, .24, .24, .24, .04]) for L in ref_len]
        n = int(rng.integers(1, 30_000))
        max_len = int(rng.choice([36, 100, 150, 250]))
        hb = random_batch(rng, n, ref_len, max_len=max_len, min_len=int(rng.integers(0, max_len + 1)), weird=bool(rng.integers(0, 2)))
        if rng.random() < 0.7:
            hb = coordinate_sorted(hb)
        # round 5: half of the short-read seeds as fixed-pitch rows (k_edits_rows, which then also tallies GC Content: the facet is on
        # below), the others through offsets (k_edits with its own second-segment step)
        rows = max_len <= 160 and rng.random() < 0.5
        if rows:
            hb = to_fixed_stride(hb)
        m = int(rng.integers(0, 3000))
        fr = rng.integers(0, n_refs, m).astype(np.uint32)
        fs = np.array([rng.integers(1, ref_len[r] + 1) for r in fr], dtype=np.uint32)
        fe = fs + np.where(rng.random(m) < 0.1, 0, rng.integers(0, 5000, m)).astype(np.uint32)
        fn = rng.choice(5, m).astype(np.uint32)
        roles = tuple(int(x) for x in rng.choice([0, 1, 2, 3, 4], 5)) if rng.random() < 0.3 else (0, 1, 2, 3, 4)
        kw = dict(facets=ffi.FACET_EDITS | ffi.FACET_FEATURES | ffi.FACET_GENERAL | ffi.FACET_GC_CONTENT, max_read_len=320, gc_seed=seed, ref_bases=bases)
        orc = oracle_py.Oracle(ref_len, primary, **kw)
        gpu = host.QcContext(ref_len, primary, lib=lib, **kw)
        orc.set_features(fr, fn, fs, fe, roles)
        gpu.set_features(fr, fn, fs, fe, roles)
        cuts = sorted(set([0, n] + [int(x) for x in rng.integers(0, n + 1, 3)]))
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            part = hb.slice(lo, hi) if hb.cols.get("seq_off") is None else take_records(hb, np.arange(lo, hi))
            part.first_record_index = lo
            orc.process_batch(part)
            gpu.process_batch(gpu.upload(part) if rng.random() < 0.5 else part)
        assert orc.finalize(allow_malformed=True) == gpu.finalize(allow_malformed=True)
        for a, b in zip(orc.edits(), gpu.edits()):
            assert (a == b).all()
        assert orc.features() == gpu.features() and orc.error_counts() == gpu.error_counts()
        go, gg = orc.gc_content(), gpu.gc_content()
        assert all((go[k] == gg[k]).all() if k == "histogram" else go[k] == gg[k] for k in go), (go, gg)
        names = [f"s{i}" for i in range(n_refs)]
        if not any(orc.error_counts().values()):
            json_equal(gpu.results(names), orc.results(names))
        gpu.close()
        print(f"extra seed {seed}: n={n} refs={n_refs} intervals={m} roles={roles} {'rows' if rows else 'offsets'} ok", flush=True)
    print("extra sweep ok")


def ingest_sweep(lib, seeds, base=0):
    """Device reader against the host reader, batch for batch and byte for byte, on shapes that stress the record index
    and the column kernels: reads from 1 base to 100 kb (records longer than a 4 KiB piece and than a 64 KiB segment),
    BGZF blocks of 0.7-60 kB, ingest chunks of 1 MiB to 1 GiB (cut records carried over), batches of 257 records to
    everything, the block cache on (buffers of the previous seed recycled)."""
    from tests.test_bam_ingest import dressed, read_all, records_of
    from tests.test_device_ingest_gpu import read_all_device, same_batches
    td = tempfile.mkdtemp(prefix="ngsq_fuzz_ingest_")
    ctx = host.QcContext([100_000, 50_000], [1, 1], lib=lib)
    for seed in range(base, base + seeds):
        rng = np.random.default_rng(9000 + seed)
        max_len = int(rng.choice([1, 3, 36, 150, 151, 300, 1000, 5000, 20_000, 100_000]))
        n = int(rng.integers(1, max(2, min(30_000, 6_000_000 // max(max_len, 40)))))
        ref_len = [100_000, 50_000]
        uniform = rng.random() < 0.4
        hb = random_batch(rng, n, ref_len, max_len=max_len, min_len=max_len if uniform else int(rng.integers(0, max_len + 1)),
                          weird=bool(rng.integers(0, 2)))
        p = os.path.join(td, "f.bam")
        # two seeds in three: read names and auxiliary data as an aligner writes them; one in three: with payloads that read as
        # chains of BAM records (tests/bamio.py fake_record_chain, a Z / B value that is a 36-byte record head)
        style = int(rng.integers(0, 3))
        names, aux = dressed(rng, hb, 2, style == 2) if style and n <= 8000 else (None, None)
        bamio.write_bam(p, hb, ["chr1", "chr2"], ref_len, block_payload=int(rng.choice([700, 4000, 60000])), names=names, aux=aux)
        os.environ["NGSQ_INGEST_RAW_MB"] = str(int(rng.choice([1, 4, 1024])))
        max_records = int(rng.choice([257, 2500, 1 << 20]))
        _, hbatches, hn = read_all(lib, p, max_records)
        dbatches, dn = read_all_device(lib, ctx, p, max_records)
        assert dn == hn == hb.n, (dn, hn, hb.n)
        if os.environ["NGSQ_INGEST_RAW_MB"] == "1024":
            same_batches(dbatches, hbatches)        # one chunk: the two readers cut the same batches
        else:                                       # a batch also ends where an ingest chunk ends: compare the records
            assert [r for b in dbatches for r in records_of(b)] == [r for b in hbatches for r in records_of(b)] == records_of(hb)
        st = read_all_device.last_stats
        print(f"ingest seed {seed}: n={n} max_len={max_len} uniform={uniform} chunk={os.environ['NGSQ_INGEST_RAW_MB']} MiB "
              f"batch={max_records} aux={('none', 'aligner', 'adversarial')[style] if names else 'none'} "
              f"walk_one={st.get('walk_one')}/{st.get('segments')} segments in {st.get('chunks')} chunks ok", flush=True)
    os.environ.pop("NGSQ_INGEST_RAW_MB", None)
    ctx.close()
    print("ingest sweep ok")


def genome_sweep(lib, seeds, base=0):
    """All seven facets on headers of up to 200 sequences (no seed had more than four): the 195 @SQ lines of the
    GRCh38 no-alt analysis set with its chromosomes scaled down by a random factor, or a random header of 5-200 sequences of
    40 bp to 3 Mbp; records on a random subset of the sequences (empty ones in between), raw or clean, sorted (then also
    streamed Coverage) or not, batches cut anywhere."""
    from tests import genome_util as gu
    from tests.util import make_edit_friendly
    facets = ffi.FACETS_DEFAULT | ffi.FACET_EDITS | ffi.FACET_FEATURES
    for seed in range(base, base + seeds):
        rng = np.random.default_rng(12000 + seed)
        if seed % 2 == 0:
            names, lens, primary = gu.header(int(rng.choice([64, 256, 1024, 4096])))
        else:
            nr = int(rng.integers(5, 201))
            lens = [int(10 ** rng.uniform(1.6, 6.5)) for _ in range(nr)]
            primary = [int(rng.random() < 0.9) for _ in range(nr)]
            names = [f"s{i}" for i in range(nr)]
        nr = len(lens)
        bases = [rng.choice(np.array([1, 2, 4, 8, 15], dtype=np.uint8), size=L, p=[.24, .24, .24, .24, .04]) for L in lens]
        live = np.flatnonzero(rng.random(nr) < float(rng.choice([0.1, 0.6, 1.0])))
        if live.size == 0:
            live = np.array([int(rng.integers(0, nr))])
        n = int(rng.integers(1, 50_000))
        max_len = int(rng.choice([36, 150, 250]))
        raw = random_batch(rng, n, [lens[r] for r in live], max_len=max_len, min_len=int(rng.integers(0, max_len + 1)), weird=bool(rng.integers(0, 2)))
        for col in ("ref_id", "mate_ref_id"):
            a = raw.cols[col]
            raw.cols[col] = np.where(a >= 0, live[np.clip(a, 0, live.size - 1)], a).astype(np.int32)
        hb = make_edit_friendly(raw, rng, bases, lens) if rng.random() < 0.6 else raw
        is_sorted = rng.random() < 0.7
        if is_sorted:
            hb = coordinate_sorted(hb)
        m = int(rng.integers(0, 4000))
        fr = rng.integers(0, nr, m).astype(np.uint32)
        fs = np.array([rng.integers(1, lens[r] + 1) for r in fr], dtype=np.uint32)
        fe = fs + np.where(rng.random(m) < 0.1, 0, rng.integers(0, 5000, m)).astype(np.uint32)
        fn = rng.choice(5, m).astype(np.uint32)
        kw = dict(facets=facets, bin_size=int(rng.choice([1000, 50_000])), max_read_len=320, gc_seed=seed, ref_bases=bases)
        orc = oracle_py.Oracle(lens, primary, **kw)
        orc.set_features(fr, fn, fs, fe)
        orc.process_batch(hb)
        rc = orc.finalize(allow_malformed=True)
        for streamed in ([False, True] if is_sorted else [False]):
            gpu = host.QcContext(lens, primary, lib=lib, sorted_input=streamed, **kw)
            gpu.set_features(fr, fn, fs, fe)
            cuts = sorted(set([0, hb.n] + [int(x) for x in rng.integers(0, hb.n + 1, 3)]))
            for lo, hi in zip(cuts[:-1], cuts[1:]):
                part = take_records(hb, np.arange(lo, hi))
                part.first_record_index = lo
                gpu.process_batch(gpu.upload(part) if rng.random() < 0.5 else part)
            assert gpu.finalize(allow_malformed=True) == rc
            compare_contexts(gpu, orc, nr, facets, kw["bin_size"], lens)
            json_equal(gpu.results(names), orc.results(names))
            gpu.close()
        print(f"genome seed {seed}: refs={nr} ({'GRCh38 names' if seed % 2 == 0 else 'random'}) total={sum(lens)} live={live.size} n={n} "
              f"sorted={is_sorted} errors={sum(orc.error_counts().values())} ok", flush=True)
    print("genome sweep ok")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome", type=int, default=0, help="seeds of the many-sequences sweep (all seven facets, up to 200 @SQ)")
    ap.add_argument("--extra", type=int, default=0, help="seeds of the Edits + Genomic Features sweep")
    ap.add_argument("--seeds", type=int, default=40)
    ap.add_argument("--sorted", type=int, default=0, help="seeds of the sorted_input (streaming Coverage) sweep")
    ap.add_argument("--ingest", type=int, default=0, help="seeds of the device-reader-against-host-reader sweep")
    ap.add_argument("--seed-base", type=int, default=0, help="first seed of every sweep (a soak behind an earlier one takes up where that left off)")
    a = ap.parse_args()
    lib = ffi.load_library()
    if a.genome:
        genome_sweep(lib, a.genome, a.seed_base)
    if a.sorted:
        sorted_sweep(lib, a.sorted, a.seed_base)
    if a.extra:
        extra_sweep(lib, a.extra, a.seed_base)
    if a.ingest:
        ingest_sweep(lib, a.ingest, a.seed_base)
    td = tempfile.mkdtemp(prefix="ngsq_fuzz_")
    for seed in range(a.seed_base, a.seed_base + a.seeds):
        rng = np.random.default_rng(1000 + seed)
        n_refs = int(rng.integers(1, 5))
        ref_len = [int(rng.integers(200, 80_000)) for _ in range(n_refs)]
        primary = [int(rng.random() < 0.8) for _ in range(n_refs)]
        max_len = int(rng.choice([17, 36, 75, 100, 150, 151, 250, 256, 257, 300]))
        n = int(rng.integers(1, 40_000))
        hb = random_batch(rng, n, ref_len, max_len=max_len, min_len=int(rng.integers(0, max_len + 1)),
                          weird=bool(rng.integers(0, 2)))
        if rng.random() < 0.5:
            hb = to_fixed_stride(hb, min_len=max_len)
        kw = dict(facets=ffi.FACETS_DEFAULT, bin_size=int(rng.choice([7, 1000, 50_000])), max_read_len=320, gc_seed=seed)
        orc = oracle_py.Oracle(ref_len, primary, **kw)
        gpu = host.QcContext(ref_len, primary, lib=lib, **kw)
        cuts = sorted(set([0, n] + [int(x) for x in rng.integers(0, n + 1, 3)]))
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            part = hb.slice(lo, hi)
            orc.process_batch(part)
            gpu.process_batch(gpu.upload(part) if rng.random() < 0.5 else part)
        assert orc.finalize(allow_malformed=True) == gpu.finalize(allow_malformed=True)
        compare_contexts(gpu, orc, n_refs, kw["facets"], kw["bin_size"], ref_len)
        names = [f"s{i}" for i in range(n_refs)]
        json_equal(gpu.results(names), orc.results(names))
        gpu.close()
        # file round trip through both ingests (qualities of 0xFF rows etc. follow the BAM rules)
        var = random_batch(rng, min(n, 6000), ref_len, max_len=max_len, weird=False)
        p = os.path.join(td, "f.bam")
        bamio.write_bam(p, var, names, ref_len, block_payload=int(rng.choice([700, 4000, 60000])))
        os.environ["NGSQ_INGEST_RAW_MB"] = str(int(rng.choice([1, 1024])))
        res = []
        for device in (False, True):
            q = host.QcContext(ref_len, primary, lib=lib, **kw)
            h = C.c_void_p()
            assert lib.ngsq_bam_open(p.encode(), 2, C.byref(h)) == 0
            while True:
                b = ffi.Batch()
                rc = (lib.ngsq_bam_next_batch_device(h, q._ctx, 2500, C.byref(b)) if device
                      else lib.ngsq_bam_next_batch(h, 2500, C.byref(b)))
                assert rc == 0, lib.ngsq_bam_last_error()
                if b.n_records == 0:
                    break
                assert lib.ngsq_process_batch(q._ctx, C.byref(b), ffi.PASS_BOTH) == 0
            lib.ngsq_bam_close(h)
            q.finalize(allow_malformed=True)
            res.append(q.results(names))
            q.close()
        json_equal(res[0], res[1])
        print(f"seed {seed}: n={n} max_len={max_len} refs={n_refs} ok", flush=True)
    print("all seeds ok")


if __name__ == "__main__":
    main()
