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
from tests.util import compare_contexts, json_equal, random_batch, to_fixed_stride  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=40)
    a = ap.parse_args()
    lib = ffi.load_library()
    td = tempfile.mkdtemp(prefix="ngsq_fuzz_")
    for seed in range(a.seeds):
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
