#!/usr/bin/env python3
"""k_features alone on 100 M records (the reads of bench.py's extra_facets leg, the synthetic gene model): its time for the
NGSQ_FEATURES_BLOCKS_PER_CU of the environment.    python tools/features_sweep.py [--tag x]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import bench  # noqa: E402
from ngs_amd import ffi, host  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", default="")
    ap.add_argument("--records", type=int, default=100_000_000)
    a = ap.parse_args()
    lib = ffi.load_library()
    cfg = host.synth_config(100_000_000, ref_len=bench.CHR1, n_refs=2, seq_model=ffi.SYNTH_SEQ_FROM_REFERENCE)
    ctx = host.QcContext([bench.CHR1, bench.CHR2], [1, 1], facets=ffi.FACET_FEATURES, max_read_len=150, timing=True, lib=lib)
    ctx.set_features(*bench.synthetic_gene_model(np))
    db = ctx.synth_device_batch(cfg, 0, a.records)
    for _ in range(3):
        ctx.reset()
        ctx.kernel_timing_reset()
        ctx.process_batch(db)
        ctx.finalize()
    t = ctx.kernel_timing()["features"]
    print(f"{a.tag:24s} k_features {t['total_ms'] / t['launches']:.3f} ms   processed {ctx.features()['processed']}", flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
