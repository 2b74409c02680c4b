#!/usr/bin/env python3
"""k_fields on 100 M x 150 bp records (bench.py's workload) for subsets of its three facets: what each part of the kernel costs.
    python tools/fields_split.py [--mixed]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from ngs_amd import ffi, host  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mixed", action="store_true")
    ap.add_argument("--records", type=int, default=100_000_000)
    a = ap.parse_args()
    lib = ffi.load_library()
    cfg = host.synth_config(100_000_000, mode=ffi.SYNTH_MIXED if a.mixed else ffi.SYNTH_FIXED, ref_len=bench.CHR1, n_refs=2)
    for name, facets in (("General", ffi.FACET_GENERAL), ("Template Length", ffi.FACET_TEMPLATE_LENGTH), ("Coverage", ffi.FACET_COVERAGE),
                         ("General + Template Length", ffi.FACET_GENERAL | ffi.FACET_TEMPLATE_LENGTH),
                         ("General + Coverage", ffi.FACET_GENERAL | ffi.FACET_COVERAGE),
                         ("all three", ffi.FACET_GENERAL | ffi.FACET_TEMPLATE_LENGTH | ffi.FACET_COVERAGE)):
        ctx = host.QcContext([bench.CHR1, bench.CHR2], [1, 1], facets=facets, max_read_len=300 if a.mixed else 150, timing=True,
                             sorted_input=True, lib=lib)
        db = ctx.synth_device_batch(cfg, 0, a.records)
        for _ in range(3):
            ctx.reset()
            ctx.kernel_timing_reset()
            ctx.process_batch(db)
            ctx.finalize()
        t = ctx.kernel_timing()
        f = t["fields"]
        print(f"{'mixed' if a.mixed else 'fixed':6s} {name:28s} k_fields {f['total_ms'] / f['launches']:.3f} ms", flush=True)
        ctx.free_batch(db)
        ctx.close()


if __name__ == "__main__":
    main()
