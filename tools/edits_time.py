#!/usr/bin/env python3
"""k_edits alone on the bench's Edits workload (reads sampled from the reference): kernel time by HIP events.
    python tools/edits_time.py [--records 100000000] [--mixed] [--iid]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ngs_amd import ffi, host  # noqa: E402

CHR1, CHR2 = 248_956_422, 242_193_529


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--records", type=int, default=100_000_000)
    ap.add_argument("--mixed", action="store_true")
    ap.add_argument("--iid", action="store_true")
    ap.add_argument("--aligner", action="store_true", help="an aligner's CIGAR mix on the fixed-length reads: 9 %% soft clips, 3 %% insertions, 3 %% deletions")
    ap.add_argument("--subst", type=float, default=0.0, help="fraction of the compared bases substituted (default: the model's 0.5 %%)")
    ap.add_argument("--gc", action="store_true", help="GC Content in the same context: the variant of k_edits_rows that tallies it")
    ap.add_argument("--tag", default="")
    a = ap.parse_args()
    lib = ffi.load_library()
    cfg = host.synth_config(100_000_000, mode=ffi.SYNTH_MIXED if a.mixed else ffi.SYNTH_FIXED, ref_len=CHR1, n_refs=2,
                            seq_model=ffi.SYNTH_SEQ_IID if a.iid else (ffi.synth_seq_subst(a.subst) if a.subst else ffi.SYNTH_SEQ_FROM_REFERENCE),
                            file_style=ffi.SYNTH_FILE_CIGAR_MIX if a.aligner else 0)
    bases = [host.synth_reference(cfg, 0, CHR1, lib), None]
    ctx = host.QcContext([CHR1, CHR2], [1, 1], facets=ffi.FACET_EDITS | (ffi.FACET_GC_CONTENT if a.gc else 0), max_read_len=300 if a.mixed else 150, timing=True, ref_bases=bases, lib=lib)
    db = ctx.synth_device_batch(cfg, 0, a.records)
    for _ in range(3):
        ctx.reset()
        ctx.kernel_timing_reset()
        ctx.process_batch(db)
        ctx.finalize()
    t = ctx.kernel_timing()
    r1, r2, vaf = ctx.edits()
    e, v = t["edits"], t["edits_vaf"]
    print(f"{a.tag:28s} k_edits {e['total_ms'] / e['launches']:.3f} ms  ({e['algo_bytes'] / e['total_ms'] / 1e6 / 8000:.3f} of 8 TB/s)   "
          f"teardown {v['total_ms']:.3f} ms   reads {int(r1.sum() + r2.sum())} vaf {int(vaf.sum())}", flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
