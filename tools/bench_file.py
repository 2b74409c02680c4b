#!/usr/bin/env python3
"""Secondary measurements quoted in DESIGN.md section 7 (never bench.py's `value`):
  (a) PCIe-inclusive rate: pinned host SoA batches through ngsq_process_batch (H2D + kernels)
  (b) file end-to-end: synthetic BGZF BAM -> `ngs qc` -> results.json, wall clock
    python tools/bench_file.py [--records 20000000] [--batch 4000000]
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from ngs_amd import build, ffi, host  # noqa: E402

CHR1, CHR2 = 248_956_422, 242_193_529


def pinned_copy(lib, hb):
    """Copy a HostBatch's columns into hipHostMalloc'd memory (kept alive by the returned list)."""
    keep, cols = [], {}
    for k, a in hb.cols.items():
        if a is None:
            cols[k] = None
            continue
        p = C.c_void_p()
        assert lib.ngsq_host_malloc_pinned(max(a.nbytes, 64), C.byref(p)) == 0
        dst = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(max(a.nbytes, 64),))
        dst[:a.nbytes] = a.view(np.uint8).reshape(-1)
        cols[k] = dst[:a.nbytes].view(a.dtype)
        keep.append(p)
    return host.HostBatch(hb.n, cols, hb.seq_stride, hb.qual_stride, hb.cigar_stride, hb.first_record_index), keep


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--records", type=int, default=20_000_000)
    ap.add_argument("--batch", type=int, default=4_000_000)
    ap.add_argument("--level", type=int, default=1)
    ap.add_argument("--skip-pcie", action="store_true")
    args = ap.parse_args()
    build.build(verbose=False)
    lib = ffi.load_library()
    out = {}
    # ---- (a) PCIe-inclusive
    if not args.skip_pcie:
        scfg = host.synth_config(100_000_000, ref_len=CHR1, n_refs=2)
        hb = host.synth_host_batch(scfg, 0, args.batch, lib)
        pb, keep = pinned_copy(lib, hb)
        ctx = host.QcContext([CHR1, CHR2], [1, 1], max_read_len=150, gc_seed=0x4E4753, lib=lib)
        for variant, b in (("pinned", pb), ("pageable", hb)):
            ctx.reset()
            ctx.process_batch(b)
            ctx.synchronize()
            t0 = time.perf_counter()
            reps = 6
            for _ in range(reps):
                ctx.process_batch(b)
            ctx.synchronize()
            dt = time.perf_counter() - t0
            out[f"h2d_inclusive_{variant}"] = {"records_per_s": round(reps * b.n / dt), "GB_per_s": round(reps * b.n * 254 / dt / 1e9, 1),
                                                "batch_records": b.n}
        ctx.finalize()
        ctx.close()
    # ---- (b) file end to end
    tmp = tempfile.mkdtemp(prefix="ngsq_bench_", dir=os.environ.get("TMPDIR", "/tmp"))
    bam = os.path.join(tmp, "synth.bam")
    fcfg = host.synth_config(args.records, ref_len=CHR1, n_refs=2)
    t0 = time.perf_counter()
    assert lib.ngsq_synth_write_bam(C.byref(fcfg), bam.encode(), args.records, args.level, 0) == 0
    out["bam_write_s"] = round(time.perf_counter() - t0, 2)
    out["bam_bytes"] = os.path.getsize(bam)
    ngs = build.build_cli(verbose=False)
    results = {}
    for ingest in ("host", "device"):
        for run_i in range(2):
            t0 = time.perf_counter()
            r = subprocess.run([ngs, "-q", "qc", bam, "GRCh38_no_alt_AnalysisSet", "-o", tmp, "--ingest", ingest],
                               capture_output=True, text=True)
            dt = time.perf_counter() - t0
            assert r.returncode == 0, r.stderr
            out[f"file_end_to_end_{ingest}_run{run_i}"] = {
                "seconds": round(dt, 2), "records_per_s": round(args.records / dt),
                "compressed_MB_per_s": round(out["bam_bytes"] / dt / 1e6)}
        results[ingest] = json.load(open(os.path.join(tmp, "synth.bam.results.json")))
    out["device_ingest_json_equals_host_ingest_json"] = results["host"] == results["device"]
    res = json.load(open(os.path.join(tmp, "synth.bam.results.json")))
    out["check_total"] = res["general"]["records"]["total"]
    out["host_cores"] = os.cpu_count()
    print(json.dumps(out))
    for f in os.listdir(tmp):
        os.remove(os.path.join(tmp, f))
    os.rmdir(tmp)


if __name__ == "__main__":
    main()
