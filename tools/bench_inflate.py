#!/usr/bin/env python3
"""Device BGZF inflate on a synthetic BAM: bytes, blocks, and the hook's wall time.  Kernel time
comes from `rocprofv3 --kernel-trace --stats -- python3 tools/bench_inflate.py`.
    python tools/bench_inflate.py [--records 2000000] [--level 6]
"""
import argparse
import ctypes as C
import os
import sys
import tempfile
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from ngs_amd import build, ffi, host  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--records", type=int, default=2_000_000)
    ap.add_argument("--level", type=int, default=6)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--crc", type=int, default=1)
    ap.add_argument("--style", type=int, default=0, help="ngsq_shared.h NGSQ_SYNTH_FILE_*: 3 = an aligner's names, tags and CIGAR mix")
    args = ap.parse_args()
    build.build(verbose=False)
    lib = ffi.load_library()
    scfg = host.synth_config(n_total=args.records, read_len=150, file_style=args.style)
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "s.bam")
        rc = lib.ngsq_synth_write_bam(C.byref(scfg), path.encode(), args.records, args.level, 0)
        assert rc == 0
        comp = open(path, "rb").read()
    ctx = host.QcContext([1000], [1], lib=lib)
    n = C.c_uint64(0)
    out = np.empty(1, np.uint8)
    lib.ngsq_bgzf_inflate_device(ctx._ctx, comp, len(comp), out.ctypes.data, 1, C.byref(n), 0)
    out = np.empty(n.value, np.uint8)
    for _ in range(args.reps):
        t0 = time.perf_counter()
        rc = lib.ngsq_bgzf_inflate_device(ctx._ctx, comp, len(comp), out.ctypes.data, out.size, C.byref(n), args.crc)
        dt = time.perf_counter() - t0
        assert rc == 0, lib.ngsq_last_error(ctx._ctx)
        print(f"compressed {len(comp)/1e6:.1f} MB -> {n.value/1e6:.1f} MB, hook wall {dt*1e3:.1f} ms "
              f"({n.value/dt/1e9:.2f} GB/s out incl. copies and allocation)", flush=True)
    # spot check against zlib on the first block
    t0 = time.perf_counter()
    d = zlib.decompressobj(31)
    first = d.decompress(comp[:70000])
    assert bytes(out[:len(first)]) == first
    print("first block matches zlib")


if __name__ == "__main__":
    main()
