#!/usr/bin/env python3
"""Are in-process scans of one BAM alike?  (the second of three scans of a freshly written file took 70 % longer.)

    python tools/steady_scan.py --records 30000000 --style 3 --scans 6 [--preread 2] [--trace]

Writes a synthetic BAM, optionally reads it --preread times with plain pread()s (a page of the page cache is promoted from
the inactive to the active list on its second access: the second read of a fresh file is the slow one whoever reads), then
scans it --scans times through ngsq_bam_open / next_batch_device / process_batch / finalize and prints every scan's time.
"""
import argparse
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ngs_amd import ffi, host  # noqa: E402

CHR1, CHR2 = 248_956_422, 242_193_529


def pread_all(path, threads=8):
    from concurrent.futures import ThreadPoolExecutor
    size = os.path.getsize(path)
    piece = 64 << 20
    fd = os.open(path, os.O_RDONLY)

    def rd(off):
        left, o = min(piece, size - off), off
        while left > 0:
            got = len(os.pread(fd, min(left, 8 << 20), o))
            if not got:
                break
            left -= got
            o += got
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=threads) as ex:
        list(ex.map(rd, range(0, size, piece)))
    os.close(fd)
    return time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--records", type=int, default=30_000_000)
    ap.add_argument("--style", type=int, default=0)
    ap.add_argument("--scans", type=int, default=6)
    ap.add_argument("--preread", type=int, default=0)
    ap.add_argument("--path", default="/tmp/steady.bam")
    ap.add_argument("--keep", action="store_true")
    args = ap.parse_args()
    lib = ffi.load_library()
    n = args.records
    if not os.path.exists(args.path):
        cfg = host.synth_config(n, ref_len=CHR1, n_refs=2, file_style=args.style)
        t0 = time.perf_counter()
        assert lib.ngsq_synth_write_bam(C.byref(cfg), args.path.encode(), n, 6, 0) == 0
        print("written in %.1f s, %d bytes" % (time.perf_counter() - t0, os.path.getsize(args.path)), flush=True)
        os.sync()
    for k in range(args.preread):
        print("plain read %d: %.3f s" % (k, pread_all(args.path)), flush=True)
    ctx = host.QcContext([CHR1, CHR2], [1, 1], max_read_len=1024, gc_seed=1, sorted_input=True, timing=False, lib=lib)
    times = []
    for rep in range(args.scans):
        ctx.reset()
        t0 = time.perf_counter()
        h = C.c_void_p()
        assert lib.ngsq_bam_open(args.path.encode(), 0, C.byref(h)) == 0
        got = 0
        while True:
            b = ffi.Batch()
            assert lib.ngsq_bam_next_batch_device(h, ctx._ctx, 1 << 22, C.byref(b)) == 0, lib.ngsq_bam_last_error()
            if b.n_records == 0:
                break
            got += int(b.n_records)
            assert lib.ngsq_process_batch(ctx._ctx, C.byref(b), ffi.PASS_BOTH) == 0
        lib.ngsq_bam_close(h)
        ctx.finalize()
        times.append(time.perf_counter() - t0)
        assert got == n
    ctx.close()
    med = sorted(times)[len(times) // 2]
    print("style %d preread %d pool %s: " % (args.style, args.preread, os.environ.get("NGSQ_POOL_MB", "default")) +
          " ".join("%.3f" % t for t in times) + "  median %.3f = %.0f M records/s; spread of scans 1.. %+.0f %%"
          % (med, n / med / 1e6, 100 * (max(times[1:]) / min(times[1:]) - 1)), flush=True)
    if not args.keep:
        os.remove(args.path)


if __name__ == "__main__":
    main()
