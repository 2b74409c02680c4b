#!/usr/bin/env python3
"""The device ingest's three ways of getting a BAM's bytes to the GPU (bam_device_reader.cpp reader_main),
on one synthetic file, page cache warm and cold: scan seconds, records/s, and the HOST's cost -- CPU seconds of the whole
process over the scan (user + system, all threads), i.e. compressed GB per core-second.

    python tools/reader_paths_ab.py [--records 60000000] [--scans 3] [--threads 1,2,4]

NGSQ_READER_PATH = pread (rounds 1-5: buffered reads into pinned memory) | mapped (the chunk is the file's mapping: no host copy)
| direct (O_DIRECT into pinned memory) | auto (mapped when the page cache holds the chunk, direct when it does not)."""
import argparse
import ctypes as C
import json
import os
import resource
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from ngs_amd import ffi, host  # noqa: E402

CHR1, CHR2 = 248_956_422, 242_193_529


def cpu_s():
    r = resource.getrusage(resource.RUSAGE_SELF)
    return r.ru_utime + r.ru_stime


def drop(path):
    os.sync()
    fd = os.open(path, os.O_RDONLY)
    try:
        os.fsync(fd)
        os.posix_fadvise(fd, 0, 0, os.POSIX_FADV_DONTNEED)
    finally:
        os.close(fd)


def scan(lib, ctx, bam, n):
    ctx.reset()
    t0, c0 = time.perf_counter(), cpu_s()
    h = C.c_void_p()
    assert lib.ngsq_bam_open(bam.encode(), 0, C.byref(h)) == 0, lib.ngsq_bam_last_error()
    got = 0
    while True:
        b = ffi.Batch()
        assert lib.ngsq_bam_next_batch_device(h, ctx._ctx, 1 << 22, C.byref(b)) == 0, lib.ngsq_bam_last_error()
        if b.n_records == 0:
            break
        got += int(b.n_records)
        assert lib.ngsq_process_batch(ctx._ctx, C.byref(b), ffi.PASS_BOTH) == 0, lib.ngsq_last_error(ctx._ctx)
    lib.ngsq_bam_close(h)
    ctx.finalize()
    assert got == n
    return time.perf_counter() - t0, cpu_s() - c0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--records", type=int, default=60_000_000)
    ap.add_argument("--scans", type=int, default=3)
    ap.add_argument("--threads", default="0")
    ap.add_argument("--realistic", type=int, default=0)
    a = ap.parse_args()
    lib = ffi.load_library()
    os.environ["NGSQ_BLOCKING_SYNC"] = "1"   # the driving thread sleeps while it waits: its spinning is not the reader's cost
    tmp = tempfile.mkdtemp(prefix="ngsq_paths_", dir=os.environ.get("TMPDIR", "/tmp"))
    bam = os.path.join(tmp, "f.bam")
    cfg = host.synth_config(a.records, ref_len=CHR1, n_refs=2, file_style=ffi.SYNTH_FILE_REALISTIC if a.realistic else 0)
    t0 = time.perf_counter()
    assert lib.ngsq_synth_write_bam(C.byref(cfg), bam.encode(), a.records, 6, 0) == 0
    size = os.path.getsize(bam)
    print(f"{a.records} records, {size / 1e9:.2f} GB written in {time.perf_counter() - t0:.0f} s", flush=True)
    os.sync()
    ctx = host.QcContext([CHR1, CHR2], [1, 1], max_read_len=1024, gc_seed=0x4E4753, sorted_input=True, lib=lib)
    rows = []
    docs = set()
    try:
        os.environ["NGSQ_READER_PATH"] = "pread"
        scan(lib, ctx, bam, a.records)   # the first file of the process: buffers allocated
        for threads in [int(x) for x in a.threads.split(",")]:
            if threads:
                os.environ["NGSQ_READER_THREADS"] = str(threads)
            else:
                os.environ.pop("NGSQ_READER_THREADS", None)
            for cache in ("warm", "cold"):
                for mode in ("pread", "mapped", "direct", "auto"):
                    if cache == "cold" and mode == "mapped":
                        continue   # (a mapping of a file that is not cached reads it page fault by page fault)
                    os.environ["NGSQ_READER_PATH"] = mode
                    ts, cs = [], []
                    for _ in range(a.scans):
                        if cache == "cold":
                            drop(bam)
                        t, c = scan(lib, ctx, bam, a.records)
                        ts.append(t)
                        cs.append(c)
                    docs.add(json.dumps(ctx.results(["chr1", "chr2"]), sort_keys=True))
                    t, c = sorted(ts)[len(ts) // 2], sorted(cs)[len(cs) // 2]
                    row = dict(threads=threads or "default", cache=cache, path=mode, seconds=round(t, 3), records_per_s=round(a.records / t / 1e6, 1),
                               compressed_GB_per_s=round(size / t / 1e9, 2), cpu_s=round(c, 3), GB_per_core_second=round(size / c / 1e9, 2),
                               seconds_each=[round(x, 3) for x in ts])
                    rows.append(row)
                    print(json.dumps(row), flush=True)
        print("documents identical across paths:", len(docs) == 1, flush=True)
    finally:
        ctx.close()
        for f in os.listdir(tmp):
            os.remove(os.path.join(tmp, f))
        os.rmdir(tmp)


if __name__ == "__main__":
    main()
