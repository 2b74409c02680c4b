#!/bin/bash
# Stage timeline (NGSQ_INGEST_TRACE=1) of one in-process scan of a file dropped from the page cache (run on the GPU box):
#   bash tools/cold_trace.sh [records]
set -u
N=${1:-60000000}
R=$GRAFT_REPO_ROOT
cd $R
python3 tools/make_bam.py /tmp/cold.bam --records $N | tail -n 1
NGSQ_INGEST_TRACE=1 python3 - <<PY
import ctypes as C, os, sys, time
sys.path.insert(0, "$R")
import bench
from ngs_amd import ffi, host
lib = ffi.load_library()
ctx = host.QcContext([248956422, 242193529], [1, 1], max_read_len=256, gc_seed=1, sorted_input=True, lib=lib)
def scan():
    ctx.reset()
    t0 = time.perf_counter()
    h = C.c_void_p()
    assert lib.ngsq_bam_open(b"/tmp/cold.bam", 0, C.byref(h)) == 0
    while True:
        b = ffi.Batch()
        assert lib.ngsq_bam_next_batch_device(h, ctx._ctx, 1 << 22, C.byref(b)) == 0
        if b.n_records == 0: break
        assert lib.ngsq_process_batch(ctx._ctx, C.byref(b), ffi.PASS_BOTH) == 0
    lib.ngsq_bam_close(h); ctx.finalize()
    return time.perf_counter() - t0
print("warm-up scan %.3f s" % scan(), file=sys.stderr)
print("dropped:", bench.drop_from_page_cache("/tmp/cold.bam"), file=sys.stderr)
print("=== COLD", file=sys.stderr)
print("cold scan %.3f s" % scan(), file=sys.stderr)
bench.drop_from_page_cache("/tmp/cold.bam")
print("storage pread GB/s (8 threads x 8 MiB): %.2f" % bench.raw_read_rate("/tmp/cold.bam"), file=sys.stderr)
PY
