#!/bin/bash
# Cold-cache in-process scans under environment settings (run on the GPU box):  bash tools/cold_ab.sh RECORDS "ENV=a" "ENV=b" ...
set -u
N=$1; shift
R=$GRAFT_REPO_ROOT
cd $R
python3 tools/make_bam.py /tmp/cold.bam --records $N | tail -n 1
for round in 1 2; do
for V in "$@"; do
env $V python3 - "$V" <<PY
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
scan()
res = []
for rep in range(3):
    assert bench.drop_from_page_cache("/tmp/cold.bam")
    res.append(scan())
print("%-44s cold scans %s" % (sys.argv[1] or "(default)", " ".join("%.3f" % r for r in res)))
PY
done
done
