#!/bin/bash
# A/B of the in-process file path under BUILD flags (run on the GPU box; rebuilds the library for every variant):
#   bash tools/file_build_ab.sh RECORDS "" "-DNGSQ_INFLATE_WAVES=7" ...      -- each quoted argument is one variant
set -u
N=$1; shift
R=$GRAFT_REPO_ROOT
cd $R
python3 tools/make_bam.py /tmp/ab.bam --records $N | tail -n 1
sync
for round in 1 2; do
for V in "$@"; do
NGSQ_EXTRA_FLAGS="$V" python3 -m ngs_amd.build --force > /tmp/build.log 2>&1 || tail -3 /tmp/build.log
python3 - "$V" <<PY
import ctypes as C, time, os, sys
sys.path.insert(0, "$R")
from ngs_amd import ffi, host
lib = ffi.load_library()
CHR1, CHR2 = 248956422, 242193529
ctx = host.QcContext([CHR1, CHR2], [1, 1], max_read_len=256, gc_seed=1, sorted_input=True, timing=True, lib=lib)
res = []
for rep in range(4):
    ctx.reset(); ctx.kernel_timing_reset()
    t0 = time.perf_counter()
    h = C.c_void_p()
    assert lib.ngsq_bam_open(b"/tmp/ab.bam", 0, C.byref(h)) == 0
    while True:
        b = ffi.Batch()
        assert lib.ngsq_bam_next_batch_device(h, ctx._ctx, 1 << 22, C.byref(b)) == 0, lib.ngsq_bam_last_error()
        if b.n_records == 0: break
        assert lib.ngsq_process_batch(ctx._ctx, C.byref(b), ffi.PASS_BOTH) == 0
    lib.ngsq_bam_close(h)
    ctx.finalize()
    res.append(time.perf_counter() - t0)
kt = ctx.kernel_timing()
print("%-32s %s  inflate %.2f ms" % (sys.argv[1] or "(default)", " ".join("%.3f" % r for r in res), kt["bgzf_inflate"]["total_ms"] / kt["bgzf_inflate"]["launches"]))
if os.environ.get("KERNELS"):
    print("   ", "  ".join("%s %.3f x%d" % (k, v["total_ms"] / v["launches"], v["launches"]) for k, v in kt.items() if v["launches"]))
PY
done
done
python3 -m ngs_amd.build --force > /dev/null 2>&1
