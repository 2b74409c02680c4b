#!/bin/bash
# A/B of the in-process file path under environment settings (run on the GPU box):
#   bash tools/file_ab.sh RECORDS "ENV1=a ENV2=b" "ENV1=c" ...     -- each quoted argument is one variant; "" = defaults
#   STYLE=3 (environment): the file dressed as an aligner's output (ngsq_shared.h NGSQ_SYNTH_FILE_REALISTIC); KERNELS=1: per-kernel times
set -u
N=$1; shift
R=$GRAFT_REPO_ROOT
cd $R
python3 - <<PY
import ctypes as C, time, os, sys
sys.path.insert(0, "$R")
from ngs_amd import ffi, host
lib = ffi.load_library()
cfg = host.synth_config($N, file_style=int(os.environ.get("STYLE", "0")))
t = time.time()
assert lib.ngsq_synth_write_bam(C.byref(cfg), b"/tmp/ab.bam", $N, 6, 0) == 0
print("bam_write_s", round(time.time() - t, 2), "bytes", os.path.getsize("/tmp/ab.bam"))
PY
sync
for round in 1 2; do
for V in "$@"; do
env $V python3 - <<PY
import ctypes as C, time, os, sys
sys.path.insert(0, "$R")
from ngs_amd import ffi, host
lib = ffi.load_library()
CHR1, CHR2 = 248956422, 242193529
ctx = host.QcContext([CHR1, CHR2], [1, 1], max_read_len=1024, gc_seed=1, sorted_input=True, timing=True, lib=lib)
res = []
for rep in range(3):
    ctx.reset(); ctx.kernel_timing_reset()
    t0 = time.perf_counter()
    h = C.c_void_p()
    assert lib.ngsq_bam_open(b"/tmp/ab.bam", 0, C.byref(h)) == 0
    got, first = 0, None
    while True:
        b = ffi.Batch()
        assert lib.ngsq_bam_next_batch_device(h, ctx._ctx, 1 << 22, C.byref(b)) == 0, lib.ngsq_bam_last_error()
        if b.n_records == 0: break
        got += int(b.n_records)
        if first is None: first = (time.perf_counter(), got)
        assert lib.ngsq_process_batch(ctx._ctx, C.byref(b), ffi.PASS_BOTH) == 0
    st = ffi.IngestStats()
    lib.ngsq_bam_device_stats(h, C.byref(st))
    lib.ngsq_bam_close(h)
    ctx.finalize()
    t1 = time.perf_counter()
    res.append((t1 - t0, (got - first[1]) / (t1 - first[0]) / 1e6))
kt = ctx.kernel_timing()
print("    ingest:", " ".join("%s=%d" % (k, getattr(st, k)) for k, _ in ffi.IngestStats._fields_[:6]))
print("%-40s" % "$V", " ".join("%.3fs/%.0fM" % r for r in res), " inflate %.2f ms" % (kt["bgzf_inflate"]["total_ms"] / kt["bgzf_inflate"]["launches"]))
if os.environ.get("KERNELS"):
    print("   ", "  ".join("%s %.3f x%d" % (k, v["total_ms"] / v["launches"], v["launches"]) for k, v in kt.items() if v["launches"]))
PY
done
done
