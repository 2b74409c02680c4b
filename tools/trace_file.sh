#!/bin/bash
# Where the time of the file path goes (run on the GPU box):  bash tools/trace_file.sh [records] [level]
# NGSQ_INGEST_TRACE=1 makes the device reader print the wall clock of its stages per chunk.
set -u
N=${1:-60000000}
LV=${2:-6}
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd $R
python3 - <<PY
import ctypes as C, time, os, sys
sys.path.insert(0, "$R")
from ngs_amd import ffi, host
lib = ffi.load_library()
cfg = host.synth_config($N)
t = time.time()
assert lib.ngsq_synth_write_bam(C.byref(cfg), b"/tmp/trace.bam", $N, $LV, 0) == 0
print("bam_write_s", round(time.time() - t, 2), "bytes", os.path.getsize("/tmp/trace.bam"))
PY
sync
echo "nproc $(nproc); cpu.max $(cat /sys/fs/cgroup/cpu.max 2>/dev/null); mem $(free -g | sed -n 2p)"
for i in 1 2; do
  T0=$(date +%s.%N)
  NGSQ_INGEST_TRACE=1 ./ngs_amd/ngs -v qc /tmp/trace.bam GRCh38_no_alt_AnalysisSet -o /tmp 2>&1 | grep -v "Processed\|ngs::qc" | grep "\[ngs\]\|first pinned\|NUMA\|chunk:" | head -n 14 ; true
  python3 -c "import time; print('cli wall %.3f s' % (time.time() - $T0))"
done
