#!/bin/bash
# wall clock of `ngs qc FILE` under environment settings (run on the GPU box):  bash tools/cli_ab.sh RECORDS "ENV=.." ...
set -u
N=$1; shift
R=$GRAFT_REPO_ROOT
cd $R
python3 - <<PY
import ctypes as C, os, sys
sys.path.insert(0, "$R")
from ngs_amd import ffi, host
lib = ffi.load_library()
cfg = host.synth_config($N)
assert lib.ngsq_synth_write_bam(C.byref(cfg), b"/tmp/cli.bam", $N, 6, 0) == 0
PY
sync
for round in 1 2 3; do
for V in "$@"; do
python3 - <<PY
import subprocess, time, os
env = dict(os.environ)
for kv in "$V".split():
    k, v = kv.split("=", 1); env[k] = v
t = time.time()
r = subprocess.run(["./ngs_amd/ngs", "qc", "/tmp/cli.bam", "GRCh38_no_alt_AnalysisSet", "-o", "/tmp"], env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
print("%-32s rc %d  %.3f s  json %d bytes" % ("$V", r.returncode, time.time() - t, os.path.getsize("/tmp/cli.bam.results.json")))
PY
done
done
