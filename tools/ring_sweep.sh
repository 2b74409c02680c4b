# Kernel time of the device inflate for several LDS ring sizes (run on the GPU box):
#   RINGS="8192 4096" bash tools/ring_sweep.sh [records]
# Only bgzf_inflate.hip depends on the macro: it is touched and rebuilt with the flag, the other objects stay.
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
N=${1:-4000000}
for r in ${RINGS:-8192 4096}; do
  touch ngs_amd/csrc/bgzf_inflate.hip
  NGSQ_EXTRA_FLAGS=-DNGSQ_INFLATE_RING=$r python -m ngs_amd.build > /tmp/build_$r.log 2>&1 || { tail -n 5 /tmp/build_$r.log; continue; }
  echo "== RING $r"
  (cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/ring_$r -o out --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/bench_inflate.py --records $N > $GRAFT_REPO_ROOT/gpurun_out/ring_$r.log 2>&1)
  tail -n 2 gpurun_out/ring_$r.log
  grep -h "k_bgzf" gpurun_out/ring_$r/*kernel_stats.csv | cut -d, -f1-5 | sed 's/(.*)//' | cut -c1-120
done
touch ngs_amd/csrc/bgzf_inflate.hip
python -m ngs_amd.build > /tmp/build_default.log 2>&1
echo "== default build"
timeout 600 python -m pytest tests/test_device_ingest_gpu.py -x -q 2>&1 | tail -n 3
