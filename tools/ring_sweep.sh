set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for r in ${RINGS:-32768 16384 8192}; do
  NGSQ_EXTRA_FLAGS=-DNGSQ_INFLATE_RING=$r python -m ngs_amd.build --force > /tmp/build_$r.log 2>&1 || { tail -n 5 /tmp/build_$r.log; continue; }
  echo "== RING $r"
  timeout 300 python -m pytest tests/test_device_ingest_gpu.py -x -q 2>&1 | tail -n 2
  (cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/ring_$r -o out --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/bench_inflate.py --records 4000000 > $GRAFT_REPO_ROOT/gpurun_out/ring_$r.log 2>&1)
  head -n 2 gpurun_out/ring_$r/out_kernel_stats.csv | tail -n 1 | cut -d, -f1-4 | cut -c1-40,120-
done
