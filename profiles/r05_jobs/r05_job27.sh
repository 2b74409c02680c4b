#!/bin/bash
# round 5, GPU job 27: far-match bytes through the vector cache behind a buffer_inv per emit against the sc1 loads: kernel time, zlib parity
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
kt() { (cd /tmp && rm -rf /tmp/p27 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p27 -o out -- python3 $GRAFT_REPO_ROOT/tools/bench_inflate.py --records 3000000 --style $1 --reps 3 > /tmp/p27.log 2>&1; grep -c "matches zlib" /tmp/p27.log; python3 - <<'PY'
import csv, glob
for f in glob.glob('/tmp/p27/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_bgzf_inflate' in r['Name']: print('   inflate calls %s avg %.3f ms min %.3f' % (r['Calls'], float(r['AverageNs'])/1e6, float(r['MinNs'])/1e6))
PY
); }
for v in sc1 plain sc1 plain; do
  if [ $v = plain ]; then F=-DNGSQ_INFLATE_FAR_PLAIN; else F=; fi
  NGSQ_EXTRA_FLAGS="$F" python -m ngs_amd.build --force > gpurun_out/j27_build.log 2>&1 || tail -3 gpurun_out/j27_build.log
  for st in 0 3; do echo "== $v style $st"; kt $st; done
done
NGSQ_EXTRA_FLAGS=-DNGSQ_INFLATE_FAR_PLAIN python -m ngs_amd.build --force > /dev/null 2>&1
timeout 900 python -m pytest tests/test_device_ingest_gpu.py -x -q -m gpu 2>&1 | tail -2
