# records per thread of k_features (measurement builds): bash tools/features_rpt.sh
set -u
cd $GRAFT_REPO_ROOT
for e in 2 4 6 8; do
  touch ngs_amd/csrc/features_kernel.hip
  NGSQ_EXTRA_FLAGS="-DNGSQ_FEATURES_RPT=$e" python -m ngs_amd.build > /tmp/b.log 2>&1 || tail -3 /tmp/b.log
  python bench.py --steps 3 --warmup 1 --cpu-sample 0 --h2d-batch 0 --file-records 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('RPT=$e', d['extra_facets']['kernels']['features'])"
done
touch ngs_amd/csrc/features_kernel.hip; python -m ngs_amd.build > /dev/null 2>&1
