# phase switches of k_edits (measurement builds): bash tools/edits_exp.sh
set -u
cd $GRAFT_REPO_ROOT
for e in 0 1 2; do
  touch ngs_amd/csrc/kernels.hip
  NGSQ_EXTRA_FLAGS="-DEDITS_EXP=$e" python -m ngs_amd.build > /tmp/b.log 2>&1 || tail -3 /tmp/b.log
  python bench.py --steps 3 --warmup 1 --cpu-sample 0 --h2d-batch 0 --file-records 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('EDITS_EXP=$e', d['extra_facets']['kernels']['edits'])"
done
touch ngs_amd/csrc/kernels.hip; python -m ngs_amd.build > /dev/null 2>&1
