# Measurement builds of k_cov_stream (ST_EXP switches in cov_stream.hip): which phase bounds it.
# Run on the GPU box:  bash tools/exp_stream.sh   (results are wrong for ST_EXP != 0; timing only)
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for x in ${EXPS:-0 1 2 3 4}; do
  touch ngs_amd/csrc/cov_stream.hip
  NGSQ_EXTRA_FLAGS="-DST_EXP=$x ${EXTRA:-}" python -m ngs_amd.build > /tmp/build_$x.log 2>&1 || { tail -n 5 /tmp/build_$x.log; continue; }
  echo "== ST_EXP $x ${EXTRA:-}"
  NGSQ_EXTRA_FLAGS="-DST_EXP=$x ${EXTRA:-}" timeout 300 python bench.py --cpu-sample 0 ${BENCH_ARGS:-} 2>&1 | tail -n 1 | grep -o '"fields": {[^}]*}\|"cov_stream": {[^}]*}\|"ms_per_step": [0-9.]*\|FAILED[^"]*'
done
touch ngs_amd/csrc/cov_stream.hip
python -m ngs_amd.build > /tmp/build_final.log 2>&1
