# rocprofv3 kernel stats of the non-default bench workloads (run on the GPU box via gpurun):
#   --workload mixed (BASELINE configs[4] shape) and --coverage array (the path the N >= 2 scaling runs take)
# -> gpurun_out/<prefix>_mixed_kernel_stats.csv, <prefix>_array_kernel_stats.csv      usage: bash tools/profile_variants.sh r01
set -u
P=${1:-r01}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in mixed array; do
  O=$R/gpurun_out/prof_${P}_$v
  mkdir -p $O
  if [ $v = mixed ]; then A="--workload mixed"; else A="--coverage array"; fi
  rocprofv3 --kernel-trace --stats -d $O -o out --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --cpu-sample 0 $A > $O/log.txt 2>&1
  S=$(find $O -name '*kernel_stats.csv' | head -n 1)
  cp "$S" $R/gpurun_out/${P}_${v}_kernel_stats.csv
  tail -n 1 $O/log.txt | cut -c1-200
done
