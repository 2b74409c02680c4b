#!/bin/bash
# Collect the rocprofv3 evidence bench.py's roofline refers to (run on the GPU box via gpurun):
#   1. --kernel-trace --stats of the default bench command  -> per-kernel average durations
#   2. --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes (MI355X_MICROARCH.md, rocprofv3 PMC slots)
# and summarise them into profiles/<prefix>_kernel_stats.csv and profiles/<prefix>_traffic.json
# (copied to gpurun_out/ so they travel back).      usage: bash tools/profile_round.sh r01
set -u
P=${1:-r01}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$P
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/stats -o out --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --cpu-sample 0 > $O/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/fetch -o out --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-timing > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/write -o out --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-timing > $O/write.log 2>&1
cd $R
S=$(find $O/stats -name '*kernel_stats.csv' | head -n 1)
F=$(find $O/fetch -name '*counter_collection.csv' | head -n 1)
W=$(find $O/write -name '*counter_collection.csv' | head -n 1)
python3 tools/collect_traffic.py "$F" "$W" "$S" gpurun_out/$P
tail -n 1 $O/stats.log
head -n 12 "$S"
