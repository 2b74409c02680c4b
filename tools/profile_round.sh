#!/bin/bash
# Collect the rocprofv3 evidence bench.py's roofline refers to (run on the GPU box via gpurun):
#   1. --kernel-trace --stats of the default bench command (facet kernels), of --workload mixed, and of the file path
#   2. --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes (MI355X_MICROARCH.md, rocprofv3 PMC slots)
# and summarise them into gpurun_out/<prefix>_*  (copy what is to be judged into profiles/).   usage: bash tools/profile_round.sh r02
set -u
P=${1:-r06}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$P
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
LEGS="--cpu-sample 0 --h2d-batch 0 --file-records 0 --extra-facet-legs 0 --mixed-records 0 --all-facets-records 0 --whole-genome-records 0 --repeats 1"
rocprofv3 --kernel-trace --stats -d $O/stats -o out --output-format csv -- python3 $R/bench.py --live-traffic 0 --steps 20 --warmup 3 $LEGS > $O/stats.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/mixed -o out --output-format csv -- python3 $R/bench.py --live-traffic 0 --workload mixed --steps 20 --warmup 3 --cpu-sample 0 --repeats 1 --whole-genome-records 0 > $O/mixed.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/extra -o out --output-format csv -- python3 $R/bench.py --live-traffic 0 --steps 3 --warmup 1 --repeats 1 --cpu-sample 0 --h2d-batch 0 --file-records 0 --mixed-records 0 --all-facets-steps 3 --whole-genome-records 0 > $O/extra.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/file -o out --output-format csv -- python3 $R/bench.py --live-traffic 0 --steps 3 --warmup 1 --cpu-sample 0 --h2d-batch 0 --extra-facet-legs 0 --mixed-records 0 --all-facets-records 0 --repeats 1 --file-records 24000000 --file-realistic-records 24000000 --whole-genome-records 0 > $O/file.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/genome -o out --output-format csv -- python3 $R/bench.py --live-traffic 0 --steps 3 --warmup 1 --repeats 1 --cpu-sample 0 --h2d-batch 0 --file-records 0 --mixed-records 0 --all-facets-records 0 --extra-facet-legs 0 > $O/genome.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/fetch -o out --output-format csv -- python3 $R/bench.py --live-traffic 0 --steps 2 --warmup 1 --no-timing $LEGS > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/write -o out --output-format csv -- python3 $R/bench.py --live-traffic 0 --steps 2 --warmup 1 --no-timing $LEGS > $O/write.log 2>&1
# the same two PMC passes for the offsets-layout quality kernel (--workload mixed) and for the device inflate (tools/bench_inflate.py:
# one launch over 4 M records = 1.09 GB of inflated bytes)
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/mfetch -o out --output-format csv -- python3 $R/bench.py --live-traffic 0 --workload mixed --steps 2 --warmup 1 --repeats 1 --no-timing --cpu-sample 0 > $O/mfetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/mwrite -o out --output-format csv -- python3 $R/bench.py --live-traffic 0 --workload mixed --steps 2 --warmup 1 --repeats 1 --no-timing --cpu-sample 0 > $O/mwrite.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/ifetch -o out --output-format csv -- python3 $R/tools/bench_inflate.py --records 4000000 --reps 2 > $O/ifetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/iwrite -o out --output-format csv -- python3 $R/tools/bench_inflate.py --records 4000000 --reps 2 > $O/iwrite.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/istats -o out --output-format csv -- python3 $R/tools/bench_inflate.py --records 4000000 --reps 2 > $O/istats.log 2>&1
cd $R
python3 tools/collect_traffic.py "$(find $O/mfetch -name '*counter_collection.csv' | head -n 1)" "$(find $O/mwrite -name '*counter_collection.csv' | head -n 1)" "$(find $O/mixed -name '*kernel_stats.csv' | head -n 1)" gpurun_out/${P}_mixed "python bench.py --live-traffic 0 --workload mixed (100 M reads of 50-300 bp, offsets layout)" > /dev/null
python3 tools/collect_traffic.py "$(find $O/ifetch -name '*counter_collection.csv' | head -n 1)" "$(find $O/iwrite -name '*counter_collection.csv' | head -n 1)" "$(find $O/istats -name '*kernel_stats.csv' | head -n 1)" gpurun_out/${P}_inflate "python tools/bench_inflate.py --records 4000000 (404 MB of BGZF blocks -> 1095 MB in one launch)" > /dev/null
S=$(find $O/stats -name '*kernel_stats.csv' | head -n 1)
F=$(find $O/fetch -name '*counter_collection.csv' | head -n 1)
W=$(find $O/write -name '*counter_collection.csv' | head -n 1)
python3 tools/collect_traffic.py "$F" "$W" "$S" gpurun_out/$P
cp "$(find $O/mixed -name '*kernel_stats.csv' | head -n 1)" gpurun_out/${P}_mixed_kernel_stats.csv
cp "$(find $O/extra -name '*kernel_stats.csv' | head -n 1)" gpurun_out/${P}_extra_kernel_stats.csv
cp "$(find $O/file -name '*kernel_stats.csv' | head -n 1)" gpurun_out/${P}_ingest_kernel_stats.csv
cp "$(find $O/genome -name '*kernel_stats.csv' | head -n 1)" gpurun_out/${P}_whole_genome_kernel_stats.csv
for f in stats mixed extra file genome; do grep '^{"metric"' $O/$f.log | tail -n 1 > gpurun_out/${P}_bench_$f.json; done
head -n 12 "$S" | cut -c1-160
