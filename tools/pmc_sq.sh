# SQ counters of the facet kernels (instruction mix, LDS activity, stalls), one rocprofv3 --pmc pass per counter set.
# Run on the GPU box:  bash tools/pmc_sq.sh [bench.py args]     -> gpurun_out/sq_summary.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
k=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS_ATOMIC SQ_LDS_ADDR_CONFLICT"; do
  k=$((k+1))
  rocprofv3 --pmc $set --kernel-trace -d $R/gpurun_out/pmc_set$k -o out --output-format csv -- python3 $R/bench.py --live-traffic 0 --steps 2 --warmup 1 --cpu-sample 0 --no-timing "$@" > $R/gpurun_out/pmc_set$k.log 2>&1
done
cd $R && python3 - <<'PY' | tee gpurun_out/sq_summary.txt
import csv,glob,collections
tab=collections.defaultdict(dict)
for f in sorted(glob.glob('gpurun_out/pmc_set*/**/*counter_collection.csv', recursive=True)):
    acc=collections.defaultdict(lambda:[0.0,0])
    for r in csv.DictReader(open(f)):
        name=r['Kernel_Name'].split('(')[0].replace('void ','')
        if 'ngsq::' not in name: continue
        k=(name[-34:], r['Counter_Name'])
        acc[k][0]+=float(r['Counter_Value']); acc[k][1]+=1
    for (kn,cn),v in acc.items(): tab[kn][cn]=v[0]/v[1]
for kn in sorted(tab):
    print(kn)
    for cn in sorted(tab[kn]): print('    %-26s %.4g'%(cn, tab[kn][cn]))
PY
