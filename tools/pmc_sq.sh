cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for set in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-30)
  rocprofv3 --pmc $set --kernel-trace -d $R/gpurun_out/pmc_$n -o out --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-timing > $R/gpurun_out/pmc_$n.log 2>&1
done
cd $R && python3 - <<'PY'
import csv,glob,collections
for f in sorted(glob.glob('gpurun_out/pmc_*/**/*counter_collection.csv', recursive=True)):
    acc=collections.defaultdict(lambda:[0.0,0])
    for r in csv.DictReader(open(f)):
        k=(r['Kernel_Name'].split('(')[0][-40:], r['Counter_Name'])
        acc[k][0]+=float(r['Counter_Value']); acc[k][1]+=1
    for k,v in sorted(acc.items()):
        if 'qual' in k[0] or 'fields' in k[0] or 'k_gc' in k[0] or 'cov_scan' in k[0]:
            print(k[0], k[1], '%.4g'%(v[0]/v[1]), v[1])
PY
