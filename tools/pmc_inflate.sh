cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" "SQ_WAVES SQ_INST_CYCLES_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-30)
  rocprofv3 --pmc $set --kernel-trace -d $R/gpurun_out/pinf_$n -o out --output-format csv -- python3 $R/tools/bench_inflate.py --records 4000000 --reps 1 ${INF_ARGS:-} > $R/gpurun_out/pinf_$n.log 2>&1
done
cd $R && python3 - <<'PY'
import csv,glob,collections
for f in sorted(glob.glob('gpurun_out/pinf_*/**/*counter_collection.csv', recursive=True)):
    acc=collections.defaultdict(lambda:[0.0,0])
    for r in csv.DictReader(open(f)):
        if 'inflate' not in r['Kernel_Name']: continue
        k=r['Counter_Name']
        acc[k][0]+=float(r['Counter_Value']); acc[k][1]+=1
    for k,v in sorted(acc.items()):
        print(k, '%.4g'%(v[0]/v[1]), v[1])
PY
