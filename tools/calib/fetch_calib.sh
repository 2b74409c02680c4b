# FETCH_SIZE / TCC request counters of (a) the calibration kernels of fetch_calib.hip (known byte counts) and (b) the Quality
# Score kernels of the fixed and the mixed workload.  On the GPU box:  bash tools/calib/fetch_calib.sh  -> gpurun_out/r04_fetch_calib.txt
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
hipcc -O3 --offload-arch=gfx950 $R/tools/calib/fetch_calib.hip -o /tmp/fetch_calib 2>/dev/null || exit 1
/tmp/fetch_calib 4096 10 15 > /tmp/calib_plain.txt
k=0
for set in "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_READ_sum"; do
  k=$((k+1))
  rm -rf /tmp/cal_a$k /tmp/cal_b$k /tmp/cal_c$k
  rocprofv3 --pmc $set --kernel-trace -d /tmp/cal_a$k -o out --output-format csv -- /tmp/fetch_calib 4096 10 15 > /tmp/cal_a$k.log 2>&1
  rocprofv3 --pmc $set --kernel-trace -d /tmp/cal_b$k -o out --output-format csv -- python3 $R/bench.py --live-traffic 0 --workload mixed --steps 2 --warmup 1 --cpu-sample 0 > /tmp/cal_b$k.log 2>&1
  rocprofv3 --pmc $set --kernel-trace -d /tmp/cal_c$k -o out --output-format csv -- python3 $R/bench.py --live-traffic 0 --steps 2 --warmup 1 --cpu-sample 0 --file-records 0 --h2d-batch 0 --extra-facet-legs 0 --mixed-records 0 > /tmp/cal_c$k.log 2>&1
done
{ cat /tmp/calib_plain.txt
python3 - <<'PY'
import csv,glob,collections
tab=collections.defaultdict(dict)
for f in sorted(glob.glob('/tmp/cal_[abc]*/**/*counter_collection.csv', recursive=True)):
    acc=collections.defaultdict(lambda:[0.0,0])
    for r in csv.DictReader(open(f)):
        name=r['Kernel_Name'].split('(')[0].replace('void ','')
        if 'k_calib' not in name and 'k_qual' not in name: continue
        k=(name[-36:], r['Counter_Name'])
        acc[k][0]+=float(r['Counter_Value']); acc[k][1]+=1
    for (kn,cn),v in acc.items(): tab[kn][cn]=v[0]/v[1]
for kn in sorted(tab):
    print(kn)
    for cn in sorted(tab[kn]): print('    %-26s %.5g'%(cn, tab[kn][cn]))
PY
} | tee $R/gpurun_out/r04_fetch_calib.txt
