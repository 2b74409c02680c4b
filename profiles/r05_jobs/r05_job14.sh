#!/bin/bash
# round 5, GPU job 14: where the ragged Edits time goes (rows-kernel vs walk), and the scan fix on the realistic file
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for a in "--mixed" "--mixed --subst 0.25"; do
  d=gpurun_out/prof_j14_$(echo $a | tr -d ' -.')
  rocprofv3 --kernel-trace --stats -d $d -o out -- python3 tools/edits_time.py $a --tag "j14 $a" > $d.log 2>&1
  echo "== $a"; grep k_edits $d.log
  python3 - <<PY
import csv, glob
for f in glob.glob("$d/**/out_kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "edits" in r["Name"]:
            print("   %-70s calls %4s avg %9.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
timeout 600 python -m pytest tests/test_device_ingest_gpu.py -x -q -m gpu -k "scan_both or larger_than or ragged or offsets" 2>&1 | tail -2
python tools/steady_scan.py --records 60000000 --style 3 --preread 2 2>&1 | tail -4
