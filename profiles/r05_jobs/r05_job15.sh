#!/bin/bash
# round 5, GPU job 15: M (D|N) M of any gap on the window lanes (rows + ragged): parity, fuzz, timings, the kernels' split
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_parity_gpu.py tests/test_stager.py tests/test_cli.py tests/test_hand_bam.py tests/test_device_ingest_gpu.py -x -q -m gpu -k "not full_size" > gpurun_out/r05_pytest_job15.log 2>&1; echo "pytest rc $?"; grep -n "passed\|failed" gpurun_out/r05_pytest_job15.log | tail -3
for a in "--mixed" "--mixed --subst 0.05" "--mixed --subst 0.25" "--mixed --iid" "" "--aligner" "--subst 0.25"; do python tools/edits_time.py $a --tag "r05h $a"; done 2>&1 | grep k_edits
timeout 1200 python tools/fuzz_parity.py --seeds 60 --extra 200 > gpurun_out/r05_fuzz_job15.log 2>&1; echo "fuzz rc $?"; tail -2 gpurun_out/r05_fuzz_job15.log
for a in "--mixed" "--mixed --subst 0.25"; do
  d=gpurun_out/prof_j15_$(echo $a | tr -d ' -.')
  rocprofv3 --kernel-trace --stats --output-format csv -d $d -o out -- python3 tools/edits_time.py $a --tag "j15 $a" > $d.log 2>&1
  echo "== $a"; grep k_edits $d.log
  python3 - <<PY
import csv, glob
for f in glob.glob("$d/**/out_kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "edits" in r["Name"]:
            print("   %-70s calls %4s avg %9.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
B="--steps 20 --warmup 3 --repeats 1 --cpu-sample 0 --h2d-batch 0 --file-records 0 --extra-facet-legs 0 --live-traffic 0 --mixed-records 0"
python bench.py $B > gpurun_out/af_job15.json 2>/dev/null
python - <<'PY'
import json
a = json.load(open("gpurun_out/af_job15.json"))["all_facets"]
print("all_facets", a.get("ms_per_step"), a.get("parity_check"), {k: v["avg_ms"] for k, v in a.get("kernels", {}).items()})
PY
