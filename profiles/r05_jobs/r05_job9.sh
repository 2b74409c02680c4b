#!/bin/bash
# round 5, GPU job 9: adaptive window flushes in k_edits_rows, the list drained as it fills in k_edits: parity, fuzz (rows + GC), timings
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_cli.py tests/test_hand_bam.py -x -q -m gpu -k "not full_size" > gpurun_out/r05_pytest_job9.log 2>&1; echo "pytest rc $?"; grep -n "passed\|failed" gpurun_out/r05_pytest_job9.log | tail -3
for a in "" "--subst 0.05" "--subst 0.25" "--iid" "--aligner" "--mixed" "--mixed --subst 0.05"; do python tools/edits_time.py $a --tag "r05e $a"; done 2>&1 | grep k_edits
timeout 1200 python tools/fuzz_parity.py --seeds 0 --extra 300 > gpurun_out/r05_fuzz_extra.log 2>&1; echo "fuzz extra rc $?"; tail -2 gpurun_out/r05_fuzz_extra.log; grep -c rows gpurun_out/r05_fuzz_extra.log
B="--steps 20 --warmup 3 --repeats 1 --cpu-sample 0 --h2d-batch 0 --file-records 0 --extra-facet-legs 0 --live-traffic 0 --mixed-records 0"
python bench.py $B > gpurun_out/af_job9.json 2>/dev/null
python - <<'PY'
import json
a = json.load(open("gpurun_out/af_job9.json"))["all_facets"]
print("all_facets", a.get("ms_per_step"), a.get("ms_per_step_each_loop"), a.get("parity_check"), a.get("ms_per_step_outside_kernels"), {k: v["avg_ms"] for k, v in a.get("kernels", {}).items()})
PY
echo "== mixed (k_fields with the first offset prefetched)"
B="--steps 10 --warmup 3 --repeats 1 --cpu-sample 0 --h2d-batch 0 --file-records 0 --extra-facet-legs 0 --all-facets-records 0 --live-traffic 0 --mixed-steps 20"
for r in 1 2; do python bench.py $B > gpurun_out/mix_job9_$r.json 2>/dev/null; python - <<PY
import json
m = json.load(open("gpurun_out/mix_job9_$r.json"))["mixed"]
print("mixed", m["ms_per_step"], m["hbm_frac_whole_pass"], m["parity_check"], {k: v["avg_ms"] for k, v in m["kernels"].items()})
PY
done
