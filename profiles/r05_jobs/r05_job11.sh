#!/bin/bash
# round 5, GPU job 11: the mismatching windows listed and resolved densely: parity, fuzz, timings
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_cli.py -x -q -m gpu -k "edits or gc_content or facet or hand_golden or sharded" > gpurun_out/r05_pytest_job11.log 2>&1; echo "pytest rc $?"; grep -n "passed\|failed" gpurun_out/r05_pytest_job11.log | tail -3
for a in "" "--subst 0.05" "--subst 0.25" "--iid" "--aligner"; do python tools/edits_time.py $a --tag "r05f $a"; done 2>&1 | grep k_edits
timeout 1200 python tools/fuzz_parity.py --seeds 0 --extra 200 > gpurun_out/r05_fuzz_extra.log 2>&1; echo "fuzz extra rc $?"; tail -2 gpurun_out/r05_fuzz_extra.log
B="--steps 20 --warmup 3 --repeats 1 --cpu-sample 0 --h2d-batch 0 --file-records 0 --extra-facet-legs 0 --live-traffic 0 --mixed-records 0"
python bench.py $B > gpurun_out/af_job11.json 2>/dev/null
python - <<'PY'
import json
a = json.load(open("gpurun_out/af_job11.json"))["all_facets"]
print("all_facets", a.get("ms_per_step"), a.get("ms_per_step_each_loop"), a.get("parity_check"), a.get("ms_per_step_outside_kernels"), {k: v["avg_ms"] for k, v in a.get("kernels", {}).items()})
PY
