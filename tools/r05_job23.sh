#!/bin/bash
# round 5, GPU job 23: the Edits teardown without the scan of the chunk sums (two-level sums): parity, all_facets
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_parity_gpu.py tests/test_cli.py tests/test_stager.py -x -q -m gpu -k "not full_size" > gpurun_out/r05_pytest_job23.log 2>&1; echo "pytest rc $?"; grep -n "passed\|failed" gpurun_out/r05_pytest_job23.log | tail -3
timeout 900 python tools/fuzz_parity.py --seeds 30 --extra 60 > gpurun_out/r05_fuzz_job23.log 2>&1; echo "fuzz rc $?"; tail -2 gpurun_out/r05_fuzz_job23.log
B="--steps 20 --warmup 3 --repeats 1 --cpu-sample 0 --h2d-batch 0 --file-records 0 --live-traffic 0 --mixed-records 0 --extra-facet-legs 0"
for r in 1 2; do python bench.py $B > gpurun_out/af_job23.json 2>/dev/null
python - <<'PY'
import json
d = json.load(open("gpurun_out/af_job23.json"))
a = d["all_facets"]
print("all_facets", a.get("ms_per_step"), a.get("ms_per_step_outside_kernels"), a.get("parity_check"), {k: v["avg_ms"] for k, v in a.get("kernels", {}).items()})
PY
done
