#!/bin/bash
# round 5, GPU job 12: Genomic Features with the closed brackets' counts per tile: parity, fuzz, timing
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_cli.py tests/test_oracle_golden.py -x -q -m gpu -k "features or facet or golden" > gpurun_out/r05_pytest_job12.log 2>&1; echo "pytest rc $?"; grep -n "passed\|failed" gpurun_out/r05_pytest_job12.log | tail -3
timeout 1200 python tools/fuzz_parity.py --seeds 0 --extra 150 > gpurun_out/r05_fuzz_extra.log 2>&1; echo "fuzz extra rc $?"; tail -2 gpurun_out/r05_fuzz_extra.log
B="--steps 20 --warmup 3 --repeats 1 --cpu-sample 0 --h2d-batch 0 --file-records 0 --live-traffic 0 --mixed-records 0"
for r in 1 2; do python bench.py $B > gpurun_out/af_job12_$r.json 2>/dev/null; python - <<PY
import json
d = json.load(open("gpurun_out/af_job12_$r.json")); a = d["all_facets"]
print("all_facets", a.get("ms_per_step"), a.get("parity_check"), {k: v["avg_ms"] for k, v in a.get("kernels", {}).items()})
print("extra features", d["extra_facets"]["kernels"].get("features"), "processed", d["extra_facets"].get("features_processed"))
PY
done
