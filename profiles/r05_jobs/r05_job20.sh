#!/bin/bash
# round 5, GPU job 20: k_features with the brackets' entries as scalars (no per-record searches): parity, timing
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_parity_gpu.py tests/test_cli.py -x -q -m gpu -k "features or golden or random_edge or facet_subsets" 2>&1 | tail -2
timeout 900 python tools/fuzz_parity.py --seeds 40 --extra 60 > gpurun_out/r05_fuzz_job20.log 2>&1; echo "fuzz rc $?"; tail -2 gpurun_out/r05_fuzz_job20.log
B="--steps 20 --warmup 3 --repeats 1 --cpu-sample 0 --h2d-batch 0 --file-records 0 --live-traffic 0 --mixed-records 0"
python bench.py $B > gpurun_out/af_job20.json 2>/dev/null
python - <<'PY'
import json
d = json.load(open("gpurun_out/af_job20.json"))
a = d["all_facets"]
print("all_facets", a.get("ms_per_step"), a.get("parity_check"), {k: v["avg_ms"] for k, v in a.get("kernels", {}).items()})
e = d["extra_facets"]
print("extra", {k: (v.get("avg_ms") if isinstance(v, dict) else v) for k, v in e.items() if k.startswith("edits_")}, e["kernels"], e["features_processed"])
PY
