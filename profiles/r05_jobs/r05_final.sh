#!/bin/bash
# round 5: the evidence of the final tree -- the whole -m gpu suite, the driver-shaped bench line, the rocprofv3 summaries
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r05_pytest_gpu.log 2>&1; echo "pytest rc $?"; grep -n "passed\|failed" gpurun_out/r05_pytest_gpu.log | tail -3
timeout 1700 python bench.py > gpurun_out/r05_bench_default.json 2> gpurun_out/r05_bench_default.err; echo "bench rc $?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05_bench_default.json"))
print("value", d["value"], "ms", d["ms_per_step"], d["ms_per_step_each_loop"], "gap", d["ms_per_step_outside_kernels"], "frac", d["config"]["hbm_frac_whole_pass"])
print("gpu_state", {k: v for k, v in d["gpu_state"].items() if k not in ("dpm_levels_before_the_loops",)})
print("roofline", d["roofline"])
fe = d["file_end_to_end"]
print("plain", fe.get("value"), fe.get("in_process_device_ingest", {}).get("seconds_each_scan"))
print("cli", fe.get("cli_device_ingest"))
print("realistic", {k: v for k, v in fe.get("realistic", {}).items() if k not in ("kernels", "style")})
print("ingest_roofline", d.get("ingest_roofline"))
print("mixed", {k: v for k, v in d.get("mixed", {}).items() if k != "kernels"})
print("extra", d.get("extra_facets"))
print("all_facets", d.get("all_facets"))
PY
bash tools/r05_profiles.sh > gpurun_out/r05_profiles.log 2>&1; tail -3 gpurun_out/r05_profiles.log
