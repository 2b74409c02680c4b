#!/bin/bash
# round 5, GPU job 2: Edits on reads that differ from the reference (the histogram fix), the whole -m gpu suite, the driver-shaped bench
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for a in "" "--subst 0.05" "--subst 0.25" "--iid" "--aligner"; do python tools/edits_time.py $a --tag "r05a $a"; done 2>&1 | grep -v "^HIP\|^ROCm\|^Hostname\|^Librccl"
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r05_pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -5 gpurun_out/r05_pytest_gpu.log
timeout 1500 python bench.py > gpurun_out/r05_bench_a.json 2> gpurun_out/r05_bench_a.err; echo "bench rc $?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05_bench_a.json"))
print("value", d["value"], "ms", d["ms_per_step"], d["ms_per_step_each_loop"], "gap", d["ms_per_step_outside_kernels"])
print("gpu_state", d["gpu_state"])
print("roofline", d["roofline"])
fe = d["file_end_to_end"]
print("plain", fe.get("value"), fe.get("in_process_device_ingest", {}).get("seconds_each_scan"), fe.get("page_cache_settling_reads_GB_per_s"))
print("cli", fe.get("cli_device_ingest"))
print("realistic", {k: v for k, v in fe.get("realistic", {}).items() if k != "kernels"})
print("cold", fe.get("cold_cache"))
print("mixed", {k: v for k, v in d.get("mixed", {}).items() if k != "kernels"})
print("extra", d.get("extra_facets"))
print("all_facets", d.get("all_facets"))
PY
