#!/bin/bash
# round 5, GPU job 4: parity of the fused / teardown / fields changes, then timings: Edits alone, all seven facets, the mixed shape,
# and the sweep of k_cov_stream on a CU-masked stream of its own
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
echo "== parity"
timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_cli.py tests/test_stager.py tests/test_cov_stream_gpu.py -q -m gpu -k "not full_size" > gpurun_out/r05_pytest_job4.log 2>&1; echo "pytest rc $?"; grep -n "passed\|failed" gpurun_out/r05_pytest_job4.log | tail -3
echo "== edits alone"
for a in "" "--subst 0.25" "--iid" "--aligner" "--mixed"; do python tools/edits_time.py $a --tag "r05c $a"; done 2>&1 | grep k_edits
echo "== all seven facets / mixed"
B="--steps 20 --warmup 3 --repeats 1 --cpu-sample 0 --h2d-batch 0 --file-records 0 --extra-facet-legs 0 --live-traffic 0"
python bench.py $B --mixed-steps 10 > gpurun_out/af_job4.json 2>/dev/null
python - <<'PY'
import json
d = json.load(open("gpurun_out/af_job4.json"))
a, m = d["all_facets"], d["mixed"]
print("all_facets", a.get("ms_per_step"), a.get("ms_per_step_each_loop"), a.get("parity_check"), a.get("ms_per_step_outside_kernels"), {k: v["avg_ms"] for k, v in a.get("kernels", {}).items()})
print("mixed", m.get("ms_per_step"), m.get("parity_check"), m.get("hbm_frac_whole_pass"), {k: v["avg_ms"] for k, v in m.get("kernels", {}).items()})
print("headline", d["ms_per_step"], d["gpu_state"].get("sclk_mhz"), d["gpu_state"].get("power_w"), d["gpu_state"].get("busy_pct"), d["gpu_state"]["source"])
PY
echo "== k_cov_stream on a CU-masked stream"
H="--steps 60 --warmup 5 --repeats 1 --cpu-sample 0 --mixed-records 0 --h2d-batch 0 --file-records 0 --extra-facet-legs 0 --all-facets-records 0 --live-traffic 0"
run() { env $1 python bench.py $H > gpurun_out/cu_$2.json 2>/dev/null; python - <<PY
import json
d = json.load(open("gpurun_out/cu_$2.json"))
print("%-52s" % "$1", d["ms_per_step"], d["parity_check"], {k: v["avg_ms"] for k, v in d["kernels"].items()})
PY
}
for round in 1 2; do
run "NGSQ_COV_SIDE_CUS=0" base_$round
for n in 16 32 64 96; do
run "NGSQ_COV_SIDE_CUS=$n" side${n}_$round
run "NGSQ_COV_SIDE_CUS=$n NGSQ_COV_MAIN_COMPLEMENT=1" side${n}c_$round
done
done
