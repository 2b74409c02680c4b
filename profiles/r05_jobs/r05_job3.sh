#!/bin/bash
# round 5, GPU job 3: GC tallied by the Edits kernel, Edits teardown only where Edits wrote, Genomic Features on a side stream:
# parity first, then the all-seven-facets pass under each switch
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
echo "== sysfs"; ls -l /sys/class/drm/ 2>/dev/null | grep -c card; for c in /sys/class/drm/card[0-9]*/device; do echo "$c -> $(readlink -f $c)"; done 2>/dev/null | head -12
python - <<'PY'
import ctypes as C, glob, os, sys
sys.path.insert(0, os.getcwd())
from ngs_amd import ffi
lib = ffi.load_library()
b = C.create_string_buffer(64); print("hip device 0 pci:", lib.ngsq_device_pci_bus_id(0, b, 64), b.value)
for c in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
    if os.path.realpath(c).lower().endswith(b.value.decode()):
        for f in ("pp_dpm_sclk", "pp_dpm_mclk", "gpu_busy_percent", "current_compute_partition"):
            try: print(c, f, open(os.path.join(c, f)).read().replace("\n", " | "))
            except OSError as e: print(c, f, e)
PY
echo "== parity"
timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_cli.py tests/test_stager.py -x -q -m gpu -k "edits or gc_content or mixed or features or facet or hand_golden or reset or sharded or reference_call_shape" > gpurun_out/r05_pytest_job3.log 2>&1; echo "pytest rc $?"; grep -n "passed\|failed" gpurun_out/r05_pytest_job3.log | tail -3
echo "== edits alone"
for a in "" "--subst 0.05" "--subst 0.25" "--iid" "--aligner"; do python tools/edits_time.py $a --tag "r05b $a"; done 2>&1 | grep k_edits
echo "== all seven facets"
B="--steps 5 --warmup 2 --repeats 1 --cpu-sample 0 --mixed-records 0 --h2d-batch 0 --file-records 0 --extra-facet-legs 0 --live-traffic 0"
run() { env $1 python bench.py $B > gpurun_out/af_$2.json 2>/dev/null; python - <<PY
import json
d = json.load(open("gpurun_out/af_$2.json"))["all_facets"]
print("%-34s" % "$1", d.get("ms_per_step"), d.get("ms_per_step_each_loop"), d.get("parity_check"), {k: v["avg_ms"] for k, v in d.get("kernels", {}).items()}) if "failed" not in d else print("$1", d)
PY
}
for round in 1 2; do
run "NGSQ_EDITS_NO_GC=1" nogc_$round
run "NGSQ_EDITS_NO_GC=0" fused_$round
run "NGSQ_FEATURES_SIDE=1" side_$round
run "NGSQ_FEATURES_SIDE=1 NGSQ_EDITS_NO_GC=1" side_nogc_$round
done
echo "== four waves per SIMD for the GC variant"
touch ngs_amd/csrc/edits_kernel.hip; NGSQ_EXTRA_FLAGS=-DNGSQ_EDR_WAVES=4 python -m ngs_amd.build > /dev/null 2>&1
run "NGSQ_EDITS_NO_GC=0" fused_w4
run "NGSQ_FEATURES_SIDE=1" side_w4
