#!/bin/bash
# round 5, GPU job 13: the offsets layout through the window lanes (k_edits_rows<.., RAGGED>): parity, fuzz, timings beside the lane-per-record kernel
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_parity_gpu.py tests/test_stager.py tests/test_cli.py tests/test_hand_bam.py -x -q -m gpu -k "not full_size" > gpurun_out/r05_pytest_job13.log 2>&1; echo "pytest rc $?"; grep -n "passed\|failed" gpurun_out/r05_pytest_job13.log | tail -3
for a in "--mixed" "--mixed --subst 0.05" "--mixed --subst 0.25" "--mixed --iid" "" "--aligner"; do python tools/edits_time.py $a --tag "r05g $a"; done 2>&1 | grep k_edits
echo "== lane per record (NGSQ_EDITS_PER_RECORD=1)"
for a in "--mixed" "--mixed --subst 0.05"; do NGSQ_EDITS_PER_RECORD=1 python tools/edits_time.py $a --tag "r05g per-record $a"; done 2>&1 | grep k_edits
timeout 1200 python tools/fuzz_parity.py --seeds 60 --extra 200 > gpurun_out/r05_fuzz_job13.log 2>&1; echo "fuzz rc $?"; tail -2 gpurun_out/r05_fuzz_job13.log
B="--steps 10 --warmup 3 --repeats 1 --cpu-sample 0 --file-records 0 --extra-facet-legs 0 --all-facets-records 0 --live-traffic 0 --mixed-records 0"
python bench.py $B > gpurun_out/stager_job13.json 2>/dev/null
python - <<'PY'
import json
d = json.load(open("gpurun_out/stager_job13.json"))
print("h2d", d.get("h2d_inclusive")); print("stager", d.get("stager"))
PY
