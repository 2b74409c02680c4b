#!/bin/bash
# round 5, GPU job 1: (0) parity of the changed reset/finalize path, (1) A/B of the time a step spends outside its kernels,
# (2) are the scans of one file alike, and is it the page cache or the block cache when they are not
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_cov_stream_gpu.py -x -q -m gpu 2>&1 | tail -4
B="--steps 100 --warmup 10 --cpu-sample 0 --mixed-records 0 --h2d-batch 0 --file-records 0 --extra-facet-legs 0 --live-traffic 0"
for round in 1 2; do for leg in 1 0; do
  NGSQ_STEP_LEGACY=$leg python bench.py $B > gpurun_out/gap_legacy${leg}_$round.json
  python - <<PY
import json
d = json.load(open("gpurun_out/gap_legacy${leg}_$round.json"))
k = d["kernels"]; s = sum(v["avg_ms"] for v in k.values())
print("legacy=$leg round $round: ms_per_step %.3f  sum of kernels %.3f  gap %.3f  qual %.3f" % (d["ms_per_step"], s, d["ms_per_step"] - s, k["qual"]["avg_ms"]), d["parity_check"])
PY
done; done
python tools/steady_scan.py --records 30000000 --style 3 --scans 6 --path /tmp/a.bam
python tools/steady_scan.py --records 30000000 --style 3 --scans 6 --preread 2 --path /tmp/b.bam
NGSQ_POOL_MB=16384 python tools/steady_scan.py --records 30000000 --style 3 --scans 6 --path /tmp/c.bam
