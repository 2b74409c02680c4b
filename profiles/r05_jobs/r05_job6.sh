#!/bin/bash
# round 5, GPU job 6: the whole -m gpu suite on HEAD; k_fields' offsets with the tile or not (mixed leg); chunk size on a 100 M-record aligner-style file
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r05_pytest_gpu.log 2>&1; echo "pytest rc $?"; grep -n "passed\|failed" gpurun_out/r05_pytest_gpu.log | tail -3
B="--steps 10 --warmup 3 --repeats 1 --cpu-sample 0 --h2d-batch 0 --file-records 0 --extra-facet-legs 0 --all-facets-records 0 --live-traffic 0 --mixed-steps 20"
mix() { python bench.py $B > gpurun_out/mix_$1.json 2>/dev/null; python - <<PY
import json
m = json.load(open("gpurun_out/mix_$1.json"))["mixed"]
print("%-24s" % "$1", m["ms_per_step"], m["hbm_frac_whole_pass"], m["parity_check"], {k: v["avg_ms"] for k, v in m["kernels"].items()})
PY
}
mix prefetch_1; mix prefetch_2
touch ngs_amd/csrc/fields_kernel.hip; NGSQ_EXTRA_FLAGS=-DNGSQ_FT_PREFETCH_OFFS=0 python -m ngs_amd.build > /dev/null 2>&1
mix noprefetch_1; mix noprefetch_2
touch ngs_amd/csrc/fields_kernel.hip; python -m ngs_amd.build > /dev/null 2>&1
echo "== chunk size, 100 M aligner-style records"
python tools/steady_scan.py --records 100000000 --style 3 --scans 5 --preread 2 --path /tmp/r.bam --keep
NGSQ_INGEST_RAW_MB=512 python tools/steady_scan.py --records 100000000 --style 3 --scans 5 --path /tmp/r.bam --keep
NGSQ_INGEST_RAW_MB=1024 python tools/steady_scan.py --records 100000000 --style 3 --scans 5 --path /tmp/r.bam --keep
NGSQ_INGEST_RAW_MB=256 python tools/steady_scan.py --records 100000000 --style 3 --scans 5 --path /tmp/r.bam --keep
rm -f /tmp/r.bam
