#!/bin/bash
# round 4, first GPU call: the new ingest tests, the ingest fuzz with dressed / adversarial records, and the file rate on a
# plain and on an aligner-style file of the same records
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r04_first
mkdir -p $O
timeout 900 python -m pytest tests/test_device_ingest_gpu.py -x -q -m gpu > $O/pytest_ingest.log 2>&1; echo "pytest ingest rc=$?" | tee -a $O/summary.txt
tail -5 $O/pytest_ingest.log
timeout 600 python tools/fuzz_parity.py --seeds 0 --ingest 60 > $O/fuzz_ingest.log 2>&1; echo "fuzz ingest rc=$?" | tee -a $O/summary.txt
tail -3 $O/fuzz_ingest.log
KERNELS=1 STYLE=0 bash tools/file_ab.sh 24000000 "" > $O/file_plain.log 2>&1; tail -6 $O/file_plain.log
KERNELS=1 STYLE=3 bash tools/file_ab.sh 24000000 "" > $O/file_real.log 2>&1; tail -6 $O/file_real.log
KERNELS=1 STYLE=1 bash tools/file_ab.sh 24000000 "" > $O/file_aligner.log 2>&1; tail -6 $O/file_aligner.log
