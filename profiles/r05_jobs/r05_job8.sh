#!/bin/bash
# round 5, GPU job 8: parity of the scan kernels and of k_edits' second segment, Edits on the mixed shape, the ingest after the scan went
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_device_ingest_gpu.py tests/test_hand_bam.py tests/test_cli.py -x -q -m gpu -k "not full_size" > gpurun_out/r05_pytest_job8.log 2>&1; echo "pytest rc $?"; grep -n "passed\|failed" gpurun_out/r05_pytest_job8.log | tail -3
for a in "" "--aligner" "--mixed" "--mixed --subst 0.05"; do python tools/edits_time.py $a --tag "r05d $a"; done 2>&1 | grep k_edits
timeout 900 python tools/fuzz_parity.py --extra 150 > gpurun_out/r05_fuzz_extra.log 2>&1; echo "fuzz extra rc $?"; tail -3 gpurun_out/r05_fuzz_extra.log
timeout 900 python tools/fuzz_parity.py --ingest 120 > gpurun_out/r05_fuzz_ingest.log 2>&1; echo "fuzz ingest rc $?"; tail -3 gpurun_out/r05_fuzz_ingest.log
