#!/bin/bash
# round 5, GPU job 16: the CIGAR offsets loaded a pass ahead in k_edits_rows<CIG_OFF>: A/B on one box (rebuild with -DNGSQ_EDR_CB_LATE)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
run() { for a in "--mixed" "--aligner" "--mixed --subst 0.05"; do python tools/edits_time.py $a --tag "$1 $a"; done 2>&1 | grep k_edits; }
echo "== ahead (as committed)"; run ahead
NGSQ_EXTRA_FLAGS=-DNGSQ_EDR_CB_LATE python -m ngs_amd.build --force > gpurun_out/j16_build.log 2>&1; echo "build rc $?"
echo "== late"; run late
python -m ngs_amd.build --force > gpurun_out/j16_build2.log 2>&1; echo "build rc $?"
echo "== ahead again"; run ahead
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "edits" 2>&1 | tail -2
