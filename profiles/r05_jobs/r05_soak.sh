#!/bin/bash
# round 5: a long randomised parity sweep of the final tree (offsets-layout Edits, k_features, the teardown): every sweep of tools/fuzz_parity.py
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 2400 python tools/fuzz_parity.py --seeds 400 --extra 900 --sorted 150 --ingest 150 > gpurun_out/r05_fuzz_soak.log 2>&1; echo "soak rc $?"
grep -c " ok" gpurun_out/r05_fuzz_soak.log; grep -v " ok$" gpurun_out/r05_fuzz_soak.log | tail -15
