#!/bin/bash
# round 4: the rewritten Edits kernel -- parity tests, the Edits / Features fuzz, kernel time on reads sampled from the reference
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r04_edits
mkdir -p $O
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_cli.py tests/test_shard_gloo.py -x -q -m gpu -k "edits or Edits or reference_bases or golden or three_ranks or sharded_state" > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/summary.txt
tail -15 $O/pytest.log
timeout 900 python tools/fuzz_parity.py --seeds 0 --extra ${EXTRA:-60} > $O/fuzz_extra.log 2>&1; echo "fuzz extra rc=$?" | tee -a $O/summary.txt
tail -3 $O/fuzz_extra.log
python - <<'PY' 2>&1 | tee $O/extra_leg.json
import json, sys
sys.path.insert(0, ".")
import numpy as np
import bench
from ngs_amd import ffi, host
lib = ffi.load_library()
print(json.dumps(bench.leg_extra_facets(lib, host, ffi, np, 100_000_000)))
PY
