#!/bin/bash
# round 5, GPU job 21: k_features -- blocks per CU and records per lane, again, on the kernel of this round
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for pc in 6 8 10 12 14 16 20 24 28; do NGSQ_FEATURES_BLOCKS_PER_CU=$pc python tools/features_sweep.py --tag "per_cu=$pc"; done
for e in 2 3 6 8; do
  NGSQ_EXTRA_FLAGS="-DNGSQ_FEATURES_RPT=$e" python -m ngs_amd.build --force > /tmp/b.log 2>&1 || tail -3 /tmp/b.log
  for pc in 8 12 16; do NGSQ_FEATURES_BLOCKS_PER_CU=$pc python tools/features_sweep.py --tag "rpt=$e per_cu=$pc"; done
done
