#!/bin/bash
# Round 5's committed evidence, in one call on the GPU box:  bash tools/r05_profiles.sh   -> gpurun_out/r05_*  (copy into profiles/)
#   kernel stats + traffic (tools/profile_round.sh: default pass, mixed, all seven facets, the file path), SQ counter sets of the facet
#   kernels incl. the Edits kernel that tallies GC Content (tools/pmc_sq.sh --facets 0x7F)
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
bash tools/profile_round.sh r05 > gpurun_out/r05_profile_round.log 2>&1
bash tools/pmc_sq.sh --facets 0x7F --file-records 0 --h2d-batch 0 --extra-facet-legs 0 --mixed-records 0 --all-facets-records 0 --repeats 1 > /dev/null 2>&1
{ echo "== facet kernels, per launch on 100 M x 150 bp reads sampled from the reference (bench.py --facets 0x7F): tools/pmc_sq.sh"; cat gpurun_out/sq_summary.txt; } > gpurun_out/r05_sq_counters.txt
tail -5 gpurun_out/r05_profile_round.log; ls gpurun_out/r05_*
