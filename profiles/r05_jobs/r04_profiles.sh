#!/bin/bash
# Round 4's committed evidence, in one call on the GPU box:  bash tools/r04_profiles.sh   -> gpurun_out/r04_*  (copy into profiles/)
#   kernel stats + traffic (tools/profile_round.sh), SQ counter sets of the facet kernels incl. Edits and Genomic Features
#   (tools/pmc_sq.sh) and of the inflate on the plain and the aligner-style file (tools/pmc_inflate.sh)
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
bash tools/profile_round.sh r04 > gpurun_out/r04_profile_round.log 2>&1
bash tools/pmc_sq.sh --facets 0x7F --file-records 0 --h2d-batch 0 --extra-facet-legs 0 --mixed-records 0 > /dev/null 2>&1
{ echo "== facet kernels, per launch on 100 M x 150 bp reads sampled from the reference (bench.py --facets 0x7F): tools/pmc_sq.sh"; cat gpurun_out/sq_summary.txt; } > gpurun_out/r04_sq_counters.txt
bash tools/pmc_sq.sh --workload mixed > /dev/null 2>&1
{ echo; echo "== the 50-300 bp mixed-CIGAR workload (bench.py --workload mixed): tools/pmc_sq.sh --workload mixed"; cat gpurun_out/sq_summary.txt; } >> gpurun_out/r04_sq_counters.txt
{ echo; echo "== k_bgzf_inflate, one launch over 4 M records of the plain file (404 MB -> 1095 MB): tools/pmc_inflate.sh"; bash tools/pmc_inflate.sh 2>/dev/null; } >> gpurun_out/r04_sq_counters.txt
{ echo; echo "== k_bgzf_inflate, one launch over 4 M records of the aligner-style file (504 MB -> 1479 MB): INF_ARGS='--style 3' tools/pmc_inflate.sh"; INF_ARGS="--style 3" bash tools/pmc_inflate.sh 2>/dev/null; } >> gpurun_out/r04_sq_counters.txt
tail -30 gpurun_out/r04_sq_counters.txt
