#!/bin/bash
# rocprofv3 per-kernel summary of the file path: `ngs qc` with device ingest on a synthetic BAM
# (run on the GPU box via gpurun).   usage: bash tools/profile_ingest.sh r01 [records]
set -u
P=${1:-r01}
N=${2:-8000000}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_ingest_$P
mkdir -p $O
export TMPDIR=/tmp
python3 $R/tools/make_bam.py /tmp/prof.bam --records $N > $O/make.log 2>&1
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/stats -o out --output-format csv -- $R/ngs_amd/ngs -q qc /tmp/prof.bam GRCh38_no_alt_AnalysisSet -o /tmp --ingest device > $O/run.log 2>&1
cd $R
S=$(find $O/stats -name '*kernel_stats.csv' | head -n 1)
cp "$S" gpurun_out/${P}_ingest_kernel_stats.csv
head -n 16 "$S" | cut -c1-170
