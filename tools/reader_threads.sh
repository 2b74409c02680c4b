#!/bin/bash
# File rate of ONE worker against the number of pread threads of its reader: what the host side must give a
# GPU to keep it at its rate, hence what an 8-GPU node delivers under a CPU quota.
#   bash tools/reader_threads.sh [RECORDS]     on the GPU box; STYLE=3 for the aligner-style file
# Second part: three workers sharing this box's GPU (`ngs qc --gpus 3 --same-device`), the same thread counts PER WORKER.
set -u
N=${1:-60000000}
R=$GRAFT_REPO_ROOT
cd $R
export TMPDIR=/tmp
bash tools/file_ab.sh $N "NGSQ_READER_THREADS=1" "NGSQ_READER_THREADS=2" "NGSQ_READER_THREADS=3" "NGSQ_READER_THREADS=4" "NGSQ_READER_THREADS=8" "NGSQ_READER_THREADS=14" 2>&1 | grep -v "ingest:"
echo "== ngs qc --gpus 3 --same-device, threads per worker (wall clock of the command, process start included)"
D=/tmp/rt_out; rm -rf $D; mkdir -p $D
for k in 1 2 4 8; do
  s=$(date +%s.%N)
  NGSQ_READER_THREADS=$k ./ngs_amd/ngs -q qc /tmp/ab.bam GRCh38_no_alt_AnalysisSet -o $D --gpus 3 --same-device || echo FAILED
  e=$(date +%s.%N)
  python3 -c "n=$N; dt=$e-$s; print('threads/worker $k: %.3f s = %.1f M records/s' % (dt, n/dt/1e6))"
done
