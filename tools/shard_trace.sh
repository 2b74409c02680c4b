# Stage timeline (NGSQ_INGEST_TRACE=1) of `ngs qc` with 1 and 3 workers on one file.  bash tools/shard_trace.sh [records]
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
N=${1:-60000000}
D=/tmp/shard_rate; mkdir -p $D
[ -f $D/f.bam ] || python tools/make_bam.py $D/f.bam --records $N | tail -n 1
sync
./ngs_amd/ngs -q qc $D/f.bam GRCh38_no_alt_AnalysisSet -o $D/g1 >/dev/null 2>&1
echo "== 1 worker"; NGSQ_INGEST_TRACE=1 ./ngs_amd/ngs -q qc $D/f.bam GRCh38_no_alt_AnalysisSet -o $D/g1 2>&1 | grep "\[ngs\]"
echo "== 3 workers"; NGSQ_INGEST_TRACE=1 ./ngs_amd/ngs -q qc $D/f.bam GRCh38_no_alt_AnalysisSet -o $D/g3 --gpus 3 --same-device 2>&1 | grep "\[ngs\]\|first pinned"
