# One file, 1 vs 3 workers sharing this box's GPU (`ngs qc --gpus 3 --same-device`): wall clock of the command.
# per-worker rate within 15 % of 1/3 of the one-worker rate on the same file.
# Run on the GPU box:  bash tools/shard_rate.sh [records]
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
N=${1:-60000000}
D=/tmp/shard_rate; rm -rf $D; mkdir -p $D
python tools/make_bam.py $D/f.bam --records $N | tail -n 1
sync
for rep in 1 2; do
  for g in 1 3; do
    s=$(date +%s.%N)
    if [ $g = 1 ]; then ./ngs_amd/ngs -q qc $D/f.bam GRCh38_no_alt_AnalysisSet -o $D/g$g || echo FAILED
    else ./ngs_amd/ngs -q qc $D/f.bam GRCh38_no_alt_AnalysisSet -o $D/g$g --gpus $g --same-device || echo FAILED; fi
    e=$(date +%s.%N)
    python -c "n=$N; dt=$e-$s; print('gpus $g: %.3f s = %.1f M records/s (%.1f M per worker)' % (dt, n/dt/1e6, n/dt/1e6/$g))"
  done
done
cmp $D/g1/f.bam.results.json $D/g3/f.bam.results.json && echo "documents identical"
echo "== with NGSQ_QUICK_EXIT=1"
for g in 1 3; do
  s=$(date +%s.%N)
  if [ $g = 1 ]; then NGSQ_QUICK_EXIT=1 ./ngs_amd/ngs -q qc $D/f.bam GRCh38_no_alt_AnalysisSet -o $D/g$g || echo FAILED
  else NGSQ_QUICK_EXIT=1 ./ngs_amd/ngs -q qc $D/f.bam GRCh38_no_alt_AnalysisSet -o $D/g$g --gpus $g --same-device || echo FAILED; fi
  e=$(date +%s.%N)
  python -c "n=$N; dt=$e-$s; print('gpus $g: %.3f s = %.1f M records/s (%.1f M per worker)' % (dt, n/dt/1e6, n/dt/1e6/$g))"
done
