# phase switches of k_edits (measurement builds): bash tools/edits_exp.sh "0 1 2 3" [edits_time.py args]
set -u
cd $GRAFT_REPO_ROOT
EXPS=${1:-"0 1 2 3"}; shift || true
for e in $EXPS; do
  touch ngs_amd/csrc/edits_kernel.hip
  NGSQ_EXTRA_FLAGS="-DEDITS_EXP=$e" python -m ngs_amd.build > /tmp/b.log 2>&1 || tail -3 /tmp/b.log
  python tools/edits_time.py --tag "EDITS_EXP=$e" "$@"
done
touch ngs_amd/csrc/edits_kernel.hip; python -m ngs_amd.build > /dev/null 2>&1
