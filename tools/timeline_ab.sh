# kernel timeline of the in-process file path under env settings:  bash tools/timeline_ab.sh RECORDS "ENV=a" "ENV=b" ...
set -u
N=$1; shift
cd /tmp && export TMPDIR=/tmp
for V in "$@"; do
  rm -rf /tmp/tl
  env $V rocprofv3 --kernel-trace -d /tmp/tl -o out --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/file_timeline.py run $N > /tmp/tl.log 2>&1
  echo "=== $V"; tail -1 /tmp/tl.log
  python3 $GRAFT_REPO_ROOT/tools/file_timeline.py show /tmp/tl | cut -c1-150
done
