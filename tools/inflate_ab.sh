# A/B of two versions of csrc/bgzf_inflate.hip on one box: kernel time (rocprofv3) and, with PROFILE=1, the phase split.
#   bash tools/inflate_ab.sh tools/_exp/inflate_v1.hip [records]      (B = the file in the tree)
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OTHER=$1
N=${2:-4000000}
cp ngs_amd/csrc/bgzf_inflate.hip /tmp/inflate_tree.hip
for v in other tree other tree; do
  if [ $v = other ]; then cp $OTHER ngs_amd/csrc/bgzf_inflate.hip; else cp /tmp/inflate_tree.hip ngs_amd/csrc/bgzf_inflate.hip; fi
  touch ngs_amd/csrc/bgzf_inflate.hip
  NGSQ_EXTRA_FLAGS="${FLAGS:-}" python -m ngs_amd.build > /tmp/build_$v.log 2>&1 || { tail -5 /tmp/build_$v.log; }
  rm -rf /tmp/ab_$v
  (cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/ab_$v -o out --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/bench_inflate.py --records $N --reps 2 > /tmp/ab_$v.log 2>&1)
  python3 - $v <<'PY'
import csv, glob, sys
for f in glob.glob('/tmp/ab_%s/**/*kernel_stats.csv' % sys.argv[1], recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_bgzf_inflate' in r['Name'] or 'k_bgzf_crc' in r['Name']:
            print('== %s: %s calls %s avg %.3f ms min %.3f max %.3f' % (sys.argv[1], r['Name'].split('(')[0][-20:], r['Calls'], float(r['AverageNs'])/1e6, float(r['MinNs'])/1e6, float(r['MaxNs'])/1e6))
PY
  grep -h "inflate-prof" /tmp/ab_$v.log | tail -9
done
cp /tmp/inflate_tree.hip ngs_amd/csrc/bgzf_inflate.hip; touch ngs_amd/csrc/bgzf_inflate.hip; python -m ngs_amd.build > /dev/null 2>&1
