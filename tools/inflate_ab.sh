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
  echo "== $v: $(grep -h k_bgzf_inflate $(find /tmp/ab_$v -name '*kernel_stats.csv') | cut -d, -f2-4,6,7 | head -1)  $(grep -c matches /tmp/ab_$v.log)"
  grep -h "inflate-prof" /tmp/ab_$v.log | tail -9
done
cp /tmp/inflate_tree.hip ngs_amd/csrc/bgzf_inflate.hip; touch ngs_amd/csrc/bgzf_inflate.hip; python -m ngs_amd.build > /dev/null 2>&1
