#!/bin/bash
# round 5, GPU job 26: the decoder's phase split on aligner-style blocks beside plain ones (-DNGSQ_INFLATE_PROFILE)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
NGSQ_EXTRA_FLAGS=-DNGSQ_INFLATE_PROFILE python -m ngs_amd.build --force > gpurun_out/j26_build.log 2>&1; echo "build rc $?"
for st in 0 3; do echo "== style $st"; python tools/bench_inflate.py --records 3000000 --style $st --reps 2 2>&1 | grep -v "^\[inflate-prof\].*0.00 %" | tail -16; done
python -m ngs_amd.build --force > /dev/null 2>&1
for st in 0 3; do echo "== style $st, kernel time"; (cd /tmp && rm -rf /tmp/p26 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p26 -o out -- python3 $GRAFT_REPO_ROOT/tools/bench_inflate.py --records 3000000 --style $st --reps 2 > /tmp/p26.log 2>&1; grep compressed /tmp/p26.log | tail -1; python3 - <<'PY'
import csv, glob
for f in glob.glob('/tmp/p26/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_bgzf' in r['Name']: print('   %-18s calls %s avg %.3f ms' % (r['Name'].split('(')[0][-16:], r['Calls'], float(r['AverageNs'])/1e6))
PY
); done
