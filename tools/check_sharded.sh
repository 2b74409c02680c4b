# One BAM file, several ranks (`ngs qc --gpus 3 --same-device` through tools/qc_sharded.py) against the single-process `ngs qc`: same JSON.
# Run on the GPU box:  bash tools/check_sharded.sh [records]   (the workers share GPU 0 and exchange through shared memory)
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
N=${1:-4000000}
D=/tmp/shard_check; rm -rf $D; mkdir -p $D
for kind in fixed mixed; do
  python tools/make_bam.py $D/$kind.bam --records $N $([ $kind = mixed ] && echo --mixed) | tail -n 1
  ./ngs_amd/ngs -q qc $D/$kind.bam GRCh38_no_alt_AnalysisSet -o $D/cli_$kind || echo "CLI FAILED"
  ./ngs_amd/ngs -q qc $D/$kind.bam GRCh38_no_alt_AnalysisSet -o $D/cli_array_$kind --coverage array || echo "CLI FAILED"
  ./ngs_amd/ngs -q qc $D/$kind.bam GRCh38_no_alt_AnalysisSet -o $D/cli_host_$kind --coverage array --ingest host || echo "CLI FAILED"
  python - <<PY
import json, sys
sys.path.insert(0, "$GRAFT_REPO_ROOT")
from tests.util import json_equal
a = json.load(open("$D/cli_$kind/$kind.bam.results.json"))
for other in ("cli_array", "cli_host"):
    b = json.load(open("$D/%s_$kind/$kind.bam.results.json" % other))
    try:
        json_equal(a, b); print("$kind: cli stream ==", other)
    except AssertionError as e:
        print("$kind: cli stream !=", other, str(e)[:200])
PY
  for cov in stream array; do
    python tools/qc_sharded.py $D/$kind.bam GRCh38_no_alt_AnalysisSet -o $D/sh_${kind}_$cov --gpus 3 --same-device --coverage $cov 2>&1 | tail -n 1
    python - <<PY
import json, sys
sys.path.insert(0, "$GRAFT_REPO_ROOT")
from tests.util import json_equal
a = json.load(open("$D/cli_$kind/$kind.bam.results.json")); b = json.load(open("$D/sh_${kind}_$cov/$kind.bam.results.json"))
try:
    json_equal(a, b); print("$kind $cov: 3 workers == single process")
except AssertionError as e:
    print("$kind $cov: MISMATCH", str(e)[:200])
PY
  done
done
