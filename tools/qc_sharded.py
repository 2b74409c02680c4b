#!/usr/bin/env python3
"""`ngs qc` of ONE BAM file on several GPUs of a node.  This is a launcher only: the work is
`ngs qc --gpus N` (ngs_amd/csrc/cli/ngs_main.cpp): one worker process per GPU, each ingesting its BGZF
block range on its own device (ngsq_bam_shard_open), one ngsq_exchange over RCCL before the teardown
(include/ngsq_comm.h), rank 0 writes <out>/<bam name>.results.json.

    python tools/qc_sharded.py --gpus N sample.bam GRCh38_no_alt_AnalysisSet [-o DIR] [ngs qc options ...]
    (a one-GPU box: add --same-device -- the workers share the device and exchange through shared memory)
"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from ngs_amd import build
    build.build(verbose=False)
    ngs = build.build_cli(verbose=False)
    args = sys.argv[1:]
    t0 = time.perf_counter()
    rc = subprocess.run([ngs, "-q", "qc"] + args).returncode
    dt = time.perf_counter() - t0
    if rc == 0:
        pos = [a for a in args if not a.startswith("-")]
        out_dir = args[args.index("-o") + 1] if "-o" in args else "."
        bam = next(a for a in pos if a.endswith(".bam"))
        with open(os.path.join(out_dir, os.path.basename(bam) + ".results.json")) as f:
            total = json.load(f)["general"]["records"]["total"]
        print(json.dumps({"records": total, "seconds": round(dt, 3), "records_per_s": round(total / dt)}))
    return rc


if __name__ == "__main__":
    sys.exit(main())
