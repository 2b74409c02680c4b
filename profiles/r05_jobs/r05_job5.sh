#!/bin/bash
# round 5, GPU job 5: the ingest after the memsets went (parity first), then the decoders kept off N compute units
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
echo "== parity (ingest)"
timeout 1500 python -m pytest tests/test_device_ingest_gpu.py tests/test_hand_bam.py tests/test_bam_ingest.py -q -m gpu > gpurun_out/r05_pytest_job5.log 2>&1; echo "pytest rc $?"; grep -n "passed\|failed" gpurun_out/r05_pytest_job5.log | tail -3
echo "== plain 40 M / realistic 40 M, scans after two settling reads"
python tools/steady_scan.py --records 40000000 --style 0 --scans 6 --preread 2 --path /tmp/p.bam --keep
python tools/steady_scan.py --records 40000000 --style 3 --scans 6 --preread 2 --path /tmp/r.bam --keep
for n in 8 16 32; do
  echo "-- NGSQ_INFLATE_CU_EXCLUDE=$n"
  NGSQ_INFLATE_CU_EXCLUDE=$n python tools/steady_scan.py --records 40000000 --style 0 --scans 5 --path /tmp/p.bam --keep
  NGSQ_INFLATE_CU_EXCLUDE=$n python tools/steady_scan.py --records 40000000 --style 3 --scans 5 --path /tmp/r.bam --keep
done
echo "-- again without"
python tools/steady_scan.py --records 40000000 --style 0 --scans 5 --path /tmp/p.bam --keep
python tools/steady_scan.py --records 40000000 --style 3 --scans 5 --path /tmp/r.bam --keep
echo "-- 512 MiB chunks / 1 GiB chunks on the realistic file"
NGSQ_INGEST_RAW_MB=512 python tools/steady_scan.py --records 40000000 --style 3 --scans 5 --path /tmp/r.bam --keep
NGSQ_INGEST_RAW_MB=1024 python tools/steady_scan.py --records 40000000 --style 3 --scans 5 --path /tmp/r.bam --keep
rm -f /tmp/p.bam /tmp/r.bam
