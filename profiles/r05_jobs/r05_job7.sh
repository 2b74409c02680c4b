#!/bin/bash
# round 5, GPU job 7: chunk size sweep, what the CRC costs the scan, decoders per CU -- on a 100 M-record aligner-style file and a 60 M plain one
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python tools/steady_scan.py --records 100000000 --style 3 --scans 5 --preread 2 --path /tmp/r.bam --keep
for mb in 384 640 768; do NGSQ_INGEST_RAW_MB=$mb python tools/steady_scan.py --records 100000000 --style 3 --scans 5 --path /tmp/r.bam --keep | sed "s/^/raw_mb=$mb  /"; done
NGSQ_CRC_SKIP=1 python tools/steady_scan.py --records 100000000 --style 3 --scans 5 --path /tmp/r.bam --keep | sed "s/^/crc_skip  /"
for pc in 22 23 25; do NGSQ_INFLATE_PER_CU=$pc python tools/steady_scan.py --records 100000000 --style 3 --scans 5 --path /tmp/r.bam --keep | sed "s/^/per_cu=$pc  /"; done
NGSQ_INFLATE_AHEAD=1 python tools/steady_scan.py --records 100000000 --style 3 --scans 5 --path /tmp/r.bam --keep | sed "s/^/ahead=1  /"
python tools/steady_scan.py --records 100000000 --style 3 --scans 5 --path /tmp/r.bam --keep | sed "s/^/again  /"
rm -f /tmp/r.bam
python tools/steady_scan.py --records 60000000 --style 0 --scans 5 --preread 2 --path /tmp/p.bam --keep
for mb in 256 384 768; do NGSQ_INGEST_RAW_MB=$mb python tools/steady_scan.py --records 60000000 --style 0 --scans 5 --path /tmp/p.bam --keep | sed "s/^/raw_mb=$mb  /"; done
NGSQ_CRC_SKIP=1 python tools/steady_scan.py --records 60000000 --style 0 --scans 5 --path /tmp/p.bam --keep | sed "s/^/crc_skip  /"
rm -f /tmp/p.bam
