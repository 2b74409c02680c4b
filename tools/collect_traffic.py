#!/usr/bin/env python3
"""Summarise rocprofv3 output into profiles/: per-kernel HBM traffic from the PMC passes.

    python tools/collect_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> \
        <kernel_stats.csv> <out_prefix>

Counters are collected in SEPARATE passes (FETCH_SIZE needs 3 of the 4 TCC slots,
WRITE_SIZE 2: MI355X_MICROARCH.md "rocprofv3 PMC slots").  Units: KiB per
dispatch.  gfx950 correction (same guide, section HBM): FETCH_SIZE reports exactly
half of the bytes of a wide (16 B/lane) coalesced streaming read, so the read
side is doubled; WRITE_SIZE is exact for 16 B/lane streaming stores.  Both the
raw and the corrected figures are written.
"""
import csv
import json
import shutil
import sys
from collections import defaultdict


def per_kernel(path, counter):
    acc = defaultdict(lambda: [0.0, 0])
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            a = acc[row["Kernel_Name"]]
            a[0] += float(row["Counter_Value"])
            a[1] += 1
    return {k: (v[0] / v[1] * 1024.0, v[1]) for k, v in acc.items() if v[1]}


def main():
    fetch_csv, write_csv, stats_csv, out = sys.argv[1:5]
    workload = sys.argv[5] if len(sys.argv) > 5 else "python bench.py (default: 100 M x 150 bp records per GPU, all default facets, N=1)"
    fetch = per_kernel(fetch_csv, "FETCH_SIZE")
    write = per_kernel(write_csv, "WRITE_SIZE")
    doc = {"workload": workload,
           "units": "bytes per dispatch (mean over dispatches)",
           "correction": "read side = 2 x FETCH_SIZE x 1024 (gfx950, MI355X_MICROARCH.md section HBM); write side = WRITE_SIZE x 1024",
           "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        if "ngsq::" not in k:
            continue
        fr, nf = fetch.get(k, (0.0, 0))
        wr, nw = write.get(k, (0.0, 0))
        short = k.split("(")[0].replace("void ", "")
        doc["kernels"][short] = {"fetch_raw": round(fr), "write_raw": round(wr), "dispatches": max(nf, nw),
                                 "hbm_bytes_corrected": round(2 * fr + wr)}
    with open(out + "_traffic.json", "w") as f:
        json.dump(doc, f, indent=1)
    shutil.copyfile(stats_csv, out + "_kernel_stats.csv")
    print(json.dumps(doc, indent=1))


if __name__ == "__main__":
    main()
