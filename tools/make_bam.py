#!/usr/bin/env python3
"""Write a synthetic coordinate-sorted BAM (+ .bai) of SURVEY.md 8(d)'s record distribution.
    python tools/make_bam.py OUT.bam [--records N] [--level L] [--mixed]
"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ngs_amd import build, ffi, host  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--records", type=int, default=4_000_000)
    ap.add_argument("--level", type=int, default=6)
    ap.add_argument("--mixed", action="store_true")
    ap.add_argument("--style", type=int, default=0, help="ngsq_shared.h NGSQ_SYNTH_FILE_*: 3 = an aligner's names, tags and CIGAR mix")
    a = ap.parse_args()
    build.build(verbose=False)
    lib = ffi.load_library()
    cfg = host.synth_config(a.records, mode=ffi.SYNTH_MIXED if a.mixed else ffi.SYNTH_FIXED, file_style=a.style)
    assert lib.ngsq_synth_write_bam(C.byref(cfg), a.out.encode(), a.records, a.level, 0) == 0
    print(a.out, os.path.getsize(a.out))


if __name__ == "__main__":
    main()
