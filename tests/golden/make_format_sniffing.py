#!/usr/bin/env python3
"""The reference's own format-sniffing tests (src/utils/formats.rs:192-322: file name -> detected format) and the
names its Display prints (:79-101), as a fixture (tests/golden/format_sniffing.json).  tests/test_cli.py runs this
build's `ngs qc` on every one of those file names.

    python tests/golden/make_format_sniffing.py /root/reference tests/golden/format_sniffing.json
"""
import json
import os
import re
import sys


def main():
    ref, dst = sys.argv[1], sys.argv[2]
    src = open(os.path.join(ref, "src/utils/formats.rs")).read()
    display = dict(re.findall(r'BioinformaticsFileFormat::(\w+) => write!\(f, "([^"]+)"\)', src))
    tests = src.split("#[cfg(test)]")[1]
    cases = re.findall(r'try_detect\("([^"]+)"\),\s*Some\(BioinformaticsFileFormat::(\w+)\)', tests)
    with open(dst, "w") as f:
        json.dump({"source": "stjude-rust-labs/ngs v0.4.0 src/utils/formats.rs:79-101,192-322", "display": display,
                   "cases": [{"file": a, "format": b} for a, b in cases]}, f, indent=1)
    print(dst, len(cases), "cases,", len(display), "formats")


if __name__ == "__main__":
    main()
