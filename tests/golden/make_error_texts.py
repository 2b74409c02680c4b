#!/usr/bin/env python3
"""Collect the error texts of the reference's `ngs qc` path (the literals of bail! / anyhow! / with_context in the files
the command runs through) into a fixture (tests/golden/qc_error_texts.json).  `{}` marks a formatted value.
tests/test_cli.py holds this build's messages to them.

    python tests/golden/make_error_texts.py /root/reference tests/golden/qc_error_texts.json
"""
import json
import os
import re
import sys

FILES = ["src/qc/command.rs", "src/qc.rs", "src/utils/formats/bam.rs", "src/qc/sequence_based/edits.rs",
         "src/utils/alignment.rs", "src/qc/record_based/features.rs", "src/utils/formats/gff.rs"]


def main():
    ref, dst = sys.argv[1], sys.argv[2]
    out = []
    for rel in FILES:
        path = os.path.join(ref, rel)
        if not os.path.exists(path):
            continue
        src = open(path).read()
        code = src.split("#[cfg(test)]")[0]  # the tests repeat the texts they expect
        for m in re.finditer(r'(bail!|anyhow!|with_context\(\|\|\s*(?:format!\()?|\.context\()\s*\(?\s*"((?:[^"\\]|\\.)*)"', code, re.S):
            text = re.sub(r"\\\n\s*", "", m.group(2))           # a trailing backslash joins the lines
            text = text.replace('\\"', '"').replace("\\t", "\t")
            line = code.count("\n", 0, m.start()) + 1
            if not any(o["text"] == text and o["file"] == rel for o in out):
                out.append({"file": rel, "line": line, "text": text})
    with open(dst, "w") as f:
        json.dump({"source": "stjude-rust-labs/ngs v0.4.0", "messages": out}, f, indent=1)
    for o in out:
        print(f'{o["file"]}:{o["line"]}: {o["text"]!r}')


if __name__ == "__main__":
    main()
