#!/usr/bin/env python3
"""Derive the command-line surface of `ngs qc` from the reference's clap definition (src/qc/command.rs:36-102,
`struct QcArgs`) and write it as a fixture (tests/golden/qc_cli_surface.json): positionals in order; for every option
its long name (clap: the field name with '-' for '_' unless `long = "..."` says otherwise), short letter, value name
and default.  tests/test_cli.py holds this build's `ngs qc` to it.

    python tests/golden/make_cli_surface.py /root/reference tests/golden/qc_cli_surface.json
"""
import json
import os
import re
import sys


def main():
    ref, dst = sys.argv[1], sys.argv[2]
    src = open(os.path.join(ref, "src/qc/command.rs")).read()
    body = re.search(r"pub struct QcArgs \{(.*?)\n\}", src, re.S).group(1)
    positionals, options = [], []
    attr = None
    for line in body.splitlines():
        line = line.strip()
        m = re.match(r"#\[arg\((.*)\)\]$", line)
        if m:
            attr = m.group(1)
            continue
        m = re.match(r"(\w+):\s*(.+?),$", line)
        if not m:
            continue
        field, ty = m.groups()
        a, attr = attr or "", None
        has_long = re.search(r"\blong\b", a) is not None
        if not has_long and "short" not in a:
            value = re.search(r'value_name = "([^"]+)"', a)
            positionals.append({"field": field, "value_name": value.group(1) if value else field.upper(), "type": ty})
            continue
        long = re.search(r'long = "([^"]+)"', a)
        short = re.search(r"short = '(.)'", a)
        value = re.search(r'value_name = "([^"]+)"', a)
        default = re.search(r'default_value = "([^"]+)"', a)
        options.append({"field": field, "long": long.group(1) if long else field.replace("_", "-"),
                        "short": short.group(1) if short else None, "value_name": value.group(1) if value else None,
                        "default": default.group(1) if default else None, "optional": ty.startswith("Option<")})
    with open(dst, "w") as f:
        json.dump({"source": "stjude-rust-labs/ngs v0.4.0 src/qc/command.rs:36-102 (struct QcArgs)",
                   "positionals": positionals, "options": options}, f, indent=1)
    print(dst, len(positionals), "positionals,", len(options), "options")


if __name__ == "__main__":
    main()
