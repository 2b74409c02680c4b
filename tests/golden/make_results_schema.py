#!/usr/bin/env python3
"""Derive the schema of `<prefix>.results.json` from the reference's own struct definitions and write it as a
fixture (tests/golden/results_schema.json).  `Results` (src/qc/results.rs:23-45) is serialized by serde with no
rename / skip attributes anywhere below it, so the JSON document is: the fields of every struct, by name, in
declaration order.  The fixture holds field names, declaration order and Rust type names -- data about the output
format, not source -- and tests/test_results_schema.py holds both the oracle's and the GPU path's documents to it.

    python tests/golden/make_results_schema.py /root/reference tests/golden/results_schema.json
"""
import json
import os
import re
import sys

# module paths as written in the field types -> file that defines the struct
FILES = {
    "results": "src/qc/results.rs",
    "general::metrics": "src/qc/record_based/general/metrics.rs",
    "features": "src/qc/record_based/features/metrics.rs",
    "gc_content::metrics": "src/qc/record_based/gc_content/metrics.rs",
    "template_length": "src/qc/record_based/template_length.rs",
    "quality_scores": "src/qc/record_based/quality_scores.rs",
    "coverage": "src/qc/sequence_based/coverage.rs",
    "edits": "src/qc/sequence_based/edits.rs",
    "histogram": "src/utils/histogram.rs",
}
PRIMITIVES = {"usize", "u64", "u32", "i32", "i64", "f64", "f32", "String", "bool"}


def structs_of(path):
    """{name: [(field, type), ...]} for every struct deriving Serialize in the file; asserts there is no serde attribute."""
    src = open(path).read()
    assert "#[serde(" not in src, f"{path}: a serde attribute changes the document; extend this script"
    out = {}
    for m in re.finditer(r"#\[derive\(([^)]*)\)\]\s*pub struct (\w+)\s*\{(.*?)\n\}", src, re.S):
        if "Serialize" not in m.group(1):
            continue
        fields = []
        for line in m.group(3).splitlines():
            f = re.match(r"\s*(?:pub(?:\([a-z]+\))? )?(\w+):\s*(.+?),\s*$", line)  # serde does not care about visibility
            if f:
                fields.append((f.group(1), f.group(2)))
        out[m.group(2)] = fields
    return out


def main():
    ref, dst = sys.argv[1], sys.argv[2]
    defs = {mod: structs_of(os.path.join(ref, p)) for mod, p in FILES.items()}

    def resolve(ty, mod):
        """schema node of a Rust type written inside module `mod`"""
        ty = ty.strip()
        g = re.match(r"(Option|Vec)<(.+)>$", ty)
        if g:
            return {"kind": g.group(1).lower(), "of": resolve(g.group(2), mod)}
        g = re.match(r"HashMap<(.+?),\s*(.+)>$", ty)
        if g:
            return {"kind": "map", "key": g.group(1).strip(), "of": resolve(g.group(2), mod)}
        if ty in PRIMITIVES:
            return {"kind": ty}
        path, _, name = ty.rpartition("::")
        target = path if path else mod
        if name == "Histogram" and "Histogram" not in defs.get(target, {}):
            target = "histogram"
        assert name in defs[target], f"cannot resolve {ty} (from {mod})"
        return {"kind": "struct", "name": name,
                "fields": [{"name": f, "type": t, "schema": resolve(t, target)} for f, t in defs[target][name]]}

    schema = resolve("Results", "results")
    with open(dst, "w") as f:
        json.dump({"source": "stjude-rust-labs/ngs v0.4.0, struct definitions reachable from qc::results::Results",
                   "files": FILES, "schema": schema}, f, indent=1)
    n = json.dumps(schema).count('"name"')
    print(f"{dst}: {n} names")


if __name__ == "__main__":
    main()
