#!/usr/bin/env python3
"""Read the constants the reference fixes at facet construction out of its source and write them as a fixture
(tests/golden/reference_constants.json); tests/test_abi.py holds include/ngsq.h and the library's defaults to them.

    python tests/golden/make_reference_constants.py /root/reference tests/golden/reference_constants.json
"""
import json
import os
import re
import sys


def main():
    ref, dst = sys.argv[1], sys.argv[2]

    def src(p):
        return open(os.path.join(ref, p)).read()

    def one(pattern, text, what):
        m = re.search(pattern, text, re.S)
        assert m, f"cannot find {what}"
        return m.group(1)

    qc, cov, gc, qs = src("src/qc.rs"), src("src/qc/sequence_based/coverage.rs"), src("src/qc/record_based/gc_content.rs"), \
        src("src/qc/record_based/quality_scores.rs")
    edits, hist, general = src("src/qc/sequence_based/edits.rs"), src("src/utils/histogram.rs"), src("src/qc/record_based/general.rs")
    names = {}
    for key, p in (("general", "record_based/general.rs"), ("template_length", "record_based/template_length.rs"),
                   ("gc_content", "record_based/gc_content.rs"), ("quality_score", "record_based/quality_scores.rs"),
                   ("features", "record_based/features.rs"), ("coverage", "sequence_based/coverage.rs"),
                   ("edits", "sequence_based/edits.rs")):
        names[key] = one(r'fn name\(&self\) -> &\'static str \{\s*"([^"]+)"', src("src/qc/" + p), f"name() of {key}")
    out = {
        "source": "stjude-rust-labs/ngs v0.4.0",
        "facet_names": names,
        "template_length_capacity": int(one(r"TemplateLengthFacet::with_capacity\((\d+)\)", qc, "template length capacity (qc.rs)")),
        "coverage_bin_size": int(one(r"NonZeroUsize::new\(([\d_]+)\)", qc, "coverage bin size (qc.rs)").replace("_", "")),
        "coverage_histogram_capacity": int(one(r"COVERAGE_DISTRIBUTION_HISTOGRAM_SIZE: usize = (\d+);", cov, "coverage histogram size")),
        "genome_covered_by": [int(x) for x in re.findall(r"\d+", one(r"COVERAGES_TO_CHECK: \[usize; \d+\] = \[([^\]]+)\]", cov, "genome_covered_by thresholds"))],
        "gc_truncation_length": int(one(r"TRUNCATION_LENGTH: usize = (\d+);", gc, "GC truncation length")),
        "max_quality_score": int(one(r"MAX_SCORE: usize = (\d+);", qs, "max quality score")),
        "default_histogram_capacity": int(one(r"fn default\(\) -> Self \{\s*Self::zero_based_with_capacity\((\d+)\)", hist, "Histogram::default")),
        "vaf_histogram_capacity": int(one(r"vaf_histogram: Histogram::zero_based_with_capacity\((\d+)\)", edits, "VAF histogram capacity")),
        "high_quality_mapq": int(one(r"if mapq >= (\d+)", general, "high-quality MAPQ threshold")),
    }
    with open(dst, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
