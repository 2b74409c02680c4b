#!/usr/bin/env python3
"""Extracts the sequence-name table of a supported reference genome from the reference's
data tables into a plain TSV (name <TAB> group), the only form in which this repository keeps
it.  The names are the public sequence identifiers of the GRCh38 no-alt analysis set
(GCA_000001405.15); the reference lists them in src/utils/genome/ncbi/grch38_no_alt.rs.

    python tools/extract_genome_table.py /root/reference ngs_amd/data

Groups follow the ReferenceGenome trait (src/utils/genome.rs:252-334): autosome, sex,
mitochondrion, alt, ebv, unlocalized, unplaced, decoy, other.  The primary assembly used by the
Coverage facet is autosome+sex+alt+unlocalized+unplaced (src/utils/genome.rs:59-83).
"""
import os
import re
import sys

GROUPS = {"autosomes": "autosome", "sex_chromosomes": "sex", "mitochondrion_chromosome": "mitochondrion",
          "alternative_contig_sequences": "alt", "ebv_chromosome": "ebv", "unlocalized_sequences": "unlocalized",
          "unplaced_sequences": "unplaced", "decoy_sequences": "decoy", "other_sequences": "other"}


def main():
    ref, out = sys.argv[1], sys.argv[2]
    src = open(os.path.join(ref, "src/utils/genome/ncbi/grch38_no_alt.rs")).read()
    genome = re.search(r'fn name\(&self\) -> &\'static str \{\s*"([^"]+)"', src).group(1)
    rows, group = [], None
    for line in src.splitlines():
        m = re.search(r"fn (\w+)\(&self\)", line)
        if m:
            group = GROUPS.get(m.group(1))
        m = re.search(r'sequence!\("([^"]+)"', line)
        if m and group and "#[cfg(test)]" not in line:
            rows.append((m.group(1), group))
        if line.strip().startswith("#[cfg(test)]"):
            break
    path = os.path.join(out, genome + ".tsv")
    with open(path, "w") as f:
        f.write(f"# {genome}: sequence name <TAB> group (tools/extract_genome_table.py)\n")
        for name, g in rows:
            f.write(f"{name}\t{g}\n")
    from collections import Counter
    print(path, Counter(g for _, g in rows))


if __name__ == "__main__":
    main()
