"""The header `ngs qc` meets in practice: the 195 @SQ lines of the GRCh38 no-alt analysis set
(GCA_000001405.15), for tests, the fuzzer and bench.py's `whole_genome` leg.

The reference insists that every @SQ of the BAM is in the named genome (src/qc/command.rs:258-272), loops over all
of them in pass 2 (:356), and counts 193 of the 195 as the primary assembly (22 autosomes + X, Y + 42 unlocalized +
127 unplaced; chrM and chrEBV are not: src/utils/genome/ncbi/grch38_no_alt.rs:17-285, test :308-311).

Names and groups: ngs_amd/data/GRCh38_no_alt_AnalysisSet.tsv (extracted from the reference's table).  Order: the
order of the public FASTA -- chr1..chr22, chrX, chrY, chrM, the *_random contigs, the chrUn_* contigs, chrEBV.
Lengths: chr1..chr22, X, Y, M and EBV are the assembly's real lengths.  The reference tree holds NO lengths, and
this image has no network: the lengths of the 169 unlocalized / unplaced contigs are deterministic STAND-INS drawn
log-uniformly from the real range (970 bp .. 450 kb; the shortest real contig, chrUn_KI270394v1, is 970 bp), except
the few written out below.  They give the right SHAPE -- 24 sequences of 47-249 Mbp followed by 169 of a few kb --
not the assembly's exact 3 099 922 541 bp.
"""
from __future__ import annotations

import os
import zlib
from typing import List, Tuple

CHROMOSOME_LEN = {
    "chr1": 248_956_422, "chr2": 242_193_529, "chr3": 198_295_559, "chr4": 190_214_555, "chr5": 181_538_259,
    "chr6": 170_805_979, "chr7": 159_345_973, "chr8": 145_138_636, "chr9": 138_394_717, "chr10": 133_797_422,
    "chr11": 135_086_622, "chr12": 133_275_309, "chr13": 114_364_328, "chr14": 107_043_718, "chr15": 101_991_189,
    "chr16": 90_338_345, "chr17": 83_257_441, "chr18": 80_373_285, "chr19": 58_617_616, "chr20": 64_444_167,
    "chr21": 46_709_983, "chr22": 50_818_468, "chrX": 156_040_895, "chrY": 57_227_415, "chrM": 16_569,
    "chrEBV": 171_823,
}
# contigs whose real length is remembered with confidence
KNOWN_CONTIG_LEN = {"chrUn_KI270302v1": 2_274, "chrUn_KI270394v1": 970, "chrUn_KI270442v1": 392_061,
                    "chr1_KI270706v1_random": 175_055, "chr14_GL000225v1_random": 211_173}
PRIMARY_GROUPS = ("autosome", "sex", "alt", "unlocalized", "unplaced")  # src/utils/genome.rs:59-83


def _table() -> List[Tuple[str, str]]:
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "GRCh38_no_alt_AnalysisSet.tsv")
    rows = []
    for line in open(path):
        line = line.rstrip("\n")
        if not line or line.startswith("#"):
            continue
        name, group = line.split("\t")
        rows.append((name, group))
    return rows


def _stand_in_len(name: str) -> int:
    """log-uniform in [970, 450 000], a pure function of the name"""
    h = zlib.crc32(name.encode()) / 2 ** 32
    return int(970 * (450_000 / 970) ** h)


def grch38_no_alt(scale: int = 1, min_chromosome: int = 1) -> Tuple[List[str], List[int], List[int]]:
    """(names, lengths, is_primary) in FASTA order.  `scale` divides the lengths of the 24 chromosomes only (so that a
    CPU oracle with one usize per position fits: scale 64 -> 48 Mbp of chromosomes); contigs, chrM and chrEBV keep
    their size -- they are what makes the header awkward."""
    rows = _table()
    order = {"autosome": 0, "sex": 0, "mitochondrion": 1, "unlocalized": 2, "unplaced": 3, "ebv": 4}
    rows = sorted(rows, key=lambda r: order.get(r[1], 5))  # stable: the table's order within a group
    names, lens, primary = [], [], []
    for name, group in rows:
        if group in ("autosome", "sex"):
            L = max(min_chromosome, CHROMOSOME_LEN[name] // scale)
        elif name in CHROMOSOME_LEN:
            L = CHROMOSOME_LEN[name]
        else:
            L = KNOWN_CONTIG_LEN.get(name) or _stand_in_len(name)
        names.append(name)
        lens.append(L)
        primary.append(1 if group in PRIMARY_GROUPS else 0)
    return names, lens, primary
