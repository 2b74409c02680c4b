"""Records, reference bases, gene model, FASTA and GFF files over the 195-sequence header of the GRCh38 no-alt
analysis set (ngs_amd/genome_shape.py) -- the header `ngs qc` meets in practice (src/qc/command.rs:258-272,356;
src/utils/genome/ncbi/grch38_no_alt.rs:17-285).  Chromosomes scaled down so that the oracle's one-usize-per-position
arrays fit; contigs, chrM and chrEBV at their size."""
from __future__ import annotations

import gzip
from typing import List, Sequence, Tuple

import numpy as np

from ngs_amd import host
from ngs_amd.genome_shape import grch38_no_alt
from tests.util import coordinate_sorted, make_edit_friendly, random_batch, take_records

LETTERS = "=ACMGRSVTWYHKDBN"
GENOME = "GRCh38_no_alt_AnalysisSet"
FEATURE_TYPES = ["five_prime_UTR", "three_prime_UTR", "CDS", "exon", "gene"]


def header(scale: int = 64):
    return grch38_no_alt(scale)


def empty_sequences(names: Sequence[str]) -> List[int]:
    """Sequences no record lands on: two chromosomes, every fifth contig -- empty ones between covered ones."""
    out = [names.index("chr5"), names.index("chr19")]
    out += [r for r in range(25, len(names)) if r % 5 == 0]
    return out


def genome_records(seed: int, n: int, names: Sequence[str], lens: Sequence[int], bases, clean: bool) -> host.HostBatch:
    """n records spread over the header: a third on the 24 chromosomes, the rest on chrM, chrEBV and the contigs that are
    not left empty.  `clean`: every record is one the reference would process without aborting (the CLI writes a document
    only then) -- the Edits walk succeeds, mapped records have a sequence and a position; reads that straddle the END of
    their sequence are marked duplicate (Coverage has no flag filter and counts them position by position, coverage.rs:
    162-176; Edits skips duplicates, edits.rs:227-229).  Otherwise a quarter of the records are random_batch's raw ones:
    every flag bit and CIGAR operation, positions at and beyond the end, missing ids -- both sides count the same errors."""
    rng = np.random.default_rng(seed)
    nr = len(names)
    dead = set(empty_sequences(names))
    live = [r for r in range(nr) if r not in dead]
    chrom = [r for r in live if r < 24]
    # slots: which sequence a drawn index means (chromosomes several times over)
    slots = np.array(chrom * (len(live) // (2 * len(chrom)) + 1) + live, dtype=np.int32)
    hb = random_batch(rng, n, [lens[r] for r in slots], max_len=260, min_len=20 if clean else 0, weird=not clean)
    c = hb.cols
    for col in ("ref_id", "mate_ref_id"):
        a = c[col]
        c[col] = np.where(a >= 0, slots[np.clip(a, 0, len(slots) - 1)], a).astype(np.int32)
    friendly = make_edit_friendly(hb, rng, bases, lens)
    if clean:
        hb = friendly
        c = hb.cols
        c["flag"] &= np.uint16(0xFFFF ^ 0x1)                 # unpaired: no mate reference needed (general.rs:81-83)
        straddle = rng.random(n) < 0.04
        L = np.array(lens, dtype=np.int64)[c["ref_id"]]
        c["pos"] = np.where(straddle, np.maximum(L - rng.integers(1, 100, n), 0), c["pos"]).astype(np.int32)
        c["flag"] = np.where(straddle, c["flag"] | 0x400, c["flag"]).astype(np.uint16)
    else:
        pick = rng.random(n) < 0.75
        order = np.arange(n)
        a, b = take_records(friendly, order[pick]), take_records(hb, order[~pick])
        hb = concat(a, b)
    return hb


def concat(a: host.HostBatch, b: host.HostBatch) -> host.HostBatch:
    cols = {}
    for k in host.FIXED_COLUMNS:
        cols[k] = np.concatenate([a.cols[k], b.cols[k]])
    for data, off in (("seq", "seq_off"), ("qual", "qual_off"), ("cigar", "cigar_off")):
        cols[data] = np.concatenate([a.cols[data], b.cols[data]])
        cols[off] = np.concatenate([a.cols[off], b.cols[off][1:] + a.cols[off][-1]]).astype(np.uint64)
    return host.HostBatch(a.n + b.n, cols, 0, 0, 0, 0)


def sorted_records(hb: host.HostBatch) -> host.HostBatch:
    return coordinate_sorted(hb)


def reference_bases(seed: int, lens: Sequence[int], soft_masked: bool = False):
    """4-bit codes per sequence (A C G T with 1 % N), and -- for the FASTA -- which positions are lower case."""
    rng = np.random.default_rng(seed)
    bases = [rng.choice(np.array([1, 2, 4, 8, 15], dtype=np.uint8), int(L), p=[.25, .25, .25, .24, .01]) for L in lens]
    lower = None
    if soft_masked:   # repeats: runs of a few hundred bases, about half of every sequence (the analysis set is soft-masked)
        lower = []
        for L in lens:
            m = np.zeros(int(L), dtype=bool)
            p = 0
            while p < L:
                run = int(rng.integers(50, 700))
                if rng.random() < 0.5:
                    m[p:p + run] = True
                p += run
            lower.append(m)
    return bases, lower


def write_fasta(path: str, names, bases, lower=None, width: int = 60, order=None, extra=()):
    """`order`: the FASTA's own sequence order (any; edits.rs:185-205 searches by name).  `extra`: (name, text) records
    the BAM knows nothing about."""
    lut = np.frombuffer(LETTERS.encode(), dtype=np.uint8)
    with open(path, "wb") as f:
        for name, text in extra:
            f.write(f">{name}\n{text}\n".encode())
        for r in (order if order is not None else range(len(names))):
            s = lut[bases[r]]
            if lower is not None:
                s = np.where(lower[r], s | 0x20, s).astype(np.uint8)   # ASCII lower case ('=' is never masked: codes are ACGTN)
            f.write(f">{names[r]} stand-in sequence {r}\n".encode())
            n = len(s)
            full = n // width * width
            if full:
                lines = np.empty((full // width, width + 1), dtype=np.uint8)
                lines[:, :width] = s[:full].reshape(-1, width)
                lines[:, width] = 10
                f.write(lines.tobytes())
            if n > full:
                f.write(s[full:].tobytes() + b"\n")


def gene_model(seed: int, names, lens, primary, n: int) -> Tuple[list, list]:
    """n GFF rows over the header (every feature type the facet knows, two it does not; both strands; on chrM / chrEBV
    too, which are not primary and get no interval store: features.rs:300-304) and the model the facet keeps of them."""
    rng = np.random.default_rng(seed)
    types = FEATURE_TYPES + ["transcript", "start_codon"]
    rows, model = ["##gff-version 3", "#!genome-build stand-in"], []
    for _ in range(n):
        seq = int(rng.integers(0, len(names))) if rng.random() < 0.5 else int(rng.integers(0, 24))
        s = int(rng.integers(1, lens[seq] + 1))
        e = min(lens[seq], s + (0 if rng.random() < 0.05 else int(rng.integers(0, 3000))))
        t = types[int(rng.integers(0, len(types)))]
        rows.append(f"{names[seq]}\tHAVANA\t{t}\t{s}\t{e}\t.\t{'+-'[int(rng.integers(0, 2))]}\t.\tID=f{len(rows)};gene_type=x")
        if t in FEATURE_TYPES and primary[seq]:
            model.append((seq, FEATURE_TYPES.index(t), s, e))
    return rows, model


def write_gff(path: str, rows):
    data = ("\n".join(rows) + "\n").encode()
    if path.endswith(".gz"):
        with gzip.open(path, "wb", compresslevel=1) as f:
            f.write(data)
    else:
        with open(path, "wb") as f:
            f.write(data)


def model_arrays(model):
    return [np.array(c, dtype=np.uint32) for c in zip(*model)]
