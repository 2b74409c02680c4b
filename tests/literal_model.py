"""A SECOND, independent restatement of the reference's `ngs qc` facets -- test infrastructure, like oracle/.

oracle/oracle.c is the checker every GPU result is judged by, and (oracle/oracle.h) the reference holds
no test vector for any facet's `process`: the oracle is a reading of the Rust source.  This module is another reading
of the same source, written from the .rs files alone (not from oracle.c), in the reference's own shape -- one record
at a time, a dict where the reference has a HashMap, Python integers where it has usize, a `Histogram` class with the
reference's methods -- so that tests/test_literal_model.py can hold the two readings against each other on random
records.  A slip in either shows as a difference; a misreading both share does not (the pin stays what oracle.h says).

Where the reference panics or bails (an `unwrap()` on None, a `?` on an Err), `Abort` is raised with the place: the
run would have ended there.  The decode the reference leaves to noodles is restated with the same assumptions
oracle.h lists as [N1]..[N9]; they are marked below.  All paths are relative to /root/reference.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import numpy as np

KINDS = "MIDNSHP=X"                      # noodles sam::record::cigar::op::Kind, in BAM's code order [N8]
BASES = "=ACMGRSVTWYHKDBN"               # noodles sam::record::sequence::Base, BAM's 4-bit codes [N7]
MASK64 = (1 << 64) - 1


class Abort(Exception):
    """The reference would have stopped here (panic or Err)."""

    def __init__(self, where: str, what: str):
        super().__init__(f"{where}: {what}")
        self.where, self.what = where, what


def fdiv(a: float, b: float) -> float:
    """IEEE 754 division (Rust's f64 `/`): x/0 is an infinity, 0/0 a NaN."""
    if b == 0.0:
        if a == 0.0 or a != a:
            return math.nan
        return math.copysign(math.inf, a) * math.copysign(1.0, b)
    return a / b


def jnum(x: float):
    """serde_json: a NaN or an infinity is written as null."""
    return None if (x != x or x in (math.inf, -math.inf)) else x


def jf32(x: np.float32):
    """An f32 as serde_json writes it (ryu: the shortest text that reads back as the same f32), read as JSON."""
    if not np.isfinite(x):
        return None
    return float(np.format_float_positional(np.float32(x), unique=True, trim="0"))


class Histogram:
    """src/utils/histogram.rs:152-392"""

    def __init__(self, capacity: int = 512):          # Default: zero_based_with_capacity(512), :394-398
        self.values = [0] * (capacity + 1)            # :172-178
        self.range_start, self.range_stop = 0, capacity

    def increment(self, b: int) -> bool:              # :197-199; False <-> Err(BinOutOfBoundsError)
        return self.increment_by(b, 1)

    def increment_by(self, b: int, v: int) -> bool:   # :217-224
        if b < self.range_start or b > self.range_stop:
            return False
        self.values[b] += v
        return True

    def get(self, b: int) -> int:                     # :240-247 (panics beyond the vector)
        if b < 0 or b >= len(self.values):
            raise Abort("histogram.rs:241", f"Could not lookup value for template length histogram bin: {b}.")
        return self.values[b]

    def mean(self) -> float:                          # :258-269
        s, d = 0.0, 0.0
        for i in range(self.range_start, self.range_stop + 1):
            v = self.get(i)
            d += float(v)
            s += float(v * i)
        return fdiv(s, d)

    def percentile(self, p: float) -> Optional[float]:   # :272-337
        if not (0.0 <= p <= 1.0):
            raise Abort("histogram.rs:275", "Provided percentile was not within a valid range.")
        n = 0
        for i in range(self.range_start, self.range_stop + 1):
            n += self.get(i)
        if n == 0:
            return None
        needed = p * float(n)
        collected, index = 0.0, self.range_start
        while True:
            if index > self.range_stop:
                raise Abort("histogram.rs:304", "Unknown error!")
            collected += float(self.get(index))
            if collected > needed:
                return float(index)
            if collected == needed:
                lowest = index
                index += 1
                while self.get(index) == 0:           # (runs off the vector -> get() panics)
                    index += 1
                return float(lowest) + float(index - lowest) / 2.0
            index += 1

    def median(self) -> Optional[float]:              # :344-346
        return self.percentile(0.5)

    def sum(self) -> int:                             # :367-369
        return sum(self.values)

    def count_from_top_until(self, b: int) -> int:    # :384-391
        return sum(self.get(i) for i in range(b, self.range_stop + 1))

    def json(self) -> dict:                           # serde field order :152-159
        return {"values": list(self.values), "range_start": self.range_start, "range_stop": self.range_stop}


class Record:
    """What the facets ask a noodles `Record` for."""

    def __init__(self, index, flag, mapq, ref_id, pos, mate_ref_id, tlen, cigar, seq_codes, qual):
        self.index = index                    # the record's index in the file (the pinned GC offset is drawn from it)
        self.flag, self.mapq, self.ref_id, self.pos, self.mate_ref_id, self.tlen = flag, mapq, ref_id, pos, mate_ref_id, tlen
        self.cigar = cigar                    # [(kind letter, length)]
        self.seq = seq_codes                  # [4-bit code]
        self.qual = qual                      # [score]; [] when BAM's QUAL is 0xFF-filled [N6]

    # flags (sam::record::Flags)
    def f(self, bit):
        return bool(self.flag & bit)

    def reference_sequence_id(self):          # [N1]
        return None if self.ref_id < 0 else self.ref_id

    def mate_reference_sequence_id(self):     # [N1]
        return None if self.mate_ref_id < 0 else self.mate_ref_id

    def mapping_quality(self):                # [N2]
        return None if self.mapq == 255 else self.mapq

    def alignment_start(self):                # [N3]
        return None if self.pos < 0 else self.pos + 1

    def alignment_span(self):                 # sum of the lengths of M D N = X
        return sum(n for k, n in self.cigar if k in "MDN=X")

    def alignment_end(self):                  # [N4]
        s = self.alignment_start()
        if s is None:
            return None
        e = s + self.alignment_span() - 1
        return None if e == 0 else e


def records_of(hb) -> List[Record]:
    """The records of a tests/util HostBatch in the offsets layout."""
    c = hb.cols
    assert c.get("seq_off") is not None and c.get("qual_off") is not None and c.get("cigar_off") is not None
    rid = c.get("record_id")
    out = []
    for i in range(hb.n):
        ops = []
        for w in c["cigar"][int(c["cigar_off"][i]):int(c["cigar_off"][i + 1])]:
            code = int(w) & 15
            if code > 8:
                raise Abort("noodles-bam", "invalid CIGAR op kind")          # [N8]
            ops.append((KINDS[code], int(w) >> 4))
        l = int(c["l_seq"][i])
        packed = c["seq"][int(c["seq_off"][i]):int(c["seq_off"][i]) + (l + 1) // 2]
        codes = []
        for b in packed:
            codes += [int(b) >> 4, int(b) & 15]
        codes = codes[:l]
        q = [int(x) for x in c["qual"][int(c["qual_off"][i]):int(c["qual_off"][i + 1])]]
        # ([N6]: BAM's 0xFF-filled QUAL is an EMPTY score list -- in the offsets layout of include/ngsq.h that is zero bytes, the
        # readers' job; a 0xFF that does reach a batch this way is a score of 255)
        if any(x > 93 for x in q):
            raise Abort("noodles-bam", "invalid quality score")               # [N6]
        out.append(Record(int(rid[i]) if rid is not None else hb.first_record_index + i, int(c["flag"][i]), int(c["mapq"][i]),
                          int(c["ref_id"][i]), int(c["pos"][i]), int(c["mate_ref_id"][i]), int(c["tlen"][i]), ops, codes, q))
    return out


def mix64(z: int) -> int:                     # include/ngsq_shared.h ngsq_mix64 (the build's pinned stand-in for ThreadRng)
    z = (z + 0x9E3779B97F4A7C15) & MASK64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
    return z ^ (z >> 31)


def gc_offset(gc_seed: int, record_index: int, l_seq: int) -> int:   # include/ngsq_shared.h ngsq_gc_offset_fn
    if l_seq <= 100:
        return 0
    h32 = mix64((gc_seed ^ record_index) & MASK64) >> 32
    return (h32 * (l_seq - 100)) >> 32


def fasta_base(byte: int) -> Optional[str]:   # Base::try_from on a FASTA byte [N9]: case folds; None <-> TryFromCharError
    ch = chr(byte).upper()
    return ch if ch in BASES else None


# ---------------------------------------------------------------------------------------------------------------------
class General:                                # src/qc/record_based/general.rs
    def __init__(self):
        self.r = dict(total=0, unmapped=0, duplicate=0, primary=0, secondary=0, supplementary=0, primary_mapped=0,
                      primary_duplicate=0, paired=0, read_1=0, read_2=0, proper_pair=0, singleton=0, mate_mapped=0,
                      mismatch=0, mismatch_hq=0)
        self.one: Dict[str, int] = {}
        self.two: Dict[str, int] = {}
        self.summary = None

    def process(self, rec: Record):           # :31-124
        r = self.r
        r["total"] += 1
        if rec.f(0x4):
            r["unmapped"] += 1
        if rec.f(0x400):
            r["duplicate"] += 1
        if rec.f(0x100):
            r["secondary"] += 1
        elif rec.f(0x800):
            r["supplementary"] += 1
        else:
            r["primary"] += 1
            if not rec.f(0x4):
                r["primary_mapped"] += 1
            if rec.f(0x400):
                r["primary_duplicate"] += 1
            if rec.f(0x1):
                r["paired"] += 1
                if rec.f(0x40):
                    r["read_1"] += 1
                if rec.f(0x80):
                    r["read_2"] += 1
                if not rec.f(0x4):
                    if rec.f(0x2):
                        r["proper_pair"] += 1
                    if rec.f(0x8):
                        r["singleton"] += 1
                    else:
                        r["mate_mapped"] += 1
                        a, b = rec.reference_sequence_id(), rec.mate_reference_sequence_id()
                        if a is None or b is None:
                            raise Abort("general.rs:81-83", "unwrap() on a missing reference sequence id")
                        if a != b:
                            r["mismatch"] += 1
                            mq = rec.mapping_quality()
                            if (255 if mq is None else mq) >= 5:
                                r["mismatch_hq"] += 1
        ops = self.one if rec.f(0x40) else self.two          # :103-121
        for k, _n in rec.cigar:
            ops[k] = ops.get(k, 0) + 1

    def summarize(self):                      # :126-153
        t = float(self.r["total"])
        self.summary = {
            "duplication_pct": fdiv(float(self.r["duplicate"]), t) * 100.0,
            "mapped_pct": (1.0 - fdiv(float(self.r["unmapped"]), t)) * 100.0,
            "mate_reference_sequence_id_mismatch_pct": fdiv(float(self.r["mismatch"]), t) * 100.0,
            "mate_reference_sequence_id_mismatch_hq_pct": fdiv(float(self.r["mismatch_hq"]), t) * 100.0,
        }

    def json(self):                           # general/metrics.rs
        r = self.r
        return {
            "records": {"total": r["total"], "unmapped": r["unmapped"], "duplicate": r["duplicate"],
                        "designation": {"primary": r["primary"], "secondary": r["secondary"], "supplementary": r["supplementary"]},
                        "primary_mapped": r["primary_mapped"], "primary_duplicate": r["primary_duplicate"], "paired": r["paired"],
                        "read_1": r["read_1"], "read_2": r["read_2"], "proper_pair": r["proper_pair"], "singleton": r["singleton"],
                        "mate_mapped": r["mate_mapped"], "mate_reference_sequence_id_mismatch": r["mismatch"],
                        "mate_reference_sequence_id_mismatch_hq": r["mismatch_hq"]},
            "cigar": {"read_one_cigar_ops": dict(self.one), "read_two_cigar_ops": dict(self.two)},
            "summary": None if self.summary is None else {k: jnum(v) for k, v in self.summary.items()},
        }


class TemplateLength:                         # src/qc/record_based/template_length.rs
    def __init__(self, capacity: int):
        self.h, self.processed, self.ignored, self.summary = Histogram(capacity), 0, 0, None

    def process(self, rec: Record):           # :79-87: `template_length() as usize` (an i32 sign-extended to 64 bits)
        if self.h.increment(rec.tlen & MASK64 if rec.tlen >= 0 else (rec.tlen + (1 << 64))):
            self.processed += 1
        else:
            self.ignored += 1

    def summarize(self):                      # :89-100
        d = float(self.processed) + float(self.ignored)
        self.summary = {"template_length_unknown_pct": fdiv(float(self.h.get(0)), d) * 100.0,
                        "template_length_out_of_range_pct": fdiv(float(self.ignored), d) * 100.0}

    def json(self):
        return {"histogram": self.h.json(), "records": {"processed": self.processed, "ignored": self.ignored},
                "summary": None if self.summary is None else {k: jnum(v) for k, v in self.summary.items()}}


class GcContent:                              # src/qc/record_based/gc_content.rs
    TRUNCATION_LENGTH = 100                   # :18

    def __init__(self, gc_seed: int):
        self.h = Histogram(100)
        self.gc = self.at = self.other = 0
        self.processed = self.ignored_flags = self.ignored_too_short = 0
        self.gc_seed, self.summary = gc_seed, None

    def process(self, rec: Record):           # :38-100
        if rec.f(0x400) or rec.f(0x100):
            self.ignored_flags += 1
            return
        n = len(rec.seq)
        if n < self.TRUNCATION_LENGTH:
            self.ignored_too_short += 1
            return
        # :69-74 draws the offset from ThreadRng; the build pins it (include/ngsq_shared.h), same support
        offset = gc_offset(self.gc_seed, rec.index, n) if self.TRUNCATION_LENGTH < n else 0
        gc_this_read = 0
        for i in range(self.TRUNCATION_LENGTH):
            b = BASES[rec.seq[offset + i]]
            if b in "CG":
                gc_this_read += 1
                self.gc += 1
            elif b in "AT":
                self.at += 1
            else:
                self.other += 1
        x = (float(gc_this_read) / float(self.TRUNCATION_LENGTH)) * 100.0
        pct = int(math.floor(x + 0.5))        # f64::round on a non-negative value
        if not self.h.increment(pct):
            raise Abort("gc_content.rs:93-96", "unwrap() on BinOutOfBoundsError")
        self.processed += 1

    def summarize(self):                      # :102-122
        nb = float(self.gc + self.at + self.other)
        nr = float(self.ignored_flags + self.ignored_too_short + self.processed)
        self.summary = {"gc_content_pct": fdiv(float(self.gc), nb) * 100.0,
                        "ignored_flags_pct": fdiv(float(self.ignored_flags), nr) * 100.0,
                        "ignored_too_short_pct": fdiv(float(self.ignored_too_short), nr) * 100.0}

    def json(self):                           # gc_content/metrics.rs
        return {"histogram": self.h.json(),
                "nucleobases": {"total_gc_count": self.gc, "total_at_count": self.at, "total_other_count": self.other},
                "records": {"processed": self.processed, "ignored_flags": self.ignored_flags, "ignored_too_short": self.ignored_too_short},
                "summary": None if self.summary is None else {k: jnum(v) for k, v in self.summary.items()}}


class QualityScore:                           # src/qc/record_based/quality_scores.rs
    MAX_SCORE = 93                            # :26

    def __init__(self):
        self.scores: Dict[int, Histogram] = {}

    def process(self, rec: Record):           # :37-49
        for i, val in enumerate(rec.qual):
            h = self.scores.get(i + 1)
            if h is None:
                h = self.scores[i + 1] = Histogram(self.MAX_SCORE)
            if not h.increment(val):
                raise Abort("quality_scores.rs:45", "unwrap() on BinOutOfBoundsError")

    def summarize(self):
        pass

    def json(self):
        return {"scores": {str(k): h.json() for k, h in self.scores.items()}}


class Features:                               # src/qc/record_based/features.rs
    def __init__(self, ref_names, primary, intervals, role_names):
        """intervals: (ref_id, type name, start, stop) as GenomicFeaturesFacet::try_from makes them of the GFF's records
        (:288-312: start = record.start(), stop = record.end()); role_names = the five FeatureNames, in declaration order."""
        self.names = list(ref_names)
        self.five, self.three, self.cds, self.exon, self.gene = role_names
        self.primary_names = [n for n, p in zip(ref_names, primary) if p]
        self.utr: Dict[str, list] = {}
        self.genic: Dict[str, list] = {}
        for parent in self.primary_names:     # :295-341: a store per primary sequence, empty or not
            u, g = [], []
            for rid, ty, start, stop in intervals:
                if self.names[rid] != parent:
                    continue
                if ty in (self.five, self.three, self.cds):
                    u.append((start, stop, ty))
                elif ty in (self.exon, self.gene):
                    g.append((start, stop, ty))
            self.utr[parent], self.genic[parent] = sorted(u), sorted(g)   # (rust_lapper sorts its intervals)
        self.m = dict(five=0, three=0, cds=0, intergenic=0, exonic=0, intronic=0, processed=0, ignored_flags=0, ignored_nonprimary=0)
        self.summary = None

    @staticmethod
    def find(store, start, stop):             # rust_lapper Lapper::find: half-open overlap
        return [iv for iv in store if iv[0] < stop and iv[1] > start]

    def process(self, rec: Record):           # :115-242
        m = self.m
        if rec.f(0x4):
            m["ignored_flags"] += 1
            return
        rid = rec.reference_sequence_id()
        if rid is None:
            raise Abort("features.rs:133-141", "Could not parse reference sequence id for read")
        if rid >= len(self.names):
            raise Abort("features.rs:144-157", "Could not map reference sequence id to header for read")
        seq_name = self.names[rid]
        if seq_name not in self.primary_names:
            m["ignored_nonprimary"] += 1
            return
        start = rec.alignment_start()
        if start is None:
            raise Abort("features.rs:171-174", "Could not parse record's start position.")
        end = start + rec.alignment_span()
        five = three = cds = False
        if seq_name in self.utr:
            for _s, _e, name in self.find(self.utr[seq_name], start, end + 1):
                if not five and name == self.five:
                    five = True
                    m["five"] += 1
                elif not three and name == self.three:
                    three = True
                    m["three"] += 1
                elif not cds and name == self.cds:
                    cds = True
                    m["cds"] += 1
        if seq_name in self.genic:
            has_gene = has_exon = False
            for _s, _e, name in self.find(self.genic[seq_name], start, end + 1):
                if name == self.gene:
                    has_gene = True
                elif name == self.exon:
                    has_exon = True
                if has_gene and has_exon:
                    break
            if has_gene:
                if has_exon:
                    m["exonic"] += 1
                else:
                    m["intronic"] += 1
            else:
                m["intergenic"] += 1
        m["processed"] += 1

    def summarize(self):                      # :244-262
        m = self.m
        d = float(m["ignored_flags"] + m["ignored_nonprimary"] + m["processed"])
        self.summary = {"ignored_flags_pct": fdiv(float(m["ignored_flags"]), d) * 100.0,
                        "ignored_nonprimary_chromosome_pct": fdiv(float(m["ignored_nonprimary"]), d) * 100.0}

    def json(self):                           # features/metrics.rs
        m = self.m
        return {"exonic_translation_regions": {"utr_five_prime_count": m["five"], "utr_three_prime_count": m["three"], "coding_sequence_count": m["cds"]},
                "gene_regions": {"intergenic_count": m["intergenic"], "exonic_count": m["exonic"], "intronic_count": m["intronic"]},
                "records": {"processed": m["processed"], "ignored_flags": m["ignored_flags"], "ignored_nonprimary_chromosome": m["ignored_nonprimary"]},
                "summary": None if self.summary is None else {k: jnum(v) for k, v in self.summary.items()}}


class Coverage:                               # src/qc/sequence_based/coverage.rs
    HISTOGRAM_SIZE = 2048                     # :76

    def __init__(self, ref_names, primary, bin_size: int):
        self.primary_names = [n for n, p in zip(ref_names, primary) if p]
        self.per_position: Dict[str, Histogram] = {}
        self.mean: Dict[str, float] = {}
        self.per_bin: Dict[str, List[float]] = {}
        self.median: Dict[str, float] = {}
        self.median_over_mean: Dict[str, float] = {}
        self.nonsensical = 0
        self.too_large: Dict[str, int] = {}
        self.distribution = Histogram(self.HISTOGRAM_SIZE)
        self.covered_by: Dict[str, np.float32] = {}
        self.bin_size = bin_size

    def supports(self, name):                 # :133-138
        return name in self.primary_names

    def setup(self, name, L):
        pass

    def process(self, name, L, rec: Record):  # :148-180
        h = self.per_position.get(name)
        if h is None:
            h = self.per_position[name] = Histogram(L)
        start, end = rec.alignment_start(), rec.alignment_end()
        if start is None or end is None:
            raise Abort("coverage.rs:159-160", "unwrap() on a missing alignment start / end")
        for i in range(start, end + 1):
            if not h.increment(i):
                self.nonsensical += 1

    def teardown(self, name, L):              # :182-262
        positions = self.per_position.get(name)
        if positions is None:
            return
        coverages, ignored, total = Histogram(self.HISTOGRAM_SIZE), 0, 0
        bins = self.per_bin.setdefault(name, [])
        for i in range(positions.range_start, positions.range_stop + 1):
            c = positions.get(i)
            if not coverages.increment(c):
                ignored += 1
            total += c
            if i % self.bin_size == 0:
                bins.append(float(total) / float(self.bin_size))
                total = 0
        modulo = positions.range_stop % self.bin_size
        if modulo != 0:
            bins.append(float(total) / float(modulo))
        mean = coverages.mean()
        median = coverages.median()
        if median is None:
            raise Abort("coverage.rs:233", "unwrap() on the median of an empty histogram")
        mom = fdiv(median, mean)
        del self.per_position[name]
        for i in range(coverages.range_start, coverages.range_stop + 1):
            assert self.distribution.increment_by(i, coverages.get(i))
        self.mean[name], self.median[name], self.median_over_mean[name], self.too_large[name] = mean, median, mom, ignored

    def aggregate(self):                      # :264-287
        total = self.distribution.sum()
        for v in self.too_large.values():
            total += v
        for c in (10, 20, 30, 40, 50, 60):
            n = self.distribution.count_from_top_until(c)
            with np.errstate(all="ignore"):
                self.covered_by[f"{c}x"] = np.float32(np.float32(n) / np.float32(total)) * np.float32(100.0)

    def json(self):
        return {"mean_coverage": {k: jnum(v) for k, v in self.mean.items()},
                "mean_coverage_per_bin": {k: [jnum(x) for x in v] for k, v in self.per_bin.items()},
                "median_coverage": {k: jnum(v) for k, v in self.median.items()},
                "median_over_mean_coverage": {k: jnum(v) for k, v in self.median_over_mean.items()},
                "ignored": {"nonsensical_records": self.nonsensical, "pileup_too_large_positions": dict(self.too_large)},
                "coverage_distribution": self.distribution.json(),
                "genome_covered_by": {k: jf32(v) for k, v in self.covered_by.items()}}


def consumes_reference(k):                    # src/utils/cigar.rs:6-11
    return k in "MDN=X"


def consumes_sequence(k):                     # src/utils/cigar.rs:14-23
    return k in "MIS=X"


class Edits:                                  # src/qc/sequence_based/edits.rs
    def __init__(self, fasta: Dict[str, bytes]):
        """fasta: name -> the sequence's bytes as noodles-fasta holds them (line terminators gone, case as in the file)"""
        self.fasta = fasta
        self.one, self.two, self.vaf = Histogram(), Histogram(), Histogram(100)   # :60-69
        self.refs, self.alts = Histogram(), Histogram()
        self.current: Optional[bytes] = None
        self.summary = None
        self.vaf_rows: List[tuple] = []       # what --vaf-file would hold: (sequence, position, vaf as f32)

    def supports(self, name):                 # :173-175
        return True

    def setup(self, name, L):                 # :177-215
        for n, s in self.fasta.items():
            if n == name:
                self.current = s
                break
        if self.current is None:
            raise Abort("edits.rs:203-205", f"sequence {name} not found in reference FASTA.")
        self.refs, self.alts = Histogram(L), Histogram(L)

    def process(self, name, L, rec: Record):  # :217-303
        if rec.f(0x4) or rec.f(0x400):
            return
        start = rec.alignment_start()
        if start is None:
            raise Abort("edits.rs:242", "unwrap() on a missing alignment start")
        end = start + rec.alignment_span()
        if self.current is None:
            raise Abort("edits.rs:245-251", "could not lookup reference sequence for read")
        # Sequence::get(start..end): the bases of the 1-based positions [start, end); None beyond the sequence
        lo, hi = start - 1, end - 1
        if hi > len(self.current) or lo > hi:
            raise Abort("edits.rs:257-260", "unwrap() on a slice beyond the reference sequence")
        ref = []
        for byte in self.current[lo:hi]:
            b = fasta_base(byte)
            if b is None:
                raise Abort("edits.rs:263", "TryFromCharError")
            ref.append(b)
        seq = [BASES[c] for c in rec.seq]
        flat = [k for k, n in rec.cigar for _ in range(n)]      # utils/alignment.rs:9-22
        edits = 0
        rp = qp = 0                                              # alignment.rs:48-107
        for kind in flat:
            cr, cs = consumes_reference(kind), consumes_sequence(kind)
            rb = sb = None
            if cr:
                if rp >= len(ref):
                    raise Abort("alignment.rs:61-64", "malformed record: ... consume a reference base, but no such base was found")
                rb = ref[rp]
            if cs:
                if qp >= len(seq):
                    raise Abort("alignment.rs:76-79", "malformed record: ... consume a record base, but no such base was found")
                sb = seq[qp]
            if kind == "M":                                      # edits.rs:276-291
                position = start + rp
                if rb != sb:
                    edits += 1
                    if not self.alts.increment(position):
                        raise Abort("edits.rs:283-285", "unwrap() on BinOutOfBoundsError")
                else:
                    if not self.refs.increment(position):
                        raise Abort("edits.rs:287-289", "unwrap() on BinOutOfBoundsError")
            if cr:
                rp += 1
            if cs:
                qp += 1
        if len(ref) != rp:
            raise Abort("alignment.rs:100-101", "reference sequence was not fully consumed")
        if len(seq) != qp:
            raise Abort("alignment.rs:102-103", "record sequence was not fully consumed")
        if not (self.one if rec.f(0x40) else self.two).increment(edits):
            raise Abort("edits.rs:296-300", "unwrap() on BinOutOfBoundsError")

    def teardown(self, name, L):              # :305-344
        self.current = None
        for i in range(self.refs.range_start, self.refs.range_stop + 1):
            r, a = self.refs.get(i), self.alts.get(i)
            total = r + a
            if total == 0:
                continue
            vaf = np.float32(a) / np.float32(total)
            b = int(np.float32(vaf * np.float32(100.0)))        # `as usize`: truncation
            if not self.vaf.increment(b):
                raise Abort("edits.rs:333-336", "unwrap() on BinOutOfBoundsError")
            self.vaf_rows.append((name, i, vaf))

    def aggregate(self):                      # :346-353
        self.summary = {"mean_edits_read_one": self.one.mean(), "mean_edits_read_two": self.two.mean()}

    def json(self):
        return {"read_one_edits": self.one.json(), "read_two_edits": self.two.json(), "vaf_histogram": self.vaf.json(),
                "summary": None if self.summary is None else {k: jnum(v) for k, v in self.summary.items()}}


# ---------------------------------------------------------------------------------------------------------------------
def query(records: Sequence[Record], ref_id: int, L: int):
    """bam::Reader::query over the whole sequence [N5]: the records of that sequence whose [start, end] meets [1, L]."""
    for rec in records:
        if rec.ref_id != ref_id:
            continue
        s, e = rec.alignment_start(), rec.alignment_end()
        if s is None or e is None:
            continue
        if s <= L and e >= 1:
            yield rec


def run(records: Sequence[Record], ref_names: Sequence[str], ref_len: Sequence[int], primary: Sequence[int], *, general=True,
        template_length=True, gc_content=True, quality_scores=True, coverage=True, bin_size=50_000, tlen_cap=1024, gc_seed=0,
        fasta: Optional[Dict[str, bytes]] = None, intervals=None, role_names=None, num_records: Optional[int] = None) -> dict:
    """src/qc/command.rs:226-421: pass 1 over the records, summarize, pass 2 sequence by sequence, aggregate.  num_records = `-n`
    (utils/display.rs:58-63: the counter is looked at AFTER a record is processed, so `-n 0` still takes one; pass 2 keeps ONE
    counter over all sequences and only leaves the sequence's loop, so every later sequence still takes one record).
    Returns the Results document (src/qc/results.rs:23-45) as parsed JSON."""
    rec_facets = []
    g = t = c = q = f = None
    if general:
        g = General()
        rec_facets.append(g)
    if intervals is not None:
        f = Features(ref_names, primary, intervals, role_names)
        rec_facets.append(f)
    if gc_content:
        c = GcContent(gc_seed)
        rec_facets.append(c)
    if template_length:
        t = TemplateLength(tlen_cap)
        rec_facets.append(t)
    if quality_scores:
        q = QualityScore()
        rec_facets.append(q)
    seq_facets = []
    cov = ed = None
    if coverage:
        cov = Coverage(ref_names, primary, bin_size)
        seq_facets.append(cov)
    if fasta is not None:
        ed = Edits(fasta)
        seq_facets.append(ed)
    if rec_facets:                            # :289-334
        counter = 0
        for rec in records:
            for facet in rec_facets:
                facet.process(rec)
            counter += 1
            if num_records is not None and counter >= num_records:
                break
        for facet in rec_facets:
            facet.summarize()
    if seq_facets:                            # :336-400
        counter = 0                           # :354 -- one counter for the whole pass
        for rid, (name, L) in enumerate(zip(ref_names, ref_len)):
            for facet in seq_facets:
                if facet.supports(name):
                    facet.setup(name, L)
            for rec in query(records, rid, L):
                for facet in seq_facets:
                    if facet.supports(name):
                        facet.process(name, L, rec)
                counter += 1
                if num_records is not None and counter >= num_records:
                    break
            for facet in seq_facets:
                if facet.supports(name):
                    facet.teardown(name, L)
    for facet in seq_facets:                  # :406-416
        facet.aggregate()
    return {"general": g.json() if g else None, "features": f.json() if f else None, "gc_content": c.json() if c else None,
            "template_length": t.json() if t else None, "quality_scores": q.json() if q else None,
            "coverage": cov.json() if cov else None, "edits": ed.json() if ed else None}
