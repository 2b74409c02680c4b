"""Test helpers: build SoA batches from SAM-like record descriptions, random
edge-case batches, and the parity comparison between the HIP path
(ngs_amd.host.QcContext) and the oracle (oracle.oracle_py.Oracle)."""
from __future__ import annotations

import os
import re
import sys
from typing import Dict, List, Optional, Sequence

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from ngs_amd import ffi  # noqa: E402
from ngs_amd.host import HostBatch  # noqa: E402

BASE_CODES = {c: i for i, c in enumerate("=ACMGRSVTWYHKDBN")}
CIGAR_OPS = {c: i for i, c in enumerate("MIDNSHP=X")}


def parse_cigar(s: str) -> List[int]:
    if s in ("*", ""):
        return []
    return [(int(n) << 4) | CIGAR_OPS[o] for n, o in re.findall(r"(\d+)([MIDNSHP=X])", s)]


def pack_seq(seq: str) -> np.ndarray:
    codes = [BASE_CODES[c] for c in seq]
    if len(codes) % 2:
        codes.append(0)
    a = np.array(codes, dtype=np.uint8)
    return ((a[0::2] << 4) | a[1::2]).astype(np.uint8)


def batch_from_records(records: Sequence[dict], first_record_index: int = 0) -> HostBatch:
    """records: dicts with flag, mapq, ref_id, pos (0-based), mate_ref_id, tlen, cigar (str),
    seq (str), qual (list[int] | None for missing)."""
    n = len(records)
    cols: Dict[str, Optional[np.ndarray]] = {
        "flag": np.array([r["flag"] for r in records], dtype=np.uint16),
        "mapq": np.array([r.get("mapq", 255) for r in records], dtype=np.uint8),
        "ref_id": np.array([r.get("ref_id", -1) for r in records], dtype=np.int32),
        "pos": np.array([r.get("pos", -1) for r in records], dtype=np.int32),
        "mate_ref_id": np.array([r.get("mate_ref_id", -1) for r in records], dtype=np.int32),
        "tlen": np.array([r.get("tlen", 0) for r in records], dtype=np.int32),
        "l_seq": np.array([len(r.get("seq", "")) for r in records], dtype=np.uint32),
    }
    cig = [parse_cigar(r.get("cigar", "*")) if isinstance(r.get("cigar", "*"), str) else list(r["cigar"])
           for r in records]
    cols["n_cigar"] = np.array([min(len(c), 65535) for c in cig], dtype=np.uint16)  # saturates (include/ngsq.h); the offsets say the rest
    seqs = [pack_seq(r.get("seq", "")) for r in records]
    quals = [np.array(r["qual"], dtype=np.uint8) if r.get("qual") is not None else np.zeros(0, np.uint8)
             for r in records]

    def cat(parts, dt):
        return np.concatenate(parts).astype(dt) if parts and sum(len(p) for p in parts) else np.zeros(0, dt)

    def offs(parts):
        o = np.zeros(n + 1, dtype=np.uint64)
        if n:
            o[1:] = np.cumsum([len(p) for p in parts])
        return o

    cols["seq"], cols["seq_off"] = cat(seqs, np.uint8), offs(seqs)
    cols["qual"], cols["qual_off"] = cat(quals, np.uint8), offs(quals)
    cols["cigar"], cols["cigar_off"] = cat([np.array(c, dtype=np.uint32) for c in cig], np.uint32), offs(cig)
    return HostBatch(n, cols, 0, 0, 0, first_record_index)


def random_batch(rng: np.random.Generator, n: int, ref_len: Sequence[int], max_len: int = 300,
                 min_len: int = 0, weird: bool = True, first_record_index: int = 0) -> HostBatch:
    """Independent (numpy) random records exercising every branch of the facets:
    all flag bits, all CIGAR op kinds, short/empty reads, missing qualities,
    unplaced records, positions at and beyond the sequence end."""
    nr = len(ref_len)
    flag = rng.integers(0, 1 << 12, n).astype(np.uint16)
    mapq = rng.choice([0, 3, 4, 5, 60, 255], n).astype(np.uint8)
    ref_id = rng.integers(0, nr, n).astype(np.int32)
    mate = np.where(rng.random(n) < 0.8, ref_id, rng.integers(0, nr, n)).astype(np.int32)
    l_seq = rng.integers(min_len, max_len + 1, n).astype(np.uint32)
    if weird:
        l_seq[rng.random(n) < 0.05] = 0
        l_seq[rng.random(n) < 0.10] = 100
        l_seq[rng.random(n) < 0.05] = 99
        l_seq[rng.random(n) < 0.05] = 101
    tlen = rng.choice([0, 1, 350, 1024, 1025, -1, -350, 2 ** 31 - 1, -(2 ** 31)], n).astype(np.int32)
    tlen = np.where(rng.random(n) < 0.5, rng.integers(-100, 1200, n), tlen).astype(np.int32)
    pos = np.zeros(n, dtype=np.int32)
    cig: List[List[int]] = []
    for i in range(n):
        L = int(ref_len[ref_id[i]])
        k = int(rng.integers(0, 5))
        ops = []
        for _ in range(k):
            op = int(rng.integers(0, 9))
            ln = int(rng.integers(0, 60)) if op != 3 else int(rng.integers(0, 400))
            ops.append((ln << 4) | op)
        cig.append(ops)
        r = rng.random()
        if r < 0.70:
            pos[i] = rng.integers(0, max(1, L - 200))
        elif r < 0.85:
            pos[i] = rng.integers(max(0, L - 120), L + 50)  # straddles / beyond the end
        elif r < 0.90:
            pos[i] = 0
        else:
            pos[i] = -1
    if weird:
        unplaced = rng.random(n) < 0.05
        ref_id[unplaced] = -1
        mate[rng.random(n) < 0.05] = -1
    n_cigar = np.array([len(c) for c in cig], dtype=np.uint16)
    seq_len = ((l_seq + 1) // 2).astype(np.uint64)
    seq_off = np.zeros(n + 1, dtype=np.uint64)
    seq_off[1:] = np.cumsum(seq_len)
    # any 4-bit code, biased to ACGT
    codes = rng.choice(np.array([1, 2, 4, 8, 15, 0, 3, 5, 10], dtype=np.uint8), int(seq_off[-1]) * 2,
                       p=[.24, .24, .24, .24, .01, .01, .005, .005, .01]).astype(np.uint8)
    seq = ((codes[0::2] << 4) | codes[1::2]).astype(np.uint8)
    has_q = rng.random(n) >= (0.1 if weird else 0.0)
    qlen = np.where(has_q, l_seq, 0).astype(np.uint64)
    qual_off = np.zeros(n + 1, dtype=np.uint64)
    qual_off[1:] = np.cumsum(qlen)
    qual = rng.integers(0, 94, int(qual_off[-1])).astype(np.uint8)
    cigar_off = np.zeros(n + 1, dtype=np.uint64)
    cigar_off[1:] = np.cumsum(n_cigar)
    flat = [c for ops in cig for c in ops]
    cigar = np.array(flat, dtype=np.uint32) if flat else np.zeros(0, np.uint32)
    cols = dict(flag=flag, mapq=mapq, ref_id=ref_id, pos=pos, mate_ref_id=mate, tlen=tlen, l_seq=l_seq,
                n_cigar=n_cigar, seq=seq, seq_off=seq_off, qual=qual, qual_off=qual_off, cigar=cigar,
                cigar_off=cigar_off)
    return HostBatch(n, cols, 0, 0, 0, first_record_index)


def make_edit_friendly(hb: HostBatch, rng: np.random.Generator, ref_bases: Sequence[np.ndarray],
                       ref_len: Sequence[int]) -> HostBatch:
    """Rewrite CIGARs/positions so the Edits walk succeeds for most records: the read
    length equals the read-consuming ops and the alignment stays inside the sequence."""
    n = hb.n
    cols = dict(hb.cols)
    cig, pos = [], cols["pos"].copy()
    ref_id = cols["ref_id"].copy()
    ref_id[ref_id < 0] = 0
    for i in range(n):
        l = int(cols["l_seq"][i])
        L = int(ref_len[ref_id[i]])
        ops = []
        if l > 0:
            kind = rng.integers(0, 5)
            if kind == 0 or l < 12:
                ops = [(l << 4) | 0]
            elif kind == 1:
                a = int(rng.integers(1, l - 1))
                ops = [(a << 4) | 4, ((l - a) << 4) | 0]
            elif kind == 2:
                a = int(rng.integers(1, l - 5))
                g = int(rng.integers(1, 5))
                ops = [(a << 4) | 0, (g << 4) | 1, ((l - a - g) << 4) | 0]
            elif kind == 3:
                a = int(rng.integers(1, l - 1))
                ops = [(a << 4) | 0, (int(rng.integers(1, 30)) << 4) | 2, ((l - a) << 4) | 7]
            else:
                a = int(rng.integers(1, l - 1))
                ops = [(5 << 4) | 5, (a << 4) | 8, (int(rng.integers(1, 200)) << 4) | 3, ((l - a) << 4) | 0]
        cig.append(ops)
        span = sum(c >> 4 for c in ops if (c & 15) in (0, 2, 3, 7, 8))
        pos[i] = rng.integers(0, max(1, L - span - 1)) if L > span + 2 else 0
    cols["ref_id"] = ref_id
    cols["pos"] = pos
    cols["n_cigar"] = np.array([len(c) for c in cig], dtype=np.uint16)
    off = np.zeros(n + 1, dtype=np.uint64)
    off[1:] = np.cumsum(cols["n_cigar"])
    cols["cigar_off"] = off
    flat = [c for ops in cig for c in ops]
    cols["cigar"] = np.array(flat, dtype=np.uint32) if flat else np.zeros(0, np.uint32)
    # copy the reference into most reads so that edit counts stay small
    seq = cols["seq"].copy()
    for i in range(n):
        if rng.random() < 0.7:
            l = int(cols["l_seq"][i])
            p, r = int(pos[i]), int(ref_id[i])
            chunk = ref_bases[r][p:p + l]
            if len(chunk) == l and l:
                codes = chunk.copy()
                if l % 2:
                    codes = np.append(codes, 0)
                packed = ((codes[0::2] << 4) | codes[1::2]).astype(np.uint8)
                o = int(cols["seq_off"][i])
                seq[o:o + len(packed)] = packed
    cols["seq"] = seq
    return HostBatch(n, cols, 0, 0, 0, hb.first_record_index)


def random_ref_bases(rng: np.random.Generator, ref_len: Sequence[int]) -> List[np.ndarray]:
    return [rng.choice(np.array([1, 2, 4, 8, 15], dtype=np.uint8), int(L), p=[.25, .25, .25, .24, .01])
            for L in ref_len]


def compare_contexts(gpu, orc, n_refs: int, facets: int, bin_size: int, ref_len: Sequence[int]):
    """Bit-exact comparison of every integer result array (HIP path vs oracle)."""
    assert gpu.error_counts() == orc.error_counts()
    if facets & ffi.FACET_GENERAL:
        assert gpu.general() == orc.general()
    if facets & ffi.FACET_TEMPLATE_LENGTH:
        hg, pg, ig = gpu.template_length()
        ho, po, io = orc.template_length()
        assert (pg, ig) == (po, io)
        np.testing.assert_array_equal(hg, ho)
    if facets & ffi.FACET_GC_CONTENT:
        g, o = gpu.gc_content(), orc.gc_content()
        np.testing.assert_array_equal(g.pop("histogram"), o.pop("histogram"))
        assert g == o
    if facets & ffi.FACET_QUALITY_SCORE:
        # (both tables grow with the longest read, in steps of their own: rows nobody reached are zero)
        gq, oq = gpu.quality_scores(), orc.quality_scores()
        rows = max(gq.shape[0], oq.shape[0])
        pad = lambda a: np.vstack([a, np.zeros((rows - a.shape[0], a.shape[1]), a.dtype)])
        np.testing.assert_array_equal(pad(gq), pad(oq))
    if facets & ffi.FACET_COVERAGE:
        assert gpu.coverage_nonsensical() == orc.coverage_nonsensical()
        for r in range(n_refs):
            sg, hg, ig, tg = gpu.coverage_sequence(r)
            so, ho, io, mo = orc.coverage_sequence(r)
            assert sg == so, f"sequence {r}: seen {sg} vs {so}"
            if not sg:
                continue
            np.testing.assert_array_equal(hg, ho)
            assert ig == io
            # integer bin totals -> the reference's f64 means (coverage.rs:217-230)
            L = int(ref_len[r])
            div = np.full(len(tg), float(bin_size))
            if L % bin_size:
                div[-1] = float(L % bin_size)
            np.testing.assert_array_equal(tg.astype(np.float64) / div, mo)
    if facets & ffi.FACET_FEATURES:
        assert gpu.features() == orc.features()
    if facets & ffi.FACET_EDITS:
        for a, b in zip(gpu.edits(), orc.edits()):
            np.testing.assert_array_equal(a, b)


def json_equal(a, b, path=""):
    """Parsed-JSON equality (floats compared exactly; NaN never appears: null)."""
    if isinstance(a, dict):
        assert isinstance(b, dict) and a.keys() == b.keys(), f"{path}: keys {sorted(a)} vs {sorted(b)}"
        for k in a:
            json_equal(a[k], b[k], f"{path}/{k}")
    elif isinstance(a, list):
        assert isinstance(b, list) and len(a) == len(b), f"{path}: length"
        for i, (x, y) in enumerate(zip(a, b)):
            json_equal(x, y, f"{path}[{i}]")
    else:
        assert a == b and type(a) is type(b), f"{path}: {a!r} vs {b!r}"


def to_fixed_stride(hb: HostBatch, min_len: int = 0) -> HostBatch:
    """Re-lay a variable-length batch as fixed-pitch rows (ngsq.h): seq rows padded
    with zero nibbles, qual rows padded with 0xFF (a record with missing qualities
    becomes an all-0xFF row, BAM's own encoding), cigar rows padded with zeros."""
    n = hb.n
    l_seq = hb.cols["l_seq"]
    maxl = max(int(l_seq.max()) if n else 0, min_len, 1)
    sb, qb = (maxl + 1) // 2, maxl
    cs = max(int(hb.cols["n_cigar"].max()) if n else 0, 1)
    seq = np.zeros((n, sb), dtype=np.uint8)
    qual = np.full((n, qb), 0xFF, dtype=np.uint8)
    cigar = np.zeros((n, cs), dtype=np.uint32)
    so, qo, co = hb.cols["seq_off"], hb.cols["qual_off"], hb.cols["cigar_off"]
    for i in range(n):
        a, b = int(so[i]), int(so[i + 1])
        seq[i, :b - a] = hb.cols["seq"][a:b]
        a, b = int(qo[i]), int(qo[i + 1])
        qual[i, :b - a] = hb.cols["qual"][a:b]
        a, b = int(co[i]), int(co[i + 1])
        cigar[i, :b - a] = hb.cols["cigar"][a:b]
    cols = {k: hb.cols[k] for k in FIXED_COLS}
    cols.update(seq=seq.reshape(-1), qual=qual.reshape(-1), cigar=cigar.reshape(-1), seq_off=None, qual_off=None,
                cigar_off=None)
    return HostBatch(n, cols, sb, qb, cs, hb.first_record_index)


FIXED_COLS = ["flag", "mapq", "ref_id", "pos", "mate_ref_id", "tlen", "l_seq", "n_cigar"]


def take_records(hb: HostBatch, order: np.ndarray) -> HostBatch:
    """The records of an offsets-layout batch in another order (or a subset)."""
    order = np.asarray(order, dtype=np.int64)
    cols: Dict[str, Optional[np.ndarray]] = {}
    for name in ("flag", "mapq", "ref_id", "pos", "mate_ref_id", "tlen", "l_seq", "n_cigar"):
        a = hb.cols.get(name)
        cols[name] = None if a is None else np.ascontiguousarray(a[order])
    for data, off in (("seq", "seq_off"), ("qual", "qual_off"), ("cigar", "cigar_off")):
        a, o = hb.cols[data], hb.cols[off]
        assert o is not None, "take_records needs the offsets layout"
        lens = (o[1:] - o[:-1])[order].astype(np.int64)
        new_off = np.zeros(order.size + 1, dtype=np.uint64)
        new_off[1:] = np.cumsum(lens)
        starts = o[:-1][order].astype(np.int64)
        idx = np.repeat(starts - new_off[:-1].astype(np.int64), lens) + np.arange(int(new_off[-1]), dtype=np.int64)
        cols[data] = np.ascontiguousarray(a[idx]) if idx.size else np.zeros(0, a.dtype)
        cols[off] = new_off
    return HostBatch(order.size, cols, 0, 0, 0, hb.first_record_index)


def coordinate_sorted(hb: HostBatch) -> HostBatch:
    """Records in BAM coordinate order: by (ref_id, pos), unplaced (ref_id -1) last; stable."""
    ref = hb.cols["ref_id"].astype(np.int64)
    key = np.where(ref < 0, np.int64(1) << 40, ref << 32 | (hb.cols["pos"].astype(np.int64) + 1))
    return take_records(hb, np.argsort(key, kind="stable"))
