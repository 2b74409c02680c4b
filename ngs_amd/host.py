"""Harness-side wrappers over the C ABI (ngs_amd/ffi.py): record batches as numpy
columns, a context object with the reference's facet lifecycle
(process -> finalize -> results), and the synthetic-batch helpers.

Nothing here computes: every number comes out of libngsq.so.
"""
from __future__ import annotations

import ctypes as C
import json
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import ffi

COLUMN_DTYPES = {
    "flag": np.uint16, "mapq": np.uint8, "ref_id": np.int32, "pos": np.int32, "mate_ref_id": np.int32,
    "tlen": np.int32, "l_seq": np.uint32, "n_cigar": np.uint16, "seq": np.uint8, "seq_off": np.uint64,
    "qual": np.uint8, "qual_off": np.uint64, "cigar": np.uint32, "cigar_off": np.uint64,
    "record_id": np.uint64,   # optional (include/ngsq.h): absent -> first_record_index + i
}
FIXED_COLUMNS = ["flag", "mapq", "ref_id", "pos", "mate_ref_id", "tlen", "l_seq", "n_cigar"]


class NgsqError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"ngsq error {code}: {message}")
        self.code = code
        self.message = message


@dataclass
class HostBatch:
    """A batch of records as host numpy columns (see include/ngsq.h ngsq_batch)."""
    n: int
    cols: Dict[str, Optional[np.ndarray]]
    seq_stride: int = 0
    qual_stride: int = 0
    cigar_stride: int = 0
    first_record_index: int = 0

    def struct(self) -> ffi.Batch:
        b = ffi.Batch()
        b.struct_size = C.sizeof(ffi.Batch)
        b.location = ffi.MEM_HOST
        b.n_records = self.n
        b.first_record_index = self.first_record_index
        for name, dt in COLUMN_DTYPES.items():
            a = self.cols.get(name)
            if a is None:
                setattr(b, name, None)
                continue
            assert a.dtype == dt and a.flags["C_CONTIGUOUS"], (name, a.dtype)
            setattr(b, name, a.ctypes.data if a.size else _DUMMY.ctypes.data)
        b.seq_stride, b.qual_stride, b.cigar_stride = self.seq_stride, self.qual_stride, self.cigar_stride
        return b

    def slice(self, lo: int, hi: int) -> "HostBatch":
        """Records [lo, hi) as a new batch (variable-length columns re-based)."""
        cols: Dict[str, Optional[np.ndarray]] = {}
        for name in FIXED_COLUMNS:
            a = self.cols.get(name)
            cols[name] = None if a is None else np.ascontiguousarray(a[lo:hi])
        rid = self.cols.get("record_id")
        cols["record_id"] = None if rid is None else np.ascontiguousarray(rid[lo:hi])
        for data, off, stride in (("seq", "seq_off", self.seq_stride), ("qual", "qual_off", self.qual_stride),
                                  ("cigar", "cigar_off", self.cigar_stride)):
            a, o = self.cols.get(data), self.cols.get(off)
            if a is None:
                cols[data], cols[off] = None, None
            elif o is None:
                cols[data], cols[off] = np.ascontiguousarray(a[lo * stride:hi * stride]), None
            else:
                cols[data] = np.ascontiguousarray(a[int(o[lo]):int(o[hi])])
                cols[off] = np.ascontiguousarray(o[lo:hi + 1] - o[lo])
        return HostBatch(hi - lo, cols, self.seq_stride, self.qual_stride, self.cigar_stride,
                         self.first_record_index + lo)


_DUMMY = np.zeros(16, dtype=np.uint64)


def features_struct(ref_id, name, start, stop, role_name=(0, 1, 2, 3, 4)):
    """ffi.Features over numpy copies of the columns (returned second: keep them alive for the call)."""
    cols = [np.ascontiguousarray(np.asarray(a, dtype=np.uint32)) for a in (ref_id, name, start, stop)]
    f = ffi.Features()
    f.struct_size = C.sizeof(ffi.Features)
    for k in range(5):
        f.role_name[k] = int(role_name[k])
    f.n = len(cols[0])
    f.ref_id, f.name, f.start, f.stop = (c.ctypes.data for c in cols)
    return f, cols


def synth_config(n_total: int, mode: int = ffi.SYNTH_FIXED, read_len: int = 150, min_len: int = 50,
                 max_len: int = 300, ref_len: int = 248_956_422, n_refs: int = 2,
                 seed: int = 0x4E4753, file_style: int = 0, seq_model: int = 0, genome=None, lib=None) -> ffi.SynthConfig:
    """genome: the lengths of the sequences of a GENOME-mode file (include/ngsq_shared.h): the records are spread over them."""
    s = ffi.SynthConfig()
    if genome is not None:
        lib = lib or ffi.load_library()
        glen = np.ascontiguousarray(genome, dtype=np.uint32)
        room = np.zeros(len(glen) + 1, dtype=np.uint64)
        _check(lib.ngsq_synth_genome_room(glen.ctypes.data_as(ffi.u32p), len(glen), room.ctypes.data_as(ffi.u64p)), None, lib)
        s.genome_len, s.genome_room, s.genome_n = glen.ctypes.data_as(ffi.u32p), room.ctypes.data_as(ffi.u64p), len(glen)
        s._keep = (glen, room)   # (the structure holds pointers into them)
    s.file_style, s.seq_model = file_style, seq_model
    s.seed, s.n_total, s.mode, s.read_len = seed, n_total, mode, read_len
    s.min_len, s.max_len, s.ref_len, s.n_refs = min_len, max_len, ref_len, n_refs
    return s


def synth_reference(cfg: ffi.SynthConfig, ref: int, length: int, lib=None) -> np.ndarray:
    """The synthetic reference sequence `ref` (one 4-bit code per byte): what reads of seq_model FROM_REFERENCE are sampled from."""
    lib = lib or ffi.load_library()
    out = np.empty(length, dtype=np.uint8)
    _check(lib.ngsq_synth_fill_reference(C.byref(cfg), ref, out.ctypes.data, length, 0), None, lib)
    return out


def synth_host_batch(cfg: ffi.SynthConfig, first: int, n: int, lib=None) -> HostBatch:
    """Records [first, first+n) of the synthetic file, generated by the library's host loop."""
    lib = lib or ffi.load_library()
    sb, qb, co = C.c_uint64(), C.c_uint64(), C.c_uint64()
    _check(lib.ngsq_synth_sizes(C.byref(cfg), first, n, C.byref(sb), C.byref(qb), C.byref(co)), None, lib)
    cols: Dict[str, Optional[np.ndarray]] = {k: np.zeros(n, dtype=COLUMN_DTYPES[k]) for k in FIXED_COLUMNS}
    cols["seq"] = np.zeros(sb.value, dtype=np.uint8)
    cols["qual"] = np.zeros(qb.value, dtype=np.uint8)
    if cfg.mode == ffi.SYNTH_FIXED:
        cs = 3 if cfg.file_style & ffi.SYNTH_FILE_CIGAR_MIX else 1
        cols["cigar"] = np.zeros(n * cs, dtype=np.uint32)
        cols["seq_off"] = cols["qual_off"] = cols["cigar_off"] = None
        hb = HostBatch(n, cols, (cfg.read_len + 1) // 2, cfg.read_len, cs, first)
    else:
        cols["cigar"] = np.zeros(co.value, dtype=np.uint32)
        for k in ("seq_off", "qual_off", "cigar_off"):
            cols[k] = np.zeros(n + 1, dtype=np.uint64)
        hb = HostBatch(n, cols, 0, 0, 0, first)
    st = hb.struct()
    _check(lib.ngsq_synth_fill_host(C.byref(cfg), first, n, C.byref(st)), None, lib)
    return hb


def _check(rc: int, ctx, lib):
    if rc != ffi.OK:
        msg = (lib.ngsq_last_error(ctx) if ctx else lib.ngsq_last_global_error()) or b""
        raise NgsqError(rc, msg.decode("utf-8", "replace"))


@dataclass
class DeviceBatch:
    """A batch whose columns live in device memory owned by a QcContext."""
    n: int
    ptrs: Dict[str, int]
    seq_stride: int
    qual_stride: int
    cigar_stride: int
    first_record_index: int
    seq_bytes: int
    qual_bytes: int
    cigar_ops: int
    nbytes: int = 0
    max_l_seq: int = 0   # the longest read, when known (include/ngsq.h: the quality table grows to it)

    def struct(self) -> ffi.Batch:
        b = ffi.Batch()
        b.struct_size = C.sizeof(ffi.Batch)
        b.location = ffi.MEM_DEVICE
        b.max_l_seq = self.max_l_seq
        b.n_records = self.n
        b.first_record_index = self.first_record_index
        for name in COLUMN_DTYPES:
            setattr(b, name, self.ptrs.get(name) or None)
        b.seq_stride, b.qual_stride, b.cigar_stride = self.seq_stride, self.qual_stride, self.cigar_stride
        b.seq_bytes, b.qual_bytes, b.cigar_ops = self.seq_bytes, self.qual_bytes, self.cigar_ops
        return b


class QcContext:
    """One `ngs qc` scan on one GPU: create -> process_batch* -> finalize -> results.

    Mirrors the reference driver's lifecycle (src/qc/command.rs:288-418).
    """

    def __init__(self, ref_len: Sequence[int], ref_is_primary: Optional[Sequence[int]] = None,
                 facets: int = ffi.FACETS_DEFAULT, device: int = 0, bin_size: int = 0, tlen_cap: int = 0,
                 cov_cap: int = 0, max_read_len: int = 0, gc_seed: int = 0,
                 ref_bases: Optional[Sequence[Optional[np.ndarray]]] = None, stream: int = 0,
                 timing: bool = False, sorted_input: bool = False, cov_head_guard: int = 0, lib=None,
                 ref_bases_len: Optional[Sequence[int]] = None, ref_fasta: Optional[str] = None,
                 ref_names: Optional[Sequence[str]] = None, ref_wanted: Optional[Sequence[int]] = None, fasta_threads: int = 0):
        """ref_fasta (+ ref_names): the Edits reference from a FASTA FILE (include/ngsq_reference.h) instead of ref_bases."""
        self.lib = lib or ffi.load_library()
        self._ref_len = np.asarray(ref_len, dtype=np.uint32)
        self._primary = np.asarray(ref_is_primary if ref_is_primary is not None else [1] * len(ref_len),
                                   dtype=np.uint8)
        cfg = ffi.Config()
        cfg.struct_size = C.sizeof(ffi.Config)
        cfg.facets, cfg.device, cfg.n_refs = facets, device, len(self._ref_len)
        cfg.ref_len = self._ref_len.ctypes.data_as(ffi.u32p)
        cfg.ref_is_primary = self._primary.ctypes.data_as(ffi.u8p)
        cfg.bin_size, cfg.tlen_cap, cfg.cov_cap, cfg.max_read_len = bin_size, tlen_cap, cov_cap, max_read_len
        cfg.gc_seed = gc_seed
        self._bases_keep = None
        if ref_bases is not None:
            arr = (ffi.u8p * len(self._ref_len))()
            keep = []
            for r, a in enumerate(ref_bases):
                if a is None:
                    arr[r] = None
                else:
                    a = np.ascontiguousarray(a, dtype=np.uint8)
                    # (with ref_bases_len the array may hold the FASTA's whole sequence: only min(its length, LN) bases are read)
                    assert a.size == int(self._ref_len[r]) if ref_bases_len is None else a.size >= min(int(ref_bases_len[r]), int(self._ref_len[r]))
                    keep.append(a)
                    arr[r] = a.ctypes.data_as(ffi.u8p)
            self._bases_keep = (arr, keep)
            cfg.ref_bases = arr
            if ref_bases_len is not None:
                self._bases_len = np.asarray(ref_bases_len, dtype=np.uint32)
                cfg.ref_bases_len = self._bases_len.ctypes.data_as(ffi.u32p)
        self._fasta = None
        if ref_fasta is not None:
            assert ref_bases is None and ref_names is not None and len(ref_names) == len(self._ref_len)
            cfg.ref_bases_deferred = 1
            self._fasta = C.c_void_p()
            if self.lib.ngsq_fasta_open(ref_fasta.encode(), fasta_threads, C.byref(self._fasta)) != 0:
                raise NgsqError(ffi.ERR_INVALID_ARGUMENT, (self.lib.ngsq_fasta_last_error() or b"").decode())
        cfg.stream = stream or None
        cfg.timing = 1 if timing else 0
        cfg.sorted_input = 1 if sorted_input else 0
        cfg.cov_head_guard = cov_head_guard
        self.facets = facets
        self._cfg = cfg
        self._ctx = ffi.ctx_p()
        _check(self.lib.ngsq_create(C.byref(cfg), C.byref(self._ctx)), None, self.lib)
        self._allocs: List[int] = []
        if self._fasta is not None:
            names = (C.c_char_p * len(ref_names))(*[n.encode() for n in ref_names])
            wanted = np.asarray(ref_wanted, dtype=np.uint8) if ref_wanted is not None else None
            _check(self.lib.ngsq_reference_load(self._ctx, self._fasta, names, wanted.ctypes.data_as(ffi.u8p) if wanted is not None else None),
                   self._ctx, self.lib)

    def reference_wait(self) -> Dict[str, float]:
        """Wait for the FASTA load (ngsq_process_batch does so by itself); what it did."""
        _check(self.lib.ngsq_reference_wait(self._ctx), self._ctx, self.lib)
        st = ffi.ReferenceStats()
        _check(self.lib.ngsq_reference_get_stats(self._ctx, C.byref(st)), self._ctx, self.lib)
        return {k: getattr(st, k) for k, _ in ffi.ReferenceStats._fields_ if k != "reserved"}

    # -- lifecycle
    def close(self):
        if self._ctx:
            for p in self._allocs:
                self.lib.ngsq_device_free(self._ctx, p)
            self._allocs = []
            self.lib.ngsq_destroy(self._ctx)   # (waits for a reference load that is still on its way)
            self._ctx = ffi.ctx_p()
            if getattr(self, "_fasta", None) is not None:
                self.lib.ngsq_fasta_close(self._fasta)
                self._fasta = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def process_batch(self, batch, pass_mask: int = ffi.PASS_BOTH):
        st = batch.struct()
        _check(self.lib.ngsq_process_batch(self._ctx, C.byref(st), pass_mask), self._ctx, self.lib)

    def finalize(self, allow_malformed: bool = False) -> int:
        rc = self.lib.ngsq_finalize(self._ctx)
        if rc in (ffi.ERR_MALFORMED_RECORD, ffi.ERR_LIMIT) and allow_malformed:
            return rc
        _check(rc, self._ctx, self.lib)
        return rc

    def reset(self):
        _check(self.lib.ngsq_reset(self._ctx), self._ctx, self.lib)

    def synchronize(self):
        _check(self.lib.ngsq_synchronize(self._ctx), self._ctx, self.lib)

    @property
    def stream(self) -> int:
        return self.lib.ngsq_stream(self._ctx) or 0

    # -- device memory / batches
    def device_malloc(self, nbytes: int) -> int:
        p = C.c_void_p()
        _check(self.lib.ngsq_device_malloc(self._ctx, nbytes, C.byref(p)), self._ctx, self.lib)
        self._allocs.append(p.value)
        return p.value

    def device_free(self, ptr: int):
        self._allocs.remove(ptr)
        _check(self.lib.ngsq_device_free(self._ctx, ptr), self._ctx, self.lib)

    def upload(self, hb: HostBatch) -> DeviceBatch:
        """Copy a host batch into device memory (test helper; the product path
        for host data is process_batch on the HostBatch itself)."""
        ptrs, total = {}, 0
        for name, a in hb.cols.items():
            if a is None:
                continue
            p = self.device_malloc(max(a.nbytes, 16))
            _check(self.lib.ngsq_memcpy_h2d(self._ctx, p, a.ctypes.data, a.nbytes), self._ctx, self.lib)
            ptrs[name] = p
            total += a.nbytes
        def tot(data, off, stride, itemsize=1):
            a, o = hb.cols.get(data), hb.cols.get(off)
            if a is None:
                return 0
            return int(o[hb.n]) if o is not None else hb.n * stride
        return DeviceBatch(hb.n, ptrs, hb.seq_stride, hb.qual_stride, hb.cigar_stride, hb.first_record_index,
                           tot("seq", "seq_off", hb.seq_stride), tot("qual", "qual_off", hb.qual_stride),
                           tot("cigar", "cigar_off", hb.cigar_stride), total,
                           int(hb.cols["l_seq"].max()) if hb.n and hb.cols.get("l_seq") is not None else 0)

    def synth_device_batch(self, cfg: ffi.SynthConfig, first: int, n: int) -> DeviceBatch:
        """Generate records [first, first+n) of the synthetic file directly in HBM."""
        sb, qb, co = C.c_uint64(), C.c_uint64(), C.c_uint64()
        _check(self.lib.ngsq_synth_sizes(C.byref(cfg), first, n, C.byref(sb), C.byref(qb), C.byref(co)),
               self._ctx, self.lib)
        sizes = {"flag": 2 * n, "mapq": n, "ref_id": 4 * n, "pos": 4 * n, "mate_ref_id": 4 * n, "tlen": 4 * n,
                 "l_seq": 4 * n, "n_cigar": 2 * n, "seq": sb.value, "qual": qb.value}
        if cfg.mode == ffi.SYNTH_FIXED:
            cs = 3 if cfg.file_style & ffi.SYNTH_FILE_CIGAR_MIX else 1    # (an aligner's CIGAR mix: up to three operations, fixed pitch)
            sizes["cigar"] = 4 * n * cs
            strides = ((cfg.read_len + 1) // 2, cfg.read_len, cs)
        else:
            sizes["cigar"] = 4 * co.value
            for k in ("seq_off", "qual_off", "cigar_off"):
                sizes[k] = 8 * (n + 1)
            strides = (0, 0, 0)
        ptrs = {k: self.device_malloc(v + 64) for k, v in sizes.items()}
        db = DeviceBatch(n, ptrs, strides[0], strides[1], strides[2], first, sb.value, qb.value,
                         co.value if cfg.mode != ffi.SYNTH_FIXED else n * strides[2], sum(sizes.values()))
        st = db.struct()
        _check(self.lib.ngsq_synth_fill_device(self._ctx, C.byref(cfg), first, n, C.byref(st)), self._ctx,
               self.lib)
        return db

    def download_column(self, db: DeviceBatch, name: str, count: int) -> np.ndarray:
        out = np.zeros(count, dtype=COLUMN_DTYPES[name])
        _check(self.lib.ngsq_memcpy_d2h(self._ctx, out.ctypes.data, db.ptrs[name], out.nbytes), self._ctx,
               self.lib)
        return out

    def free_batch(self, db: DeviceBatch):
        for p in db.ptrs.values():
            self.device_free(p)
        db.ptrs = {}

    # -- results
    def set_features(self, ref_id, name, start, stop, role_name=(0, 1, 2, 3, 4)):
        """Gene model of the Genomic Features facet: intervals (sequence index, name id, GFF start, GFF end)."""
        f, keep = features_struct(ref_id, name, start, stop, role_name)
        _check(self.lib.ngsq_set_features(self._ctx, C.byref(f)), self._ctx, self.lib)
        del keep

    def features(self) -> Dict[str, int]:
        m = ffi.FeaturesMetrics()
        _check(self.lib.ngsq_get_features(self._ctx, C.byref(m)), self._ctx, self.lib)
        return {k: int(getattr(m, k)) for k in ffi.FEATURES_FIELDS}

    def error_counts(self) -> Dict[str, int]:
        e = ffi.ErrorCounts()
        _check(self.lib.ngsq_get_error_counts(self._ctx, C.byref(e)), self._ctx, self.lib)
        return {k: int(getattr(e, k)) for k in ffi.ERROR_FIELDS}

    def general(self) -> Dict[str, object]:
        g = ffi.GeneralMetrics()
        _check(self.lib.ngsq_get_general(self._ctx, C.byref(g)), self._ctx, self.lib)
        d: Dict[str, object] = {k: int(getattr(g, k)) for k in ffi.GENERAL_FIELDS}
        d["read_one_cigar_ops"] = [int(x) for x in g.read_one_cigar_ops]
        d["read_two_cigar_ops"] = [int(x) for x in g.read_two_cigar_ops]
        return d

    def template_length(self):
        nb = self.lib.ngsq_tlen_bins(self._ctx)
        h = np.zeros(nb, dtype=np.uint64)
        p, i = C.c_uint64(), C.c_uint64()
        _check(self.lib.ngsq_get_template_length(self._ctx, h.ctypes.data_as(ffi.u64p), nb, C.byref(p),
                                                 C.byref(i)), self._ctx, self.lib)
        return h, int(p.value), int(i.value)

    def gc_content(self) -> Dict[str, object]:
        g = ffi.GcMetrics()
        _check(self.lib.ngsq_get_gc_content(self._ctx, C.byref(g)), self._ctx, self.lib)
        d: Dict[str, object] = {"histogram": np.array(list(g.histogram), dtype=np.uint64)}
        for k in ("total_gc_count", "total_at_count", "total_other_count", "processed", "ignored_flags",
                  "ignored_too_short"):
            d[k] = int(getattr(g, k))
        return d

    def quality_scores(self) -> np.ndarray:
        rows = self.lib.ngsq_max_read_len(self._ctx)
        q = np.zeros((rows, ffi.MAX_SCORE + 1), dtype=np.uint64)
        _check(self.lib.ngsq_get_quality_scores(self._ctx, q.ctypes.data_as(ffi.u64p), rows), self._ctx,
               self.lib)
        return q

    def coverage_sequence(self, ref: int):
        nh = self.lib.ngsq_cov_bins(self._ctx)
        nb = int(self.lib.ngsq_coverage_n_bins(self._ctx, ref))
        h = np.zeros(nh, dtype=np.uint64)
        bins = np.zeros(nb, dtype=np.uint64)
        seen, ign = C.c_int(), C.c_uint64()
        _check(self.lib.ngsq_get_coverage_sequence(self._ctx, ref, C.byref(seen), h.ctypes.data_as(ffi.u64p),
                                                   nh, C.byref(ign), bins.ctypes.data_as(ffi.u64p), nb),
               self._ctx, self.lib)
        return bool(seen.value), h, int(ign.value), bins

    def coverage_nonsensical(self) -> int:
        v = C.c_uint64()
        _check(self.lib.ngsq_get_coverage_nonsensical(self._ctx, C.byref(v)), self._ctx, self.lib)
        return int(v.value)

    def edits(self):
        r1 = np.zeros(ffi.EDITS_BINS, dtype=np.uint64)
        r2 = np.zeros(ffi.EDITS_BINS, dtype=np.uint64)
        vaf = np.zeros(ffi.VAF_BINS, dtype=np.uint64)
        _check(self.lib.ngsq_get_edits(self._ctx, r1.ctypes.data_as(ffi.u64p), r2.ctypes.data_as(ffi.u64p),
                                       ffi.EDITS_BINS, vaf.ctypes.data_as(ffi.u64p), ffi.VAF_BINS),
               self._ctx, self.lib)
        return r1, r2, vaf

    def edits_positions(self, ref: int):
        """refs_per_position / alts_per_position of one sequence (edits.rs:322-324), ref_len + 1 entries each."""
        n = int(self._ref_len[ref]) + 1
        refs, alts = np.zeros(n, dtype=np.uint32), np.zeros(n, dtype=np.uint32)
        _check(self.lib.ngsq_get_edits_positions(self._ctx, ref, refs.ctypes.data, alts.ctypes.data, n), self._ctx, self.lib)
        return refs, alts

    def results_json(self, ref_names: Sequence[str]) -> str:
        if len(ref_names) < len(self._ref_len):   # (the C entry point reads n_refs names: it cannot know how many it was given)
            raise ValueError(f"{len(ref_names)} names for a context of {len(self._ref_len)} sequences")
        names = (C.c_char_p * max(1, len(ref_names)))(*[n.encode() for n in ref_names])
        need = self.lib.ngsq_results_json(self._ctx, names, None, 0)
        if need < 0:
            _check(int(need), self._ctx, self.lib)
        buf = C.create_string_buffer(int(need) + 1)
        self.lib.ngsq_results_json(self._ctx, names, buf, int(need) + 1)
        return buf.value.decode()

    def results(self, ref_names: Sequence[str]) -> dict:
        return json.loads(self.results_json(ref_names))

    # -- measurement
    def kernel_timing(self) -> Dict[str, Dict[str, float]]:
        out = {}
        for i in range(self.lib.ngsq_kernel_timing_count(self._ctx)):
            kt = ffi.KernelTime()
            _check(self.lib.ngsq_kernel_timing(self._ctx, i, C.byref(kt)), self._ctx, self.lib)
            out[kt.name.decode()] = {"launches": int(kt.launches), "total_ms": float(kt.total_ms),
                                     "algo_bytes": int(kt.algo_bytes)}
        return out

    def kernel_timing_reset(self):
        _check(self.lib.ngsq_kernel_timing_reset(self._ctx), self._ctx, self.lib)

    # -- shard exchange
    def state_block(self, which: int):
        """(device pointer, element count, numpy dtype) of state block 0=counters 1=depth 2=edits 3=teardown
        4=chunk flags (sorted_input contexts)."""
        p, n = C.c_void_p(), C.c_uint64()
        fn = [self.lib.ngsq_state_counters, self.lib.ngsq_state_depth, self.lib.ngsq_state_edits,
              self.lib.ngsq_state_teardown, self.lib.ngsq_state_chunk_flags][which]
        _check(fn(self._ctx, C.byref(p), C.byref(n)), self._ctx, self.lib)
        return (p.value or 0), int(n.value), (np.uint64 if which in (0, 3) else np.uint8 if which == 4 else np.uint32)

    def depth_layout(self):
        """(n_diff, n_chunks, touched_lo, touched_hi): see ngsq_depth_layout."""
        v = [C.c_uint64() for _ in range(4)]
        _check(self.lib.ngsq_depth_layout(self._ctx, *[C.byref(x) for x in v]), self._ctx, self.lib)
        return tuple(int(x.value) for x in v)

    def set_scan_range(self, chunk_lo: int, chunk_hi: int, carry_in: int):
        _check(self.lib.ngsq_set_scan_range(self._ctx, chunk_lo, chunk_hi, carry_in & 0xFFFFFFFF), self._ctx, self.lib)

    def teardown(self):
        _check(self.lib.ngsq_teardown(self._ctx), self._ctx, self.lib)

    def state_download(self, which: int) -> np.ndarray:
        _, n, dt = self.state_block(which)
        a = np.zeros(n, dtype=dt)
        _check(self.lib.ngsq_state_download(self._ctx, which, a.ctypes.data, a.nbytes), self._ctx, self.lib)
        return a

    def state_upload(self, which: int, a: np.ndarray):
        a = np.ascontiguousarray(a)
        _check(self.lib.ngsq_state_upload(self._ctx, which, a.ctypes.data, a.nbytes), self._ctx, self.lib)
