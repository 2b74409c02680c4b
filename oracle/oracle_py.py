"""ORACLE -- TEST INFRASTRUCTURE ONLY.

ctypes wrapper over oracle/libngsq_oracle.so (the CPU restatement of the
reference's `ngs qc` facets, oracle.c).  Importable only from tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg; the product
(ngs_amd/) never imports this module.
"""
from __future__ import annotations

import ctypes as C
import json
import os
import subprocess
from typing import Dict, Optional, Sequence

import numpy as np

from ngs_amd import ffi

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libngsq_oracle.so")

_lib = None


def build(force: bool = False) -> str:
    srcs = [os.path.join(HERE, f) for f in ("oracle.c", "histogram.c", "oracle.h", "histogram.h", "Makefile")]
    srcs += [os.path.join(HERE, "..", "include", f) for f in ("ngsq.h", "ngsq_shared.h")]
    if force or not os.path.exists(LIB_PATH) or any(
            os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs):
        subprocess.run(["make", "-C", HERE, "libngsq_oracle.so"] + (["-B"] if force else []), check=True,
                       stdout=subprocess.DEVNULL)
    return LIB_PATH


class Hist(C.Structure):
    _fields_ = [("values", ffi.u64p), ("range_start", C.c_uint64), ("range_stop", C.c_uint64)]


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    build()
    lib = C.CDLL(LIB_PATH)
    P = C.c_void_p
    lib.orc_create.restype = P
    lib.orc_create.argtypes = [C.POINTER(ffi.Config)]
    lib.orc_destroy.argtypes = [P]
    lib.orc_fasta_base_code.restype = C.c_int
    lib.orc_fasta_base_code.argtypes = [C.c_uint8]
    lib.orc_last_error.restype = C.c_char_p
    lib.orc_last_error.argtypes = [P]
    lib.orc_process_batch.argtypes = [P, C.POINTER(ffi.Batch), C.c_uint32]
    lib.orc_finalize.argtypes = [P]
    lib.orc_get_error_counts.argtypes = [P, C.POINTER(ffi.ErrorCounts)]
    lib.orc_set_features.argtypes = [P, C.POINTER(ffi.Features)]
    lib.orc_get_features.argtypes = [P, C.POINTER(ffi.FeaturesMetrics)]
    lib.orc_get_general.argtypes = [P, C.POINTER(ffi.GeneralMetrics)]
    lib.orc_get_template_length.argtypes = [P, ffi.u64p, C.c_size_t, ffi.u64p, ffi.u64p]
    lib.orc_get_gc_content.argtypes = [P, C.POINTER(ffi.GcMetrics)]
    lib.orc_get_quality_scores.argtypes = [P, ffi.u64p, C.c_size_t]
    lib.orc_quality_rows.argtypes = [P]
    lib.orc_quality_rows.restype = C.c_uint32
    lib.orc_coverage_n_bins.restype = C.c_uint64
    lib.orc_coverage_n_bins.argtypes = [P, C.c_uint32]
    lib.orc_get_coverage_sequence.argtypes = [P, C.c_uint32, C.POINTER(C.c_int), ffi.u64p, C.c_size_t, ffi.u64p,
                                              C.POINTER(C.c_double), C.c_size_t]
    lib.orc_get_coverage_nonsensical.argtypes = [P, ffi.u64p]
    lib.orc_get_edits.argtypes = [P, ffi.u64p, ffi.u64p, C.c_size_t, ffi.u64p, C.c_size_t]
    lib.orc_results_json.restype = C.c_int64
    lib.orc_results_json.argtypes = [P, C.POINTER(C.c_char_p), C.c_char_p, C.c_size_t]
    lib.orc_stepthrough_edits.argtypes = [ffi.u8p, C.c_size_t, ffi.u8p, C.c_size_t, ffi.u32p, C.c_size_t,
                                          ffi.u64p]
    lib.orc_stepthrough_error_message.restype = C.c_char_p
    lib.orc_stepthrough_error_message.argtypes = [C.c_int]
    lib.orc_elapsed_seconds.restype = C.c_double
    lib.orc_elapsed_seconds.argtypes = [P]
    # histogram.c
    HP = C.POINTER(Hist)
    lib.orc_hist_init.argtypes = [HP, C.c_uint64]
    lib.orc_hist_init_default.argtypes = [HP]
    lib.orc_hist_free.argtypes = [HP]
    lib.orc_hist_increment.argtypes = [HP, C.c_uint64]
    lib.orc_hist_increment_by.argtypes = [HP, C.c_uint64, C.c_uint64]
    lib.orc_hist_get.restype = C.c_uint64
    lib.orc_hist_get.argtypes = [HP, C.c_uint64]
    lib.orc_hist_range_len.restype = C.c_uint64
    lib.orc_hist_range_len.argtypes = [HP]
    lib.orc_hist_in_range.argtypes = [HP, C.c_uint64]
    lib.orc_hist_mean.restype = C.c_double
    lib.orc_hist_mean.argtypes = [HP]
    for fn in ("first_quartile", "median", "third_quartile", "interquartile_range"):
        getattr(lib, "orc_hist_" + fn).argtypes = [HP, C.POINTER(C.c_int), C.POINTER(C.c_double)]
    lib.orc_hist_percentile.argtypes = [HP, C.c_double, C.POINTER(C.c_int), C.POINTER(C.c_double)]
    lib.orc_hist_sum.restype = C.c_uint64
    lib.orc_hist_sum.argtypes = [HP]
    lib.orc_hist_count_from_bottom_until.restype = C.c_uint64
    lib.orc_hist_count_from_bottom_until.argtypes = [HP, C.c_uint64]
    lib.orc_hist_count_from_top_until.restype = C.c_uint64
    lib.orc_hist_count_from_top_until.argtypes = [HP, C.c_uint64]
    lib.orc_hist_values_normalized.argtypes = [HP, C.POINTER(C.c_double)]
    _lib = lib
    return lib


class OracleError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"oracle error {code}: {msg}")
        self.code = code


class Oracle:
    """Same lifecycle and result shapes as ngs_amd.host.QcContext, computed on the CPU."""

    def __init__(self, ref_len: Sequence[int], ref_is_primary: Optional[Sequence[int]] = None,
                 facets: int = ffi.FACETS_DEFAULT, bin_size: int = 0, tlen_cap: int = 0, cov_cap: int = 0,
                 max_read_len: int = 0, gc_seed: int = 0, ref_bases=None, ref_bases_len=None):
        self.lib = load()
        self._ref_len = np.asarray(ref_len, dtype=np.uint32)
        self._primary = np.asarray(ref_is_primary if ref_is_primary is not None else [1] * len(ref_len),
                                   dtype=np.uint8)
        cfg = ffi.Config()
        cfg.struct_size = C.sizeof(ffi.Config)
        cfg.facets, cfg.n_refs = facets, len(self._ref_len)
        cfg.ref_len = self._ref_len.ctypes.data_as(ffi.u32p)
        cfg.ref_is_primary = self._primary.ctypes.data_as(ffi.u8p)
        cfg.bin_size, cfg.tlen_cap, cfg.cov_cap, cfg.max_read_len = bin_size, tlen_cap, cov_cap, max_read_len
        cfg.gc_seed = gc_seed
        self._keep = []
        if ref_bases is not None:
            arr = (ffi.u8p * len(self._ref_len))()
            for r, a in enumerate(ref_bases):
                if a is None:
                    arr[r] = None
                else:
                    a = np.ascontiguousarray(a, dtype=np.uint8)
                    self._keep.append(a)
                    arr[r] = a.ctypes.data_as(ffi.u8p)
            self._keep.append(arr)
            cfg.ref_bases = arr
            if ref_bases_len is not None:
                lens = np.asarray(ref_bases_len, dtype=np.uint32)
                self._keep.append(lens)
                cfg.ref_bases_len = lens.ctypes.data_as(ffi.u32p)
        self.max_read_len = max_read_len or 512
        self.tlen_cap = tlen_cap or 1024
        self.cov_cap = cov_cap or 2048
        self._ctx = self.lib.orc_create(C.byref(cfg))
        if not self._ctx:
            raise OracleError(-1, "orc_create failed")

    def close(self):
        if self._ctx:
            self.lib.orc_destroy(self._ctx)
            self._ctx = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def process_batch(self, hb, pass_mask: int = ffi.PASS_BOTH):
        st = hb.struct()
        rc = self.lib.orc_process_batch(self._ctx, C.byref(st), pass_mask)
        if rc:
            raise OracleError(rc, self.lib.orc_last_error(self._ctx).decode())

    def finalize(self, allow_malformed: bool = False) -> int:
        rc = self.lib.orc_finalize(self._ctx)
        if rc in (ffi.ERR_MALFORMED_RECORD, ffi.ERR_LIMIT) and allow_malformed:
            return rc
        if rc:
            raise OracleError(rc, self.lib.orc_last_error(self._ctx).decode())
        return rc

    def elapsed_seconds(self) -> float:
        return float(self.lib.orc_elapsed_seconds(self._ctx))

    def set_features(self, ref_id, name, start, stop, role_name=(0, 1, 2, 3, 4)):
        from ngs_amd import host
        f, keep = host.features_struct(ref_id, name, start, stop, role_name)
        rc = self.lib.orc_set_features(self._ctx, C.byref(f))
        if rc != 0:
            raise RuntimeError(f"orc_set_features: {rc}")

    def features(self) -> Dict[str, int]:
        m = ffi.FeaturesMetrics()
        self.lib.orc_get_features(self._ctx, C.byref(m))
        return {k: int(getattr(m, k)) for k in ffi.FEATURES_FIELDS}

    def error_counts(self) -> Dict[str, int]:
        e = ffi.ErrorCounts()
        self.lib.orc_get_error_counts(self._ctx, C.byref(e))
        return {k: int(getattr(e, k)) for k in ffi.ERROR_FIELDS}

    def general(self):
        g = ffi.GeneralMetrics()
        self.lib.orc_get_general(self._ctx, C.byref(g))
        d = {k: int(getattr(g, k)) for k in ffi.GENERAL_FIELDS}
        d["read_one_cigar_ops"] = [int(x) for x in g.read_one_cigar_ops]
        d["read_two_cigar_ops"] = [int(x) for x in g.read_two_cigar_ops]
        return d

    def template_length(self):
        nb = self.tlen_cap + 1
        h = np.zeros(nb, dtype=np.uint64)
        p, i = C.c_uint64(), C.c_uint64()
        rc = self.lib.orc_get_template_length(self._ctx, h.ctypes.data_as(ffi.u64p), nb, C.byref(p), C.byref(i))
        assert rc == 0
        return h, int(p.value), int(i.value)

    def gc_content(self):
        g = ffi.GcMetrics()
        self.lib.orc_get_gc_content(self._ctx, C.byref(g))
        d = {"histogram": np.array(list(g.histogram), dtype=np.uint64)}
        for k in ("total_gc_count", "total_at_count", "total_other_count", "processed", "ignored_flags",
                  "ignored_too_short"):
            d[k] = int(getattr(g, k))
        return d

    def quality_scores(self, rows: int = 0) -> np.ndarray:
        """[rows][94]; rows = 0: the table's starting size or the longest read met, whichever is larger."""
        rows = rows or max(self.max_read_len, int(self.lib.orc_quality_rows(self._ctx)))
        q = np.zeros((rows, ffi.MAX_SCORE + 1), dtype=np.uint64)
        rc = self.lib.orc_get_quality_scores(self._ctx, q.ctypes.data_as(ffi.u64p), rows)
        assert rc == 0
        return q

    def coverage_sequence(self, ref: int):
        nh = self.cov_cap + 1
        nb = int(self.lib.orc_coverage_n_bins(self._ctx, ref))
        h = np.zeros(nh, dtype=np.uint64)
        means = np.zeros(nb, dtype=np.float64)
        seen, ign = C.c_int(), C.c_uint64()
        rc = self.lib.orc_get_coverage_sequence(self._ctx, ref, C.byref(seen), h.ctypes.data_as(ffi.u64p), nh,
                                                C.byref(ign), means.ctypes.data_as(C.POINTER(C.c_double)), nb)
        assert rc == 0, rc
        return bool(seen.value), h, int(ign.value), means

    def coverage_nonsensical(self) -> int:
        v = C.c_uint64()
        self.lib.orc_get_coverage_nonsensical(self._ctx, C.byref(v))
        return int(v.value)

    def edits(self):
        r1 = np.zeros(ffi.EDITS_BINS, dtype=np.uint64)
        r2 = np.zeros(ffi.EDITS_BINS, dtype=np.uint64)
        vaf = np.zeros(ffi.VAF_BINS, dtype=np.uint64)
        rc = self.lib.orc_get_edits(self._ctx, r1.ctypes.data_as(ffi.u64p), r2.ctypes.data_as(ffi.u64p),
                                    ffi.EDITS_BINS, vaf.ctypes.data_as(ffi.u64p), ffi.VAF_BINS)
        assert rc == 0
        return r1, r2, vaf

    def results_json(self, ref_names: Sequence[str]) -> str:
        names = (C.c_char_p * max(1, len(ref_names)))(*[n.encode() for n in ref_names])
        need = self.lib.orc_results_json(self._ctx, names, None, 0)
        if need < 0:
            raise OracleError(int(need), "results_json")
        buf = C.create_string_buffer(int(need) + 1)
        self.lib.orc_results_json(self._ctx, names, buf, int(need) + 1)
        return buf.value.decode()

    def results(self, ref_names: Sequence[str]) -> dict:
        return json.loads(self.results_json(ref_names))


def fasta_codes(text: bytes) -> np.ndarray:
    """One FASTA sequence's bytes (line terminators already removed) as the oracle wants its ref_bases: [N9]
    orc_fasta_base_code per byte, a refused byte as 0xFF."""
    lib = load()
    table = np.array([lib.orc_fasta_base_code(b) & 0xFF for b in range(256)], dtype=np.uint8)
    return table[np.frombuffer(text, dtype=np.uint8)]
