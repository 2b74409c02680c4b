"""A numpy stand-in for the state blocks and the ranged teardown of a QcContext, used ONLY
to drive the library's exchange protocol (ngs_amd/csrc/exchange.cpp, through ngsq_exchange_state with
host memory) on CPU where no GPU exists.  It follows the library's depth-block layout (per sequence:
L+2 difference entries padded to 4096, then one sum per chunk) and the teardown semantics of
coverage.rs:182-246 on a chunk range with a carry."""
from __future__ import annotations

import ctypes as C

import numpy as np

CH = 4096


class FakeCtx:
    def __init__(self, ref_len, bin_size=1000, cov_cap=64):
        self.ref_len = list(ref_len)
        self.bin_size, self.cov_cap = bin_size, cov_cap
        self.first_chunk = [0]
        for L in self.ref_len:
            self.first_chunk.append(self.first_chunk[-1] + -(-(L + 2) // CH))
        self.n_chunks = self.first_chunk[-1]
        self.n_diff = self.n_chunks * CH
        n_super = -(-self.n_chunks // 256)
        self.depth = np.zeros(self.n_diff + self.n_chunks + 3 + n_super + 3, dtype=np.uint32)
        self.nb = cov_cap + 2
        self.bin_off = [0]
        for L in self.ref_len:
            self.bin_off.append(self.bin_off[-1] + 1 + L // bin_size + (L % bin_size != 0))
        self.td = np.zeros(len(ref_len) * self.nb + self.bin_off[-1] + 1, dtype=np.uint64)
        self.counters = np.zeros(8 + len(ref_len), dtype=np.uint64)  # [.. , seen per sequence]
        self.t_lo, self.t_hi = None, None
        self.scan = (0, self.n_chunks, 0)
        self.flags = None  # sorted_input contexts: one byte per chunk, 1 = finished while streaming (state block 4)

    def add_reads(self, ref, starts, ends):
        """+1 at start, -1 at end+1 (1-based positions) for reads on sequence `ref`."""
        off = self.first_chunk[ref] * CH
        for s, e in zip(starts, ends):
            for g, v in ((off + s, 1), (off + e + 1, 0xFFFFFFFF)):
                self.depth[g] += np.uint32(v)
                self.depth[self.n_diff + g // CH] += np.uint32(v)
                self.t_lo = g if self.t_lo is None else min(self.t_lo, g)
                self.t_hi = g + 1 if self.t_hi is None else max(self.t_hi, g + 1)
            self.counters[8 + ref] += 1

    def shard_state(self):
        """ffi.ShardState over the numpy blocks (host memory) with the four operations as callbacks."""
        from ngs_amd import ffi

        def u32(ptr, n):
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint32)), shape=(int(n),))

        def sync(_u):
            return 0

        def halo_add(_u, c0, c1, diff, sums):
            with np.errstate(over="ignore"):
                self.depth[c0 * CH:c1 * CH] += u32(diff, (c1 - c0) * CH)
                self.depth[self.n_diff + c0:self.n_diff + c1] += u32(sums, c1 - c0)
            return 0

        def summary(_u, b0, b1, in_ranges, n_in, out2):
            out = u32(out2, 2)
            out[0] = int(self.depth[self.n_diff + b0:self.n_diff + b1].astype(np.uint64).sum()) & 0xFFFFFFFF
            bad = 0
            if self.flags is not None:
                for k in range(n_in):
                    bad |= int(self.flags[in_ranges[2 * k]:in_ranges[2 * k + 1]].any())
            out[1] = bad
            return 0

        def teardown_range(_u, b0, b1, words, front, part, parts):
            carry = 0
            if front:
                w = u32(words, 2 * 64)
                for r in range(64):
                    if (front >> r) & 1:
                        carry = (carry + int(w[2 * r])) & 0xFFFFFFFF
            self.set_scan_range(b0, b1, carry)
            self.teardown()
            self.vaf_part = (part, parts)
            return 0

        self._touched = np.array([self.t_lo if self.t_lo is not None else 0xFFFFFFFFFFFFFFFF,
                                  self.t_hi if self.t_lo is not None else 0], dtype=np.uint64)
        st = ffi.ShardState()
        st.struct_size = C.sizeof(ffi.ShardState)
        st.memory = ffi.MEM_HOST
        st.counters, st.n_counters = self.counters.ctypes.data, self.counters.size
        st.depth, st.n_depth = self.depth.ctypes.data, self.depth.size
        st.n_diff, st.n_chunks = self.n_diff, self.n_chunks
        st.teardown, st.n_teardown = self.td.ctypes.data, self.td.size
        st.chunk_flags = self.flags.ctypes.data if self.flags is not None else None
        st.touched = self._touched.ctypes.data
        self._cb = (ffi.STATE_SYNC_FN(sync), ffi.STATE_HALO_ADD_FN(halo_add), ffi.STATE_SUMMARY_FN(summary),
                    ffi.STATE_TEARDOWN_FN(teardown_range))
        st.synchronize, st.halo_add, st.summary, st.teardown_range = self._cb
        return st

    def depth_layout(self):
        return self.n_diff, self.n_chunks, self.t_lo or 0, (self.t_hi or 0) if self.t_lo is not None else 0

    def set_scan_range(self, lo, hi, carry):
        self.scan = (lo, hi, carry & 0xFFFFFFFF)

    def teardown(self):
        lo, hi, carry = self.scan
        run = np.uint32(carry)
        for c in range(lo, hi):
            r = max(i for i in range(len(self.ref_len)) if self.first_chunk[i] <= c and self.first_chunk[i + 1] > self.first_chunk[i])
            if not (self.first_chunk[r] <= c < self.first_chunk[r + 1]):
                continue
            seen = self.counters[8 + r] != 0
            L = self.ref_len[r]
            base = (c - self.first_chunk[r]) * CH
            seg = self.depth[c * CH:(c + 1) * CH]
            with np.errstate(over="ignore"):
                d = (np.cumsum(seg.astype(np.uint64)) + np.uint64(run)).astype(np.uint32)
                run = d[-1]
            if not seen:
                continue
            for t in range(CH):
                i = base + t
                if i > L:
                    break
                dep = int(d[t])
                self.td[r * self.nb + (dep if dep <= self.cov_cap else self.cov_cap + 1)] += 1
                b = 0 if i == 0 else 1 + (i - 1) // self.bin_size
                self.td[len(self.ref_len) * self.nb + self.bin_off[r] + b] += dep
            seg[:] = 0
