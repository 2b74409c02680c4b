"""A numpy stand-in for the state blocks and the ranged teardown of a QcContext, used ONLY
to drive ngs_amd/shard.py's exchange protocol on CPU (gloo) where no GPU exists.  It follows the
library's depth-block layout (per sequence: L+2 difference entries padded to 4096, then one sum
per chunk) and the teardown semantics of coverage.rs:182-246 on a chunk range with a carry."""
from __future__ import annotations

import numpy as np
import torch

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

    def views(self):
        v = {"counters": torch.from_numpy(self.counters.view(np.int64)),
             "depth": torch.from_numpy(self.depth.view(np.int32)),
             "teardown": torch.from_numpy(self.td.view(np.int64))}
        if self.flags is not None:
            v["flags"] = torch.from_numpy(self.flags)
        return v

    def synchronize(self):
        pass

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
