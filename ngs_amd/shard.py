"""Multi-GPU plumbing of the `ngs qc` scan: one process per GPU, records
sharded by contiguous ranges (= contiguous BGZF block ranges of a sorted BAM), one
integer sum of the shard states before teardown.

Every facet's state after `process` is a sum of per-record integer contributions,
so the exchange is element-wise addition of two blocks per context:
  counters  uint64  (all record-facet tallies/histograms, `seen`, error counts)
  depth     uint32  (coverage difference arrays + their chunk sums; wrap-around mod 2^32)
torch.distributed (backend "nccl" = RCCL over xGMI; "gloo" on CPU for tests) has no
unsigned reductions, so the blocks are viewed as int64 / int32: two's-complement
addition is the same bit pattern.  PyTorch here is plumbing for the collective only.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np


def shard_range(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """(first record, record count) of `rank`: contiguous, balanced, covering [0, n_total)."""
    base, extra = divmod(n_total, world)
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


class _DevArray:
    """Zero-copy __cuda_array_interface__ view of a library-owned device block."""

    def __init__(self, ptr: int, n: int, typestr: str):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}


def device_views(ctx, torch, device_index: int):
    """torch tensors aliasing the context's counters (int64) and depth (int32) blocks."""
    dev = torch.device("cuda", device_index)
    p, n, _ = ctx.state_block(0)
    counters = torch.as_tensor(_DevArray(p, n, "<i8"), device=dev)
    p, n, _ = ctx.state_block(1)
    depth = torch.as_tensor(_DevArray(p, n, "<i4"), device=dev) if n else None
    return counters, depth


def allreduce_state(ctx, dist, torch, views) -> None:
    """Sum the shard states of all ranks in place (RCCL).  Call between the last
    process_batch and finalize; every rank then finalizes the whole-file result."""
    counters, depth = views
    ctx.synchronize()  # the context runs on its own stream
    dist.all_reduce(counters)
    if depth is not None:
        dist.all_reduce(depth)
    torch.cuda.synchronize()


def allreduce_blocks_cpu(blocks: Sequence[np.ndarray], dist, torch) -> List[np.ndarray]:
    """The same exchange on host arrays (gloo): uint64 / uint32 blocks summed with wrap-around."""
    out = []
    for b in blocks:
        assert b.dtype in (np.uint64, np.uint32), b.dtype
        signed = b.view(np.int64 if b.dtype == np.uint64 else np.int32).copy()
        t = torch.from_numpy(signed)
        dist.all_reduce(t)
        out.append(t.numpy().view(b.dtype))
    return out
