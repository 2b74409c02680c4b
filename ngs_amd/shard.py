"""Multi-GPU plumbing of the `ngs qc` scan: one process per GPU, records
sharded by contiguous ranges (= contiguous BGZF block ranges of a sorted BAM), and one
exchange of integer state before the sequence-facet teardown.

Every facet's state after `process` is a sum of per-record integer contributions.  The
record-facet state (`counters`, ~130 KB) is summed with one all-reduce.  The coverage
state (`depth`: ~1 GB of difference entries for chr1) is NOT all-reduced: in a
coordinate-sorted file a shard writes only its own stretch of the reference axis, so

  1. ranks all-gather the chunk range [lo, hi) they wrote (a chunk = 4096 entries);
  2. the axis is cut at the sorted `lo`s: rank k OWNS [lo_k, lo_next) -- a disjoint cover;
  3. entries a rank wrote inside another rank's range (the read-length halo at a shard
     boundary, a few KB) are sent to the owner and added there;
  4. the carry of owner k is the sum of the ranges in front of it (every difference array sums
     to zero per sequence, so one running sum over the whole block is enough): the sum of a
     rank's own range before the halos travels in the halo message of step 3, and since every
     rank sees every halo it adds the incoming parts itself -- no further collective;
  5. every rank tears down only its own chunks (ngsq_set_scan_range) -- the scan is split
     N ways -- and the small teardown results (depth histograms, bin totals) are all-reduced.

If the shards are not sorted (written ranges overlap so much that step 3 would move more than
`HALO_LIMIT_BYTES`), the protocol falls back to all-reducing the whole depth block.

Contexts created with `sorted_input` (streaming Coverage, csrc/cov_stream.hip) have already finished
most chunks of their stretch while scanning: those chunks are flagged, hold no entries and are skipped
by the teardown; only the seams (the first `cov_head_guard` positions of a shard, its last few reads,
sequence boundaries) are on the depth array and take part in steps 1-5 unchanged.  What must not
happen is another shard's entry landing in a flagged chunk -- a read of the shard in front reaching
beyond the guard, or shards that are not in coordinate order: that is detected here (every rank
raises) instead of giving a wrong depth.

torch.distributed ("nccl" = RCCL over xGMI; "gloo" for CPU tests) has no unsigned reductions:
blocks are viewed as int64 / int32, two's-complement addition is the same bit pattern.
PyTorch is plumbing for the collectives only; nothing here computes facet results.
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import numpy as np

COV_CHUNK = 4096                 # ngs_amd/csrc/kernels.h COV_CHUNK
HALO_LIMIT_BYTES = 64 << 20      # per rank; beyond this the depth block is all-reduced instead


def shard_range(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """(first record, record count) of `rank`: contiguous, balanced, covering [0, n_total)."""
    base, extra = divmod(n_total, world)
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


class _DevArray:
    """Zero-copy __cuda_array_interface__ view of a library-owned device block."""

    def __init__(self, ptr: int, n: int, typestr: str):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}


def device_views(ctx, torch, device_index: int) -> Dict[str, object]:
    """torch tensors aliasing the context's state blocks: counters (int64), depth (int32),
    teardown (int64)."""
    dev = torch.device("cuda", device_index)
    out = {}
    for name, which, ts in (("counters", 0, "<i8"), ("depth", 1, "<i4"), ("teardown", 3, "<i8"), ("flags", 4, "|u1")):
        p, n, _ = ctx.state_block(which)
        out[name] = torch.as_tensor(_DevArray(p, n, ts), device=dev) if n else None
    return out


def allreduce_state(ctx, dist, torch, views) -> None:
    """The simple exchange: sum counters and the WHOLE depth block of all ranks in place.
    Call between the last process_batch and finalize."""
    ctx.synchronize()  # the context runs on its own stream
    dist.all_reduce(views["counters"])
    if views.get("depth") is not None:
        dist.all_reduce(views["depth"])
    if torch.cuda.is_available():
        torch.cuda.synchronize()


def plan_owners(ranges: Sequence[Tuple[int, int]], n_chunks: int):
    """From every rank's written chunk range -> (owned range per rank, transfers).

    ranges[r] = (lo, hi) chunks, lo == hi for a rank that wrote nothing.
    Returns own[r] = (b0, b1) and xfer[(src, dst)] = (c0, c1): chunks src sends to dst.
    Pure function: every rank computes the same plan from the all-gathered ranges."""
    world = len(ranges)
    owners = sorted((r for r in range(world) if ranges[r][1] > ranges[r][0]), key=lambda r: (ranges[r][0], r))
    own = [(0, 0)] * world
    for k, r in enumerate(owners):
        b0 = 0 if k == 0 else ranges[r][0]
        b1 = n_chunks if k + 1 == len(owners) else ranges[owners[k + 1]][0]
        own[r] = (b0, max(b0, b1))
    xfer = {}
    for s in range(world):
        lo, hi = ranges[s]
        if hi <= lo:
            continue
        for d in owners:
            if d == s:
                continue
            c0, c1 = max(lo, own[d][0]), min(hi, own[d][1])
            if c1 > c0:
                xfer[(s, d)] = (c0, c1)
    return own, owners, xfer


def _all_gather_ints(vals: List[int], dist, torch, device) -> List[List[int]]:
    t = torch.tensor(vals, dtype=torch.int64, device=device)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [[int(x) for x in o.tolist()] for o in out]


def owner_teardown(ctx, dist, torch, views, coll_device=None) -> dict:
    """Steps 1-5 of the module docstring; leaves the context torn down (call ctx.finalize()
    afterwards).  `coll_device`: device the collectives run on ("cpu" for gloo: small tensors
    are staged through the host); default = where the views live.  Returns a small report."""
    rank, world = dist.get_rank(), dist.get_world_size()
    counters, depth, td = views["counters"], views.get("depth"), views["teardown"]
    dev = counters.device
    cdev = torch.device(coll_device) if coll_device is not None else dev
    staged = cdev != dev

    def allreduce_(t):
        if staged:
            h = t.to(cdev)
            dist.all_reduce(h)
            t.copy_(h.to(dev))
        else:
            dist.all_reduce(t)

    def torch_done():  # the context launches on its own stream: torch's work must have landed first
        if dev.type == "cuda":
            torch.cuda.synchronize()

    ctx.synchronize()
    allreduce_(counters)
    report = {"mode": "none", "halo_bytes": 0}
    if depth is None or depth.numel() == 0:
        torch_done()
        ctx.teardown()
        ctx.synchronize()
        return report
    n_diff, n_chunks, t_lo, t_hi = ctx.depth_layout()
    lo = t_lo // COV_CHUNK
    hi = min(n_chunks, -(-t_hi // COV_CHUNK)) if t_hi > t_lo else lo
    ranges = [(a, b) for a, b in _all_gather_ints([lo, hi], dist, torch, cdev)]
    own, owners, xfer = plan_owners(ranges, n_chunks)
    out_bytes = [0] * world
    for (s, d), (c0, c1) in xfer.items():
        out_bytes[s] += (c1 - c0) * (COV_CHUNK + 1) * 4
    diff = depth[:n_diff]
    sums = depth[n_diff:n_diff + n_chunks]

    # streaming contexts: no exchanged entry may fall into a chunk this rank has already finished (the verdict
    # travels with the halo message below; with the all-reduce fallback it needs a message of its own)
    flags = views.get("flags")
    bad = 0
    if flags is not None and flags.numel():
        for (s, d), (c0, c1) in xfer.items():
            if d == rank and bool(flags[c0:c1].any().item()):
                bad = 1
        if max(out_bytes) > HALO_LIMIT_BYTES and bool(flags.any().item()):
            bad = 1
    overlap = RuntimeError("sorted_input shards overlap: records of another shard reach into positions this shard "
                           "already finished (cov_head_guard too small, or the shards are not in coordinate "
                           "order); re-run without sorted_input")

    if max(out_bytes) > HALO_LIMIT_BYTES:
        if flags is not None and flags.numel() and max(b for row in _all_gather_ints([bad], dist, torch, cdev) for b in row):
            raise overlap
        # unsorted shards: the written ranges overlap -- sum the whole block, every rank scans all
        allreduce_(depth)
        report["mode"] = "allreduce"
        torch_done()
        ctx.teardown()
        ctx.synchronize()
        return report

    # ---- step 3: halo entries to their owners (all-gather of the padded outgoing buffers:
    # a few KB per rank for sorted shards; works on every backend)
    mine = [(d, c0, c1) for (s, d), (c0, c1) in sorted(xfer.items()) if s == rank]
    parts = []
    for d, c0, c1 in mine:
        parts.append(diff[c0 * COV_CHUNK:c1 * COV_CHUNK])
        parts.append(sums[c0:c1])
    # The same message carries two more words: the sum of this rank's owned range BEFORE the halos arrive (step 4
    # needs the sums after; every rank sees every halo, so it can add the incoming parts itself -- one collective and
    # one device-to-host sync less per step) and the verdict of the check above.
    b0, b1 = own[rank]
    pre_total = int(sums[b0:b1].sum().item()) & 0xFFFFFFFF if b1 > b0 else 0
    max_len = max(max(out_bytes) // 4, 1)
    buf = torch.zeros(max_len + 2, dtype=torch.int32, device=cdev)
    if parts:
        flat = torch.cat([p.to(cdev) for p in parts])
        buf[:flat.numel()] = flat
    buf[max_len] = pre_total - (1 << 32) if pre_total >= (1 << 31) else pre_total  # two's complement: the same bits
    buf[max_len + 1] = bad
    gathered = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(gathered, buf)
    moved = []  # (owner, device scalar: sum of the chunk sums one halo carries), read back in one transfer below
    for s in range(world):
        off = 0
        for (s2, d), (c0, c1) in sorted(xfer.items()):
            if s2 != s:
                continue
            n_e, n_c = (c1 - c0) * COV_CHUNK, c1 - c0
            part_sums = gathered[s][off + n_e:off + n_e + n_c]
            moved.append((d, part_sums.sum()))
            if d == rank and s != rank:
                diff[c0 * COV_CHUNK:c1 * COV_CHUNK] += gathered[s][off:off + n_e].to(dev)
                sums[c0:c1] += part_sums.to(dev)
            off += n_e + n_c
    words = torch.stack([g[max_len:max_len + 2] for g in gathered]).flatten()
    if moved:
        words = torch.cat([words, torch.stack([m[1] for m in moved]).to(words.dtype)])
    words = [int(x) for x in words.tolist()]  # the one read-back of this step
    if any(words[2 * r + 1] for r in range(world)):
        raise overlap
    totals = [words[2 * r] & 0xFFFFFFFF for r in range(world)]
    for k, (d, _) in enumerate(moved):
        totals[d] = (totals[d] + words[2 * world + k]) & 0xFFFFFFFF  # what owner d's range sums to once the halos are in
    report["mode"] = "owner"
    report["halo_bytes"] = out_bytes[rank]

    # ---- step 4: carry of each owner = sum of the owned ranges in front of it (mod 2^32)
    carry = 0
    for r in owners:
        if r == rank:
            break
        carry = (carry + totals[r]) & 0xFFFFFFFF
    report["owned_chunks"] = (b0, b1)

    # ---- step 5: tear down the owned chunks only, then sum the partial results
    torch_done()
    ctx.set_scan_range(b0, b1, carry)
    ctx.teardown()
    ctx.synchronize()
    allreduce_(td)
    torch_done()
    return report


def allreduce_blocks_cpu(blocks: Sequence[np.ndarray], dist, torch) -> List[np.ndarray]:
    """Element-wise sum of host uint64 / uint32 blocks over all ranks (gloo), wrap-around."""
    out = []
    for b in blocks:
        assert b.dtype in (np.uint64, np.uint32), b.dtype
        signed = b.view(np.int64 if b.dtype == np.uint64 else np.int32).copy()
        t = torch.from_numpy(signed)
        dist.all_reduce(t)
        out.append(t.numpy().view(b.dtype))
    return out


# ---- one BAM file, several GPUs (include/ngsq_bam.h "sharded device ingest") ---------------------
def open_file_shard(lib, ctx, path: str, rank: int, world: int, dist, torch, coll_device="cpu", threads: int = 2):
    """Open shard `rank` of `world` of a BAM file on the context's GPU and agree with the other ranks on
    the record boundaries: every rank inflates + indexes its BGZF block range (the shard stays resident
    in HBM), assuming the first plausible record chain; the ranks all-gather (records, assumed begin,
    found end) and a rank whose assumed begin differs from its predecessor's end re-indexes from there
    (repeated until stable: a corrected rank reports a new end).  Returns (bam handle, ffi.ShardInfo);
    batches then come from lib.ngsq_bam_next_batch_device(handle, ctx, ...), numbered from the records
    of the shards in front (first_record_index is shard-invariant).  Close with ngsq_bam_close."""
    import ctypes as C

    from . import ffi

    h = C.c_void_p()
    if lib.ngsq_bam_open(path.encode(), threads, C.byref(h)) != 0:
        raise RuntimeError(lib.ngsq_bam_last_error().decode())
    info = ffi.ShardInfo()
    if lib.ngsq_bam_shard_prepare(h, ctx, rank, world, C.byref(info)) != 0:
        raise RuntimeError(lib.ngsq_bam_last_error().decode())
    begin = 0  # keep the assumption
    for _ in range(world + 1):
        rows = _all_gather_ints([int(info.n_records), int(info.begin_voffset), int(info.end_voffset)], dist, torch,
                                coll_device)
        # shard k+1 must begin where shard k's record chain ends; shards without a record start pass it on
        want = [rows[0][1]]
        for k in range(1, world):
            want.append(rows[k - 1][2] if rows[k - 1][2] else rows[k][1])
        stable = all(want[k] == rows[k][1] for k in range(world))
        first = sum(rows[k][0] for k in range(rank))
        begin = want[rank] if want[rank] != rows[rank][1] else 0
        if lib.ngsq_bam_shard_commit(h, begin, first, C.byref(info)) != 0:
            raise RuntimeError(lib.ngsq_bam_last_error().decode())
        if stable:
            return h, info
    raise RuntimeError("shard boundaries did not settle")
