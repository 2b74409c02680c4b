"""N > 1 path on CPU (gloo, world_size 2): the shard ranges and the state exchange of
ngs_amd/shard.py, and shard invariance of the facets: records split
over ranks + element-wise integer sums == the whole file, including the GC window
offsets (pure function of the record's index in the whole file)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_ranges_partition_the_file():
    from ngs_amd.shard import shard_range
    for n in (0, 1, 7, 1000, 10 ** 9 + 7):
        for world in (1, 2, 3, 8):
            pos = 0
            for r in range(world):
                first, cnt = shard_range(n, r, world)
                assert first == pos and cnt >= 0
                pos += cnt
            assert pos == n
            counts = [shard_range(n, r, world)[1] for r in range(world)]
            assert max(counts) - min(counts) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    try:
        sys.path.insert(0, ROOT)
        import torch
        import torch.distributed as dist
        from ngs_amd import ffi, host, shard
        from oracle import oracle_py
        from tests.util import random_batch

        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)

        # (1) wrap-around sums of uint32 / uint64 blocks through signed all_reduce
        rng = np.random.default_rng(100 + rank)
        b32 = rng.integers(0, 2 ** 32, 1000, dtype=np.uint64).astype(np.uint32)
        b32[:10] = 0xFFFFFFFF  # the -1 of a coverage difference array
        b64 = rng.integers(0, 2 ** 63, 500, dtype=np.uint64) * 2 + 1
        got32, got64 = shard.allreduce_blocks_cpu([b32, b64], dist, torch)
        want32, want64 = np.zeros_like(b32), np.zeros_like(b64)
        for r in range(world):
            g = np.random.default_rng(100 + r)
            x = g.integers(0, 2 ** 32, 1000, dtype=np.uint64).astype(np.uint32)
            x[:10] = 0xFFFFFFFF
            want32 += x
            want64 += g.integers(0, 2 ** 63, 500, dtype=np.uint64) * 2 + 1
        assert (got32 == want32).all() and (got64 == want64).all()

        # (2) shard invariance: each rank scans its contiguous record range, the integer
        # results are summed, and must equal the scan of the whole file
        ref_len = [30_000, 4_000]
        whole = random_batch(np.random.default_rng(7), 6001, ref_len, weird=True)
        first, cnt = shard.shard_range(whole.n, rank, world)
        mine = whole.slice(first, first + cnt)
        assert mine.first_record_index == first
        kw = dict(facets=ffi.FACETS_DEFAULT & ffi.FACETS_RECORD_BASED, max_read_len=320, gc_seed=11)
        o = oracle_py.Oracle(ref_len, **kw)
        o.process_batch(mine)
        o.finalize(allow_malformed=True)
        g = o.general()
        blocks = [np.array([g[k] for k in ffi.GENERAL_FIELDS] + g["read_one_cigar_ops"] + g["read_two_cigar_ops"],
                           dtype=np.uint64)]
        h, p, i = o.template_length()
        blocks.append(np.concatenate([h, np.array([p, i], dtype=np.uint64)]))
        gc = o.gc_content()
        blocks.append(np.concatenate([gc["histogram"], np.array(
            [gc[k] for k in ("total_gc_count", "total_at_count", "total_other_count", "processed",
                             "ignored_flags", "ignored_too_short")], dtype=np.uint64)]))
        blocks.append(o.quality_scores().reshape(-1))
        summed = shard.allreduce_blocks_cpu(blocks, dist, torch)
        if rank == 0:
            w = oracle_py.Oracle(ref_len, **kw)
            w.process_batch(whole)
            w.finalize(allow_malformed=True)
            gw = w.general()
            want = [np.array([gw[k] for k in ffi.GENERAL_FIELDS] + gw["read_one_cigar_ops"] +
                             gw["read_two_cigar_ops"], dtype=np.uint64)]
            h, p, i = w.template_length()
            want.append(np.concatenate([h, np.array([p, i], dtype=np.uint64)]))
            gcw = w.gc_content()
            want.append(np.concatenate([gcw["histogram"], np.array(
                [gcw[k] for k in ("total_gc_count", "total_at_count", "total_other_count", "processed",
                                  "ignored_flags", "ignored_too_short")], dtype=np.uint64)]))
            want.append(w.quality_scores().reshape(-1))
            for a, b in zip(summed, want):
                assert (a == b).all()
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, "FAIL " + traceback.format_exc()))


def test_world_size_2_gloo():
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in results:
        assert msg == "ok", f"rank {rank}: {msg}"


# ---------------------------------------------------------------------------------------------
# owner-computes coverage teardown (ngs_amd/shard.py owner_teardown) with a numpy context
# ---------------------------------------------------------------------------------------------
def test_plan_owners_is_a_disjoint_cover():
    from ngs_amd.shard import plan_owners
    rng = np.random.default_rng(3)
    for _ in range(200):
        world, n_chunks = int(rng.integers(1, 9)), int(rng.integers(1, 500))
        ranges = []
        for r in range(world):
            if rng.random() < 0.2:
                ranges.append((5, 5))
            else:
                a = int(rng.integers(0, n_chunks))
                ranges.append((a, int(rng.integers(a + 1, n_chunks + 1))))
        own, owners, xfer = plan_owners(ranges, n_chunks)
        covered = np.zeros(n_chunks, dtype=int)
        for r in owners:
            covered[own[r][0]:own[r][1]] += 1
        if owners:
            assert (covered == 1).all()
        # every written chunk of every rank reaches exactly one owner (itself or via a transfer)
        for s, (lo, hi) in enumerate(ranges):
            reach = np.zeros(n_chunks, dtype=int)
            if hi > lo:
                reach[max(lo, own[s][0]):min(hi, own[s][1])] += 1
            for (a, d), (c0, c1) in xfer.items():
                if a == s:
                    assert own[d][0] <= c0 and c1 <= own[d][1]
                    reach[c0:c1] += 1
            assert (reach[lo:hi] == 1).all() and reach.sum() == max(0, hi - lo)


def _owner_worker(rank, world, port, q, layout):
    try:
        sys.path.insert(0, ROOT)
        import torch
        import torch.distributed as dist
        from ngs_amd import shard
        from tests.fake_ctx import FakeCtx

        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        ref_len = [30_000, 9_000, 12_345]
        rng = np.random.default_rng(5)
        n = 4000
        refs = np.sort(rng.integers(0, 3, n)) if layout != "unsorted" else rng.integers(0, 3, n)
        starts = np.array([rng.integers(1, ref_len[r] - 400) for r in refs])
        if layout != "unsorted":
            order = np.lexsort((starts, refs))
            refs, starts = refs[order], starts[order]
        ends = starts + rng.integers(0, 300, n)
        ends[::97] += 5000  # a few long skips crossing shard boundaries
        ends = np.minimum(ends, np.array([ref_len[r] for r in refs]))

        def fill(ctx, lo, hi):
            for r in range(3):
                m = refs[lo:hi] == r
                ctx.add_reads(r, starts[lo:hi][m], ends[lo:hi][m])

        whole = FakeCtx(ref_len)
        fill(whole, 0, n)
        whole.teardown()
        if layout == "empty_rank" and rank == 1:
            first, cnt = 0, 0
        elif layout == "empty_rank":
            first, cnt = shard.shard_range(n, 0 if rank == 0 else rank - 1, world - 1)
        else:
            first, cnt = shard.shard_range(n, rank, world)
        mine = FakeCtx(ref_len)
        fill(mine, first, first + cnt)
        if layout == "unsorted":
            shard.HALO_LIMIT_BYTES = 1 << 10  # force the all-reduce fallback
        if layout.startswith("flags"):
            # streaming contexts flag the chunks they finished on their own.  "flags_ok": a chunk no other rank
            # wrote to; "flags_overlap": the first chunk this rank wrote -- the halo of the rank in front lands there
            mine.flags = np.zeros(mine.n_chunks, dtype=np.uint8)
            _, _, t_lo, t_hi = mine.depth_layout()
            if layout == "flags_ok":
                mid = (t_lo + t_hi) // 2 // 4096
                mine.flags[mid] = 1 if not mine.depth[mid * 4096:(mid + 1) * 4096].any() else 0
            elif rank > 0:
                mine.flags[t_lo // 4096] = 1
            try:
                rep = shard.owner_teardown(mine, dist, torch, mine.views())
                assert layout == "flags_ok", "an exchanged entry in a finished chunk was accepted"
            except RuntimeError as e:
                assert layout == "flags_overlap" and "overlap" in str(e), e
                dist.barrier()
                dist.destroy_process_group()
                q.put((rank, "ok"))
                return
        rep = shard.owner_teardown(mine, dist, torch, mine.views()) if not layout.startswith("flags") else rep
        assert rep["mode"] == ("allreduce" if layout == "unsorted" else "owner"), rep
        assert (mine.td == whole.td).all(), "teardown results differ from the single-context scan"
        assert (mine.counters == whole.counters).all()
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, "FAIL " + traceback.format_exc()))


@pytest.mark.parametrize("world,layout", [(2, "sorted"), (3, "sorted"), (3, "empty_rank"), (2, "unsorted"),
                                          (3, "flags_ok"), (3, "flags_overlap")])
def test_owner_teardown_gloo(world, layout):
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_owner_worker, args=(r, world, port, q, layout)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in results:
        assert msg == "ok", f"rank {rank}: {msg}"
