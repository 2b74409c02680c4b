"""N > 1 path on CPU (world_size 2-3, no GPU): the shard ranges; the transports of the exchange
(ngs_amd/csrc/comm.cpp: POSIX shared memory, and callbacks over torch.distributed/gloo) on host buffers;
the exchange protocol itself (ngs_amd/csrc/exchange.cpp through ngsq_exchange_state) over a numpy shard
state; and shard invariance of the facets: records split over ranks + element-wise integer
sums == the whole file, including the GC window offsets (pure function of the record's index in the
whole file)."""
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


def _make_comm(transport, rank, world, port):
    """(comm, cleanup): 'gloo' = callbacks over torch.distributed, 'shm' = the library's shared-memory transport
    (small slots so that every collective takes several rounds)."""
    from ngs_amd import shard
    if transport == "gloo":
        import torch
        import torch.distributed as dist
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        comm = shard.Comm.torch_dist(dist, torch)

        def done():
            comm.destroy()
            dist.barrier()
            dist.destroy_process_group()
        return comm, done
    if transport == "rccl-double":
        # the library's RCCL transport (device buffers, collectives on the context's stream, grouped ncclSend/ncclRecv)
        # bound to tests/rccl_double instead of librccl: three ranks on ONE GPU, which RCCL itself refuses
        os.environ["NGSQ_RCCL_LIB"] = rccl_double_path()
        boot = shard.Comm.shm(f"/ngsq-test-{port}", rank, world, slot_bytes=8192)
        uid = np.frombuffer(shard.unique_id() if rank == 0 else bytes(128), dtype=np.uint8)
        uid = boot.allgather(uid)[0].tobytes()
        comm = shard.Comm.rccl(rank, world, uid, 0)
        assert comm.kind == "rccl"

        def done():
            boot.barrier()
            comm.destroy()
            boot.destroy()
        return comm, done
    comm = shard.Comm.shm(f"/ngsq-test-{port}", rank, world, slot_bytes=8192)

    def done():
        comm.barrier()
        comm.destroy()
    return comm, done


def rccl_double_path() -> str:
    """tests/rccl_double/librccl_double.so, compiled on demand (hipcc; host code only, links the HIP runtime)."""
    import fcntl
    import subprocess
    d = os.path.join(ROOT, "tests", "rccl_double")
    src, out = os.path.join(d, "rccl_double.cpp"), os.path.join(d, "librccl_double.so")
    with open(os.path.join(d, ".build.lock"), "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
            subprocess.run(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-fPIC", "-shared", src, "-o", out + ".tmp", "-lrt"],
                           check=True, capture_output=True)
            os.replace(out + ".tmp", out)
    return out


def _run_ranks(target, world, *args):
    import multiprocessing as mp
    if "rccl-double" in args:
        rccl_double_path()   # compiled once here, not by three workers racing for the lock
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, q) + args) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in results:
        assert msg == "ok", f"rank {rank}: {msg}"


def _transport_worker(rank, world, port, q, transport):
    try:
        sys.path.insert(0, ROOT)
        comm, done = _make_comm(transport, rank, world, port)
        assert (comm.rank, comm.world) == (rank, world) and comm.kind == {"gloo": "custom", "shm": "shm", "rccl-double": "rccl"}[transport]

        def block(r, dtype, n):
            g = np.random.default_rng(100 + r)
            x = g.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32) if dtype == np.uint32 else \
                g.integers(0, 2 ** 63, n, dtype=np.uint64) * 2 + 1
            x[:10] = np.iinfo(dtype).max   # the -1 of a coverage difference array: sums wrap around
            return x
        for dtype, n in ((np.uint32, 1000), (np.uint64, 500), (np.uint32, 70_001), (np.uint64, 1)):
            got = comm.allreduce(block(rank, dtype, n))
            want = np.zeros(n, dtype=dtype)
            with np.errstate(over="ignore"):
                for r in range(world):
                    want += block(r, dtype, n)
            assert (got == want).all(), (dtype, n)
        rows = comm.allgather_ints([rank, 7 * rank + 1, 2 ** 63 + rank])
        assert rows == [[r, 7 * r + 1, 2 ** 63 + r] for r in range(world)]
        big = comm.allgather(np.arange(30_000, dtype=np.uint8) + rank)
        assert all((big[r] == (np.arange(30_000, dtype=np.uint8) + r)).all() for r in range(world))
        # grouped point to point: every rank sends two messages of different sizes to every other rank
        def msg(s, d, k):
            return (np.arange(5000 * (k + 1) + 13 * s + d, dtype=np.uint32) * (s + 1) + d + 1000 * k).astype(np.uint32)
        sends = [(d, msg(rank, d, k)) for d in range(world) if d != rank for k in range(2)]
        recvs = [(s, np.zeros_like(msg(s, rank, k))) for s in range(world) if s != rank for k in range(2)]
        comm.sendrecv(sends, recvs)
        i = 0
        for s in range(world):
            if s == rank:
                continue
            for k in range(2):
                assert (recvs[i][1] == msg(s, rank, k)).all(), (s, k)
                i += 1
        comm.sendrecv([], [])
        comm.barrier()
        done()
        q.put((rank, "ok"))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, "FAIL " + traceback.format_exc()))


@pytest.mark.parametrize("world,transport", [(2, "gloo"), (3, "gloo"), (2, "shm"), (3, "shm")])
def test_transports_on_host_buffers(world, transport):
    _run_ranks(_transport_worker, world, transport)


def _worker(rank, world, port, q):
    try:
        sys.path.insert(0, ROOT)
        from ngs_amd import ffi, host, shard
        from oracle import oracle_py
        from tests.util import random_batch

        comm, done = _make_comm("gloo", rank, world, port)

        # shard invariance: each rank scans its contiguous record range, the integer
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
        summed = [comm.allreduce(b) for b in blocks]
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
        done()
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, "FAIL " + traceback.format_exc()))


def test_world_size_2_gloo():
    _run_ranks(_worker, 2)


# ---------------------------------------------------------------------------------------------
# owner-computes coverage teardown (ngs_amd/csrc/exchange.cpp) over a numpy shard state
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


def _owner_worker(rank, world, port, q, layout, transport):
    try:
        sys.path.insert(0, ROOT)
        from ngs_amd import ffi, shard
        from tests.fake_ctx import FakeCtx

        if layout == "unsorted":
            os.environ["NGSQ_HALO_LIMIT_BYTES"] = "1024"  # force the all-reduce fallback
        comm, done = _make_comm(transport, rank, world, port)
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
        if layout.startswith("flags"):
            # streaming contexts flag the chunks they finished on their own.  "flags_ok": a chunk no other rank
            # wrote to; "flags_overlap": the first chunk this rank wrote -- the halo of the rank in front lands there
            mine.flags = np.zeros(mine.n_chunks, dtype=np.uint8)
            _, _, t_lo, t_hi = mine.depth_layout()
            if layout == "flags_ok":
                mid = (t_lo + t_hi) // 2 // 4096
                mine.flags[mid] = 1 if not mine.depth[mid * 4096:(mid + 1) * 4096].any() else 0
            elif layout == "flags_sender":
                # shards out of coordinate order at a boundary -- rank 0 has "finished while streaming" the last
                # chunk it wrote, which lies where rank 1 starts: rank 1 owns it and would tally its positions again
                if rank == 0:
                    mine.flags[(t_hi - 1) // 4096] = 1
            elif rank > 0:
                mine.flags[t_lo // 4096] = 1
            try:
                rep = comm.exchange_state(mine.shard_state())
                assert layout == "flags_ok", "an exchanged entry in a finished chunk was accepted"
            except shard.CommError as e:
                assert layout in ("flags_overlap", "flags_sender") and e.code == ffi.ERR_UNSORTED and "overlap" in str(e), e
                done()
                q.put((rank, "ok"))
                return
        if layout == "layout_mismatch":
            # a rank whose state blocks have another shape (built against another ABI, other sequences, other quality rows)
            # must be refused by every rank before anything is summed, not summed misaligned
            if rank == 1:
                mine.counters = np.zeros(mine.counters.size + 3, dtype=np.uint64)
            before = mine.counters.copy()
            try:
                comm.exchange_state(mine.shard_state())
                raise AssertionError("state blocks of different sizes were summed")
            except shard.CommError as e:
                assert e.code == ffi.ERR_STATE and "layouts differ" in str(e) and "counters" in str(e), e
            assert (mine.counters == before).all()
            done()
            q.put((rank, "ok"))
            return
        rep = comm.exchange_state(mine.shard_state()) if not layout.startswith("flags") else rep
        assert rep["mode"] == ("allreduce" if layout == "unsorted" else "owner"), rep
        assert (mine.td == whole.td).all(), "teardown results differ from the single-context scan"
        assert (mine.counters == whole.counters).all()
        if layout == "sorted" and world > 1:
            assert sum(comm.allgather_ints([rep["halo_bytes"]])[r][0] for r in range(world)) > 0
            assert mine.vaf_part == (rank, world)
        done()
        q.put((rank, "ok"))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, "FAIL " + traceback.format_exc()))


@pytest.mark.parametrize("world,layout,transport", [(2, "sorted", "gloo"), (3, "sorted", "gloo"), (3, "sorted", "shm"),
                                                    (3, "empty_rank", "gloo"), (2, "unsorted", "gloo"), (3, "unsorted", "shm"),
                                                    (3, "flags_ok", "gloo"), (3, "flags_overlap", "gloo"),
                                                    (3, "flags_overlap", "shm"), (3, "flags_sender", "shm"), (2, "flags_sender", "gloo"),
                                                    (3, "layout_mismatch", "shm"), (2, "layout_mismatch", "gloo")])
def test_owner_teardown_world_2_3(world, layout, transport):
    _run_ranks(_owner_worker, world, layout, transport)


def test_missing_rccl_is_an_error_code_not_a_crash(tmp_path):
    """a dlopen candidate that fails must fall through to the next name / NGSQ_ERR_UNSUPPORTED (the message
    carries the loader's reason), and a launch whose rank 0 cannot load RCCL agrees on the shared-memory transport
    instead of leaving the other ranks waiting for a unique id.  Fresh processes: the library is bound once per process."""
    import subprocess
    from ngs_amd import ffi
    code = (
        "import ctypes as C, sys\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "from ngs_amd import ffi\n"
        "lib = ffi.load_library()\n"
        "buf = (C.c_uint8 * ffi.COMM_ID_BYTES)()\n"
        "rc = lib.ngsq_comm_unique_id(buf)\n"
        "print(rc, (lib.ngsq_comm_last_error(None) or b'').decode())\n"
        "print(lib.ngsq_comm_rccl_version())\n")
    # a candidate that does not exist, and none of the real names resolvable either (empty search path is not possible
    # to force portably: where librccl IS installed the call succeeds through the later candidates -- also not a crash)
    env = dict(os.environ, NGSQ_RCCL_LIB=str(tmp_path / "no_such_librccl.so"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stderr)        # round 2: SIGSEGV (139)
    rc, msg = r.stdout.splitlines()[0].split(" ", 1) if " " in r.stdout.splitlines()[0] else (r.stdout.splitlines()[0], "")
    assert int(rc) in (ffi.OK, ffi.ERR_UNSUPPORTED, ffi.ERR_DEVICE)
    # a file that exists but is no library: every candidate after it is still tried
    junk = tmp_path / "librccl_junk.so"
    junk.write_bytes(b"not an ELF file")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, NGSQ_RCCL_LIB=str(junk)), timeout=120)
    assert r.returncode == 0, (r.returncode, r.stderr)
    # a library that loads but lacks the entry points (libz): unsupported, with the missing symbol named
    import ctypes.util
    z = ctypes.util.find_library("z")
    if z:
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, NGSQ_RCCL_LIB=z), timeout=120)
        assert r.returncode == 0, (r.returncode, r.stderr)
        first = r.stdout.splitlines()[0]
        assert first.startswith(str(ffi.ERR_UNSUPPORTED)) and "symbol missing: nccl" in first, first
        assert r.stdout.splitlines()[1] == "0"


def _fallback_worker(rank, world, port, q, lib_path):
    try:
        sys.path.insert(0, ROOT)
        os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_PORT=str(port), NGSQ_RCCL_LIB=lib_path)
        from ngs_amd import shard
        comm = shard.comm_from_env(0, "rccl")
        assert comm.kind == "shm" and comm.fallback_reason and "RCCL" in comm.fallback_reason, comm.fallback_reason
        got = comm.allreduce(np.full(5, rank + 1, dtype=np.uint64))
        assert list(got) == [sum(range(1, world + 1))] * 5
        comm.barrier()
        comm.destroy()
        q.put((rank, "ok"))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, "FAIL " + traceback.format_exc()))


def test_launch_without_rccl_agrees_on_shared_memory():
    import ctypes.util
    z = ctypes.util.find_library("z")
    if not z:
        pytest.skip("no libz to stand in for a librccl without entry points")
    _run_ranks(_fallback_worker, 2, z)


def _stall_worker(rank, world, port, q, lib_path):
    try:
        sys.path.insert(0, ROOT)
        os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_PORT=str(port), NGSQ_RCCL_LIB=lib_path,
                          RCCL_DOUBLE_STALL_S="120", NGSQ_RCCL_INIT_TIMEOUT_S="1.5", NGSQ_RCCL_SKIP_DEVICE_CHECK="1")
        import time
        from ngs_amd import ffi, shard
        t0 = time.time()
        comm = shard.comm_from_env(0, "rccl")
        dt = time.time() - t0
        assert comm.kind == "shm" and "did not return within" in (comm.fallback_reason or ""), comm.fallback_reason
        assert 1.0 < dt < 30.0, dt
        assert ffi.load_library().ngsq_comm_rccl_stuck() == 1
        got = comm.allreduce(np.full(3, rank + 1, dtype=np.uint64))
        assert list(got) == [sum(range(1, world + 1))] * 3
        comm.barrier()
        comm.destroy()
        q.put((rank, "ok"))
        q.close()
        q.join_thread()     # (the message is on its way before ...)
        os._exit(0)         # ... a thread of this process is still inside the stalled ncclCommInitRank
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, "FAIL " + traceback.format_exc()))
        q.close()
        q.join_thread()
        os._exit(1)


def test_rccl_init_that_never_returns_is_given_up_on():
    """ncclCommInitRank has no time limit of its own.  tests/rccl_double stalls in it for two minutes;
    ngsq_comm_create_rccl gives up after NGSQ_RCCL_INIT_TIMEOUT_S on every rank, the ranks agree on the shared-memory
    transport through the segment they met in, the exchange runs, and ngsq_comm_rccl_stuck() tells the process to leave
    with _exit().  (No GPU: NGSQ_RCCL_SKIP_DEVICE_CHECK, which only this test sets.)"""
    _run_ranks(_stall_worker, 2, rccl_double_path())
