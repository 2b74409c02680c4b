"""Harness-side binding of the multi-GPU step (include/ngsq_comm.h): one process per GPU, records
sharded by contiguous ranges (= contiguous BGZF block ranges of a sorted BAM), and ONE exchange of
integer state before the sequence-facet teardown.

The protocol itself -- counters all-reduce, ownership plan, point-to-point halos, split teardown,
all-reduce of the partial results -- is C++ (ngs_amd/csrc/exchange.cpp) over a transport
(ngs_amd/csrc/comm.cpp): RCCL called directly from the library, POSIX shared memory, or callbacks.
What is here: the ctypes wrappers, the launcher-side bootstrap (who tells whom the RCCL unique id),
and a callback transport over torch.distributed (gloo) for the CPU tests.  Nothing here computes.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import ffi

COV_CHUNK = 4096  # ngs_amd/csrc/kernels.h COV_CHUNK


class CommError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"ngsq comm error {code}: {message}")
        self.code = code
        self.message = message


def shard_range(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """(first record, record count) of `rank`: contiguous, balanced, covering [0, n_total)."""
    base, extra = divmod(n_total, world)
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


def unique_id(lib=None) -> bytes:
    """rank 0: ncclGetUniqueId (128 bytes) -- hand it to the other ranks, then Comm.rccl on every rank."""
    lib = lib or ffi.load_library()
    buf = (C.c_uint8 * ffi.COMM_ID_BYTES)()
    rc = lib.ngsq_comm_unique_id(buf)
    if rc:
        raise CommError(rc, (lib.ngsq_comm_last_error(None) or b"").decode())
    return bytes(buf)


class Comm:
    """An ngsq_comm handle.  Create with Comm.rccl / Comm.shm / Comm.torch_dist; destroy() when done."""

    def __init__(self, lib, handle, keep=None):
        self.lib, self._h, self._keep = lib, handle, keep
        self.rank, self.world = lib.ngsq_comm_rank(handle), lib.ngsq_comm_world(handle)
        self.kind = lib.ngsq_comm_kind(handle).decode()
        self.fallback_reason: Optional[str] = None   # comm_from_env: why this is not the transport asked for

    # -- constructors
    @staticmethod
    def _made(lib, rc, h, keep=None):
        if rc:
            raise CommError(rc, (lib.ngsq_comm_last_error(None) or b"").decode())
        return Comm(lib, h, keep)

    @classmethod
    def rccl(cls, rank: int, world: int, uid: bytes, device: int, lib=None) -> "Comm":
        lib = lib or ffi.load_library()
        assert len(uid) == ffi.COMM_ID_BYTES
        h = ffi.comm_p()
        buf = (C.c_uint8 * ffi.COMM_ID_BYTES).from_buffer_copy(uid)
        return cls._made(lib, lib.ngsq_comm_create_rccl(rank, world, buf, device, C.byref(h)), h)

    @classmethod
    def shm(cls, name: str, rank: int, world: int, slot_bytes: int = 0, lib=None) -> "Comm":
        lib = lib or ffi.load_library()
        h = ffi.comm_p()
        return cls._made(lib, lib.ngsq_comm_create_shm(name.encode(), rank, world, slot_bytes, C.byref(h)), h)

    @classmethod
    def torch_dist(cls, dist, torch, lib=None) -> "Comm":
        """A custom transport: the three callbacks of ngsq_comm_ops over an initialised torch.distributed
        process group with CPU tensors (gloo).  The buffers are the library's host memory, wrapped in place."""
        lib = lib or ffi.load_library()
        rank, world = dist.get_rank(), dist.get_world_size()

        def view(ptr, nbytes, dtype=np.uint8):
            a = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(int(nbytes),))
            return torch.from_numpy(a.view(dtype))

        def allreduce(_user, buf, count, eb):
            try:  # no unsigned reductions in torch: two's-complement addition is the same bit pattern
                dist.all_reduce(view(buf, count * eb, np.int64 if eb == 8 else np.int32))
                return 0
            except Exception:  # noqa: BLE001
                import traceback
                traceback.print_exc()
                return 1

        def allgather(_user, send, recv, nbytes):
            try:
                out = view(recv, nbytes * world)
                dist.all_gather([out[r * nbytes:(r + 1) * nbytes] for r in range(world)], view(send, nbytes))
                return 0
            except Exception:  # noqa: BLE001
                import traceback
                traceback.print_exc()
                return 1

        def sendrecv(_user, sends, ns, recvs, nr):
            try:
                reqs, seen = [], {}
                for i in range(nr):
                    m = recvs[i]
                    k = seen[("r", m.peer)] = seen.get(("r", m.peer), -1) + 1
                    if m.bytes:
                        reqs.append(dist.irecv(view(m.buf, m.bytes), src=m.peer, tag=k))
                for i in range(ns):
                    m = sends[i]
                    k = seen[("s", m.peer)] = seen.get(("s", m.peer), -1) + 1
                    if m.bytes:
                        reqs.append(dist.isend(view(m.buf, m.bytes), dst=m.peer, tag=k))
                for r in reqs:
                    r.wait()
                return 0
            except Exception:  # noqa: BLE001
                import traceback
                traceback.print_exc()
                return 1

        ops = ffi.CommOps()
        ops.struct_size = C.sizeof(ffi.CommOps)
        ops.allreduce_sum = ffi.ALLREDUCE_FN(allreduce)
        ops.allgather = ffi.ALLGATHER_FN(allgather)
        ops.sendrecv = ffi.SENDRECV_FN(sendrecv)
        h = ffi.comm_p()
        return cls._made(lib, lib.ngsq_comm_create_custom(rank, world, C.byref(ops), C.byref(h)), h, keep=ops)

    def destroy(self):
        if self._h:
            self.lib.ngsq_comm_destroy(self._h)
            self._h = ffi.comm_p()

    def _check(self, rc):
        if rc:
            raise CommError(rc, (self.lib.ngsq_comm_last_error(self._h) or b"").decode())

    # -- host-buffer collectives
    def barrier(self):
        self._check(self.lib.ngsq_comm_barrier(self._h))

    def allgather(self, a: np.ndarray) -> np.ndarray:
        """rows = every rank's array (same shape and dtype on all ranks)."""
        a = np.ascontiguousarray(a)
        out = np.zeros((self.world,) + a.shape, dtype=a.dtype)
        self._check(self.lib.ngsq_comm_allgather_host(self._h, a.ctypes.data, out.ctypes.data, a.nbytes))
        return out

    def allgather_ints(self, vals: Sequence[int]) -> List[List[int]]:
        return [[int(x) for x in row] for row in self.allgather(np.asarray(vals, dtype=np.uint64))]

    def allreduce(self, a: np.ndarray) -> np.ndarray:
        """Element-wise wrap-around sum of a uint32 / uint64 array over all ranks (a copy)."""
        assert a.dtype in (np.uint32, np.uint64), a.dtype
        out = np.ascontiguousarray(a).copy()
        self._check(self.lib.ngsq_comm_allreduce_host(self._h, out.ctypes.data, out.size, out.dtype.itemsize))
        return out

    def sendrecv(self, sends: Sequence[Tuple[int, np.ndarray]], recvs: Sequence[Tuple[int, np.ndarray]]):
        """Grouped point to point: sends = (peer, array), recvs = (peer, array to fill)."""
        def pack(msgs):
            arr = (ffi.P2P * max(1, len(msgs)))()
            for i, (peer, a) in enumerate(msgs):
                assert a.flags["C_CONTIGUOUS"]
                arr[i].peer, arr[i].buf, arr[i].bytes = peer, a.ctypes.data, a.nbytes
            return arr
        s, r = pack(sends), pack(recvs)
        self._check(self.lib.ngsq_comm_sendrecv_host(self._h, s, len(sends), r, len(recvs)))

    # -- the exchange
    def exchange(self, ctx) -> dict:
        """ngsq_exchange on a host.QcContext (or any object with ._ctx); call ctx.finalize() afterwards."""
        rep = ffi.ExchangeReport()
        rep.struct_size = C.sizeof(ffi.ExchangeReport)
        self._check(self.lib.ngsq_exchange(ctx._ctx, self._h, C.byref(rep)))
        return _report(rep)

    def exchange_state(self, state: "ffi.ShardState") -> dict:
        rep = ffi.ExchangeReport()
        rep.struct_size = C.sizeof(ffi.ExchangeReport)
        self._check(self.lib.ngsq_exchange_state(C.byref(state), self._h, C.byref(rep)))
        return _report(rep)

    def scan_file_shard(self, ctx, path: str, batch_records: int = 1 << 20, threads: int = 2, on_batch=None,
                        pass_mask: int = ffi.PASS_BOTH, begin_hook=None):
        """The loop of one `ngs qc --gpus N` worker: ngsq_bam_shard_open (this rank's BGZF block range of the
        file, streamed through the chunked pipeline), every batch through ngsq_process_batch, then
        ngsq_bam_shard_verify (collective) -- and, when a shard's assumed first record turns out wrong, ngsq_reset
        and one more scan of that shard.  Returns (ffi.ShardInfo, rounds of re-scanning, records this rank scanned)."""
        h = C.c_void_p()
        if self.lib.ngsq_bam_open(path.encode(), threads, C.byref(h)) != 0:
            raise RuntimeError(self.lib.ngsq_bam_last_error().decode())
        try:
            self._check(self.lib.ngsq_bam_shard_open(h, ctx._ctx, self._h))
            if begin_hook is not None:
                begin_hook(h)          # tests: plant a wrong assumption (ngsq_bam_shard_begin with another offset)
            rounds, mine, scanning = 0, 0, True
            n, err = 0, None
            while True:
                if scanning:
                    n, err = 0, None
                    while True:
                        b = ffi.Batch()
                        if self.lib.ngsq_bam_next_batch_device(h, ctx._ctx, batch_records, C.byref(b)) != 0:
                            err = self.lib.ngsq_bam_last_error().decode()
                            break
                        if b.n_records == 0:
                            break
                        if on_batch is not None:
                            on_batch(b)
                        if self.lib.ngsq_process_batch(ctx._ctx, C.byref(b), pass_mask) != 0:
                            err = (self.lib.ngsq_last_error(ctx._ctx) or b"").decode()
                            break
                        n += int(b.n_records)
                info, again = ffi.ShardInfo(), C.c_int(0)
                # (a rank that failed calls it all the same: every rank then gets an error instead of waiting -- unless the
                # scan ran from an ASSUMED first record: then the assumption may be what failed, and the shard is scanned
                # again from the offset its neighbour confirms)
                rc = self.lib.ngsq_bam_shard_verify(h, ctx._ctx, self._h, C.byref(info), C.byref(again))
                if rc != 0 and scanning and err is not None:   # (only the round that scanned reports its own failure: ADVICE r4)
                    raise RuntimeError(err)
                self._check(rc)
                if scanning and err is None:
                    mine = n
                err = None                                      # forgiven by this round's verdict
                if not again.value:
                    return info, rounds, mine
                rounds += 1
                scanning = bool(info.rescan)    # (a rank that keeps its state does not touch its reader this round)
                if scanning:
                    ctx.reset()
        finally:
            self.lib.ngsq_bam_close(h)


def _report(rep) -> dict:
    return {"mode": ffi.EXCHANGE_MODES[rep.mode], "halo_bytes": int(rep.halo_bytes_sent),
            "halo_bytes_received": int(rep.halo_bytes_received),
            "owned_chunks": (int(rep.owned_chunk_lo), int(rep.owned_chunk_hi)), "host_syncs": int(rep.host_syncs)}


def plan_owners(ranges: Sequence[Tuple[int, int]], n_chunks: int, lib=None):
    """ngsq_exchange_plan: from every rank's written chunk range -> (own[r] = (b0, b1), owners by position,
    xfer[(src, dst)] = (c0, c1))."""
    lib = lib or ffi.load_library()
    world = len(ranges)
    r = np.asarray(ranges, dtype=np.uint64).reshape(-1)
    own = np.zeros(2 * world, dtype=np.uint64)
    order = np.zeros(world, dtype=np.uint32)
    n_own = C.c_uint32()
    cap = world * world
    xf = np.zeros(4 * cap, dtype=np.uint64)
    n = lib.ngsq_exchange_plan(r.ctypes.data_as(ffi.u64p), world, n_chunks, own.ctypes.data_as(ffi.u64p),
                               order.ctypes.data_as(ffi.u32p), C.byref(n_own), xf.ctypes.data_as(ffi.u64p), cap)
    if n < 0:
        raise CommError(int(n), (lib.ngsq_comm_last_error(None) or b"").decode())
    xfer = {(int(xf[4 * k]), int(xf[4 * k + 1])): (int(xf[4 * k + 2]), int(xf[4 * k + 3])) for k in range(n)}
    return ([(int(own[2 * k]), int(own[2 * k + 1])) for k in range(world)], [int(x) for x in order[:n_own.value]], xfer)


# ---- launcher-side bootstrap -----------------------------------------------------------------------
def _launcher_tag() -> str:
    """A name every rank of one launch derives alike and no other launch can: the rendezvous port, the
    launcher's pid and the launcher's start time (clock ticks since boot, /proc/<pid>/stat field 22)."""
    ppid = os.getppid()
    try:
        with open(f"/proc/{ppid}/stat") as f:
            start = f.read().rsplit(")", 1)[1].split()[19]
    except OSError:
        start = "0"
    return f"{os.environ.get('MASTER_PORT', '0')}-{ppid}-{start}"


def comm_from_env(device: int, kind: str = "rccl", lib=None) -> "Comm":
    """The communicator of one rank of a torchrun-style launch on ONE node (RANK / WORLD_SIZE / MASTER_PORT
    in the environment).  The ranks first meet in a shared-memory segment named after the launch; for kind
    'rccl' rank 0's ncclGetUniqueId travels through it and every rank calls ncclCommInitRank on its device.
    No torch, no process group."""
    lib = lib or ffi.load_library()
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    tag = _launcher_tag()
    boot = Comm.shm(f"/ngsq-{tag}", rank, world, slot_bytes=1 << 16, lib=lib)
    why = None
    comm = None
    if kind == "rccl":
        uid = bytes(ffi.COMM_ID_BYTES)
        if rank == 0:
            try:
                uid = unique_id(lib)
            except CommError as e:  # librccl could not be loaded: the others must hear of it, not wait for an id
                why = str(e)
        msg = np.frombuffer(uid + bytes([0 if why else 1]), dtype=np.uint8)
        got = boot.allgather(msg)[0].tobytes()
        if not got[ffi.COMM_ID_BYTES]:
            why = why or "rank 0 could not load RCCL"
        else:
            try:
                comm = Comm.rccl(rank, world, got[:ffi.COMM_ID_BYTES], device, lib)
            except CommError as e:      # e.g. two ranks on one GPU: RCCL refuses that
                why = str(e)
        ok = boot.allgather_ints([0 if why else 1])
        if all(int(v[0]) for v in ok):
            boot.barrier()
            boot.destroy()
            return comm
        # the ranks agree: not every one of them has an RCCL communicator.  The exchange still runs natively,
        # host-staged through shared memory; whoever reports numbers must say so (Comm.fallback_reason).
        if comm is not None:
            comm.destroy()
        bad = [r for r, v in enumerate(ok) if not int(v[0])]
        why = why or f"ncclCommInitRank failed on rank(s) {bad}"
    big = Comm.shm(f"/ngsq-{tag}-x", rank, world, slot_bytes=32 << 20, lib=lib)
    big.fallback_reason = why
    boot.barrier()
    boot.destroy()
    return big
