#!/usr/bin/env python3
"""bench.py -- `ngs qc` record-scanning hot path on N MI355X GPUs of one node.

    python bench.py --gpus N --steps K --warmup W          (launches its own N ranks, one per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W     (same ranks, launched by torchrun)

Workload (BASELINE.json configs[2], the configuration the metric is quoted on
that fits one GPU): per GPU 100 M synthetic 150 bp reads resident in HBM as SoA
columns, coordinate-sorted over a chr1-sized reference (L = 248 956 422), ALL
default QC facets (General, Template Length, GC Content, Quality Score,
Coverage).  One step = one full pass of the hot path over the resident shard:
reset -> the facet kernels over every record -> (N > 1: ngsq_exchange -- RCCL
called from the library: counters all-reduce, point-to-point coverage halos,
teardown split N ways, all-reduce of the partial results) -> integer results on
the host.

Prints ONE JSON line on rank 0 (contract in the task statement) with the
`roofline` of the dominant kernel (Quality Score: 150 of the 254 algorithmic
bytes per record), the `cpu_baseline` (the C oracle, 1 core, bounded sample; an
all-cores figure beside it), and -- at N = 1 -- the two other rates the metric
is about: `h2d_inclusive` (pinned host SoA batches through ngsq_process_batch)
and `file_end_to_end` (a synthetic BGZF BAM through `ngs qc` to the JSON), with
`ingest_roofline` for the device inflate kernel that bounds the latter.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CHR1 = 248_956_422
CHR2 = 242_193_529
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
GC_SEED = 0x4E4753


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--repeats", type=int, default=3, help="timed loops of --steps steps each; `value` is the median loop")
    ap.add_argument("--records", type=int, default=100_000_000, help="records per GPU")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--workload", choices=["fixed", "mixed"], default="fixed")
    ap.add_argument("--mixed-max-len", type=int, default=300, help="longest read of --workload mixed (50..N bp)")
    ap.add_argument("--cpu-sample", type=int, default=10_000_000,
                    help="records of the workload timed on the CPU oracle (0 = skip)")
    ap.add_argument("--no-timing", action="store_true", help="no per-kernel HIP event brackets")
    ap.add_argument("--force-dist", action="store_true",
                    help="run ngsq_exchange over RCCL even with one rank (plumbing check)")
    ap.add_argument("--coverage", choices=["auto", "stream", "array"], default="auto",
                    help="stream: sorted_input context, Coverage finishes positions while the sorted records stream by; "
                         "array: difference arrays + teardown scan (any record order); auto: stream up to 0.5 records per "
                         "reference position in the whole file (whole-genome depths: measured faster there), array above "
                         "(the weak-scaling runs pile N x 100 M reads on chr1: DESIGN.md section 5.4)")
    ap.add_argument("--transport", choices=["rccl", "shm"], default="rccl",
                    help="transport of the N > 1 exchange: rccl = RCCL over xGMI called from the library; shm = its "
                         "shared-memory host transport (test boxes: ranks sharing one GPU)")
    ap.add_argument("--same-gpu", action="store_true", help="all ranks on GPU 0 (boxes with one GPU; needs --transport shm)")
    ap.add_argument("--emulate-shard", default="",
                    help="R/W: on ONE GPU, scan the records shard R of a W-GPU run would scan (the W x --records file's "
                         "slice, W times the depth, head guard as for rank R) -- kernel cost of a shard without the exchange")
    ap.add_argument("--facets", type=lambda x: int(x, 0), default=0x1F,
                    help="facet mask (default 0x1F = all default facets; 0x0E = BASELINE configs[1]; 0x20 adds Edits with "
                         "a synthetic reference resident in HBM, 0x40 Genomic Features with a synthetic gene model)")
    ap.add_argument("--file-records", type=int, default=60_000_000,
                    help="records PER GPU of the synthetic BAM of the file_end_to_end leg (0 = skip that leg); at N > 1 one file "
                         "of N x this many records is scanned by `ngs qc --gpus N` (fewer when writing it would take longer "
                         "than --file-write-budget seconds: the writer is zlib on the host cores)")
    ap.add_argument("--file-write-budget", type=float, default=450.0,
                    help="seconds the N > 1 file leg may spend writing its BAM (8 x 60 M records take zlib level 6 on 16 host cores "
                         "~390 s; the driver allows a run 1800 s)")
    ap.add_argument("--file-big-records", type=int, default=0,
                    help="N = 1: a second, larger file scanned in process beside the --file-records one (0 = skip; 200 M records "
                         "take the zlib writer ~160 s)")
    ap.add_argument("--file-realistic-records", type=int, default=150_000_000,
                    help="N = 1: records of the aligner-style file (Illumina names, NM/MD/MC/AS/XS/MQ/RG/SA/XA/B tags, 15 %% multi-"
                         "operation CIGARs, real mate positions) scanned beside the plain one (0 = skip)")
    ap.add_argument("--file-realistic-budget", type=float, default=170.0, help="seconds the aligner-style file may take to write")
    ap.add_argument("--mixed-records", type=int, default=100_000_000,
                    help="N = 1: records of the 50-300 bp mixed-CIGAR workload timed beside the headline one (0 = skip)")
    ap.add_argument("--mixed-steps", type=int, default=30)
    ap.add_argument("--file-level", type=int, default=6, help="zlib level of that BAM")
    ap.add_argument("--h2d-batch", type=int, default=4_000_000, help="records per host batch of the h2d_inclusive leg (0 = skip)")
    ap.add_argument("--extra-facet-records", type=int, default=100_000_000, help="records of the Edits / Genomic Features leg")
    ap.add_argument("--all-facets-records", type=int, default=100_000_000,
                    help="N = 1: records of the all-seven-facets pass (Edits with the reference in HBM, Genomic Features with a gene model) "
                         "timed beside the headline one (0 = skip)")
    ap.add_argument("--all-facets-steps", type=int, default=15)
    ap.add_argument("--whole-genome-records", type=int, default=100_000_000,
                    help="N = 1: records of the whole_genome leg -- the 195-sequence GRCh38 header at full length, records on all of it (0 = skip)")
    ap.add_argument("--whole-genome-steps", type=int, default=7)
    ap.add_argument("--live-traffic", type=int, default=1,
                    help="1 (N = 1, default workload): two short passes of this script under rocprofv3 --pmc (FETCH_SIZE, WRITE_SIZE -- "
                         "separate passes, child processes started before this one touches HIP) give roofline.traffic of THIS run; "
                         "0, or any failure: the committed profiles/*_traffic.json")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--pmc-inflate-child", default="", help=argparse.SUPPRESS)   # path of the BAM this child scans once under rocprofv3 --pmc
    ap.add_argument("--extra-facet-legs", type=int, default=1,
                    help="1: also time the Edits and Genomic Features kernels on a 10 M-record slice (N = 1 only)")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` starts its own ranks.  Nothing in this function touches HIP.
# ---------------------------------------------------------------------------------------------
def launch_ranks(n: int) -> int:
    import socket
    from ngs_amd import build
    build.build(verbose=False)  # once, before the ranks start
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    alive = set(range(n))
    while alive:
        for r in sorted(alive):
            code = procs[r].poll()
            if code is None:
                continue
            alive.discard(r)
            if code != 0 and rc == 0:
                rc = code
                print(f"bench.py: rank {r} exited with {code}; stopping the other ranks", file=sys.stderr)
                for o in alive:
                    procs[o].terminate()  # exactly the processes started above
        time.sleep(0.05)
    return rc


def build_once():
    """Every rank may be the first to arrive on a fresh checkout: serialise the (usually no-op) build."""
    import fcntl
    from ngs_amd import build
    with open(os.path.join(ROOT, "ngs_amd", ".build.lock"), "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        build.build(verbose=False)



# ---------------------------------------------------------------------------------------------
# what the GPU ran at while the timed loops ran: sysfs of the amdgpu driver, no HIP, no child process
# ---------------------------------------------------------------------------------------------
def _sysfs_cards():
    import glob
    return sorted(os.path.dirname(p) for p in glob.glob("/sys/class/drm/card[0-9]*/device/pp_dpm_sclk"))


def _read(path):
    try:
        with open(path) as f:
            return f.read()
    except OSError:
        return None


def _dpm_current(text):
    """pp_dpm_sclk / pp_dpm_mclk: one line per level, the current one ends with '*' ('1: 2400Mhz *')."""
    if not text:
        return None
    import re
    for ln in text.splitlines():
        if ln.rstrip().endswith("*"):
            m = re.search(r"(\d+)\s*[Mm][Hh]z", ln)
            if m:
                return int(m.group(1))
    return None


def gpu_state_once(card):
    import glob
    out = {"sclk_mhz": _dpm_current(_read(os.path.join(card, "pp_dpm_sclk"))), "mclk_mhz": _dpm_current(_read(os.path.join(card, "pp_dpm_mclk")))}
    for hw in glob.glob(os.path.join(card, "hwmon", "hwmon*")):
        for key, name, scale in (("power_w", "power1_average", 1e-6), ("power_w", "power1_input", 1e-6), ("power_cap_w", "power1_cap", 1e-6),
                                 ("temp_edge_c", "temp1_input", 1e-3), ("temp_junction_c", "temp2_input", 1e-3), ("temp_mem_c", "temp3_input", 1e-3)):
            v = _read(os.path.join(hw, name))
            if v and key not in out:
                try:
                    out[key] = round(int(v.split()[0]) * scale, 1)
                except ValueError:
                    pass
    b = _read(os.path.join(card, "gpu_busy_percent"))
    if b:
        try:
            out["busy_pct"] = int(b.split()[0])
        except ValueError:
            pass
    return out


class ClockSampler:
    """Samples the card's clocks, power and temperature every 50 ms on a thread of its own while the timed loops run (the loops
    spend their time inside ctypes calls, which release the interpreter lock)."""

    def __init__(self, index, pci=""):
        import threading
        cards = _sysfs_cards()
        # the card whose PCI address is the HIP device's (a host shows every card of the node, a container's HIP only its own)
        mine = [c for c in cards if pci and os.path.realpath(c).lower().endswith(pci.lower())]
        self.how = "matched by PCI address " + pci if mine else "NOT matched to the HIP device (no PCI address match): card by ordinal"
        self.card = mine[0] if mine else (cards[index] if index < len(cards) else (cards[0] if cards else None))
        self.levels = {k: (_read(os.path.join(self.card, k)) or "").split("\n") for k in ("pp_dpm_sclk", "pp_dpm_mclk")} if self.card else {}
        self.samples, self._stop = [], threading.Event()
        self._t = threading.Thread(target=self._run, daemon=True) if self.card else None

    def _run(self):
        while not self._stop.is_set():
            self.samples.append(gpu_state_once(self.card))
            self._stop.wait(0.05)

    def __enter__(self):
        if self._t:
            self._t.start()
        return self

    def __exit__(self, *a):
        self._stop.set()
        if self._t:
            self._t.join(timeout=1.0)

    def summary(self):
        if not self.card:
            return {"source": "unavailable: no /sys/class/drm/card*/device/pp_dpm_sclk on this machine"}
        out = {"source": self.card + " (amdgpu sysfs, sampled every 50 ms during the timed loops; " + self.how + ")", "samples": len(self.samples),
               "dpm_levels_before_the_loops": {k: [x.strip() for x in v if x.strip()] for k, v in self.levels.items()}}
        for key in ("sclk_mhz", "mclk_mhz", "power_w", "power_cap_w", "temp_edge_c", "temp_junction_c", "temp_mem_c", "busy_pct"):
            vals = sorted(v[key] for v in self.samples if v.get(key) is not None)
            if vals:
                out[key] = {"min": vals[0], "median": vals[len(vals) // 2], "max": vals[-1]}
        return out


def settle_page_cache(path: str) -> list:
    """Two plain parallel pread() passes over a freshly written file.  A page of the page cache moves from the inactive to the
    active list on its SECOND access, and that pass is slow whoever makes it (measured, round 5: plain preads of a fresh
    3.8 GB file 0.043 s, again 0.364 s, then 0.04 s for good; the second of three scans of a fresh file was the 70 % outlier
    VERDICT r4 found -- with the block cache four times as large just the same).  The timed scans then all see the same file."""
    return [round(os.path.getsize(path) / 1e9 / max(_timed(lambda: raw_read_rate(path)), 1e-9), 1) for _ in range(2)]


def _timed(fn):
    t0 = time.perf_counter()
    fn()
    return time.perf_counter() - t0


def median(xs):
    v = sorted(xs)
    return v[len(v) // 2] if len(v) % 2 else 0.5 * (v[len(v) // 2 - 1] + v[len(v) // 2])


def main() -> int:
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args.gpus)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # RCCL prints a version banner on stdout through C stdio -- buffered, so it comes out when a rank exits, behind the
    # JSON line.  The line has stdout to itself: file descriptor 1 of every rank (and of what it starts) is stderr from
    # here on, the line is written to a copy of the original descriptor.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    build_once()   # (touches no HIP; before the profiled child passes, so that they only ever LOAD a built library: ADVICE r4)
    if args.pmc_inflate_child:
        return pmc_inflate_child(args, json_fd)
    live = (None, None)
    args.live_inflate = None
    if world == 1 and args.live_traffic and not args.pmc_child and args.workload == "fixed" and not args.emulate_shard and args.file_records > 0:
        args.live_inflate = live_inflate_traffic(args)   # (child processes; nothing in this process has touched HIP yet)
    if (world == 1 and args.live_traffic and not args.pmc_child and args.workload == "fixed" and not args.emulate_shard and not args.force_dist
            and (args.facets & 0x08)):
        live = live_pmc_traffic(args)   # (child processes; nothing in this process has touched HIP yet)
        if args.mixed_records > 0 and args.facets == 0x1F and live[0] is not None:   # (not after a failure: at most one time limit is spent)
            args.live_mixed = live_pmc_traffic(args, mixed=True)

    import numpy as np
    from ngs_amd import ffi, host, shard

    lib = ffi.load_library()
    if lib.ngsq_device_count() < 1:
        print("bench.py: no HIP device visible; the hot path has no CPU fallback", file=sys.stderr)
        return 3
    device = 0 if args.same_gpu else local_rank
    use_dist = world > 1 or args.force_dist
    comm = None
    if use_dist:
        if world > 1:
            comm = shard.comm_from_env(device, args.transport, lib)
        else:
            comm = shard.Comm.rccl(0, 1, shard.unique_id(lib), device, lib)

    n = args.records
    mixed = args.workload == "mixed"
    max_len = args.mixed_max_len if mixed else args.read_len
    # the whole synthetic file has world * n records; this rank owns the contiguous
    # record range [rank*n, (rank+1)*n) = a contiguous BGZF block range of a sorted BAM
    # (weak scaling: ngs_amd.shard.shard_range(n * world, rank, world) == (rank * n, n))
    emu_rank, emu_world = (int(x) for x in args.emulate_shard.split("/")) if args.emulate_shard else (rank, world)
    scfg = host.synth_config(n * emu_world, mode=ffi.SYNTH_MIXED if mixed else ffi.SYNTH_FIXED,
                             read_len=args.read_len, max_len=args.mixed_max_len, ref_len=CHR1, n_refs=2,
                             # with Edits in the mask the reads are sampled from the reference they are compared with
                             seq_model=ffi.SYNTH_SEQ_FROM_REFERENCE if args.facets & ffi.FACET_EDITS else ffi.SYNTH_SEQ_IID)
    if args.coverage == "auto":
        args.coverage = "stream" if emu_world * n / CHR1 <= 0.5 else "array"
    ref_bases = [host.synth_reference(scfg, r, L, lib) for r, L in enumerate((CHR1, CHR2))] if args.facets & ffi.FACET_EDITS else None
    ctx = host.QcContext([CHR1, CHR2], [1, 1], facets=args.facets, device=device,
                         max_read_len=max_len, gc_seed=GC_SEED, timing=not args.no_timing,
                         sorted_input=args.coverage == "stream", ref_bases=ref_bases,
                         # shards behind the first: positions a read of the shard in front may still cover
                         # (the synthetic reads span at most 5.3 kb; real files: `ngs qc --gpus` keeps 1 Mi)
                         cov_head_guard=(1 << 16) if emu_rank > 0 else 0, lib=lib)
    if args.facets & ffi.FACET_FEATURES:
        ctx.set_features(*synthetic_gene_model(np))
    t_gen = time.perf_counter()
    db = ctx.synth_device_batch(scfg, emu_rank * n, n)
    t_gen = time.perf_counter() - t_gen

    def sync():
        ctx.synchronize()
        if comm is not None:
            comm.barrier()

    report = {}

    def step():
        ctx.reset()
        ctx.process_batch(db)
        if comm is not None:
            report.update(comm.exchange(ctx))   # include/ngsq_comm.h: the one exchange of a sharded scan
        ctx.finalize()

    for _ in range(args.warmup):
        step()
    sync()
    ctx.kernel_timing_reset()
    # the timed region: `repeats` loops of EXACTLY --steps steps, each bracketed by barrier + synchronize; `value` is the median
    # loop (all of them are in the line, with the clocks the card ran at), so that a slow box shows as one
    loops = []
    pci_buf = __import__("ctypes").create_string_buffer(64)
    pci = pci_buf.value.decode() if lib.ngsq_device_pci_bus_id(device, pci_buf, 64) > 0 else ""
    pci = pci_buf.value.decode()
    with ClockSampler(device, pci) as clocks:
        for _ in range(max(1, args.repeats)):
            sync()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step()
            sync()
            dt = time.perf_counter() - t0
            if comm is not None:
                dt = float(comm.allgather(np.array([dt], dtype=np.float64)).max())
            loops.append(dt)
    elapsed = median(loops)
    total_records = n * world
    parity = check_invariants(ctx, ffi, total_records, args, mixed, emu_world > world)
    timing = ctx.kernel_timing()
    # N = 1: the OTHER Coverage path on the same resident records, so that the 1 -> 2 step of the scaling curve (which
    # switches from streaming to the difference arrays: DESIGN.md section 5.4) can be read as communication only
    coverage_paths = None
    if world == 1 and not args.emulate_shard and (args.facets & ffi.FACET_COVERAGE) and not args.force_dist and not args.pmc_child:
        other = "array" if args.coverage == "stream" else "stream"
        coverage_paths = {args.coverage: round(elapsed / args.steps * 1e3, 3)}
        try:
            ctx2 = host.QcContext([CHR1, CHR2], [1, 1], facets=args.facets, device=device, max_read_len=max_len, gc_seed=GC_SEED,
                                  timing=False, sorted_input=other == "stream", ref_bases=ref_bases, lib=lib)
            if args.facets & ffi.FACET_FEATURES:
                ctx2.set_features(*synthetic_gene_model(np))
            k2 = max(5, min(args.steps, 40))
            for _ in range(3):
                ctx2.reset(); ctx2.process_batch(db); ctx2.finalize()
            ctx2.synchronize()
            t2 = time.perf_counter()
            for _ in range(k2):
                ctx2.reset(); ctx2.process_batch(db); ctx2.finalize()
            ctx2.synchronize()
            coverage_paths[other] = round((time.perf_counter() - t2) / k2 * 1e3, 3)
            coverage_paths["same_integers"] = bool((ctx2.coverage_sequence(0)[1] == ctx.coverage_sequence(0)[1]).all())
            ctx2.close()
        except Exception as e:  # noqa: BLE001 -- reported, never required
            coverage_paths["failed"] = f"{type(e).__name__}: {e}"

    rc = 0 if parity.startswith("ok") else 1
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = total_records * args.steps / elapsed
        n_ops = (db.cigar_ops / n) if n else 1.0
        algo_rec = (25.0 + 4.0 * n_ops + db.seq_bytes / n + db.qual_bytes / n) if n else 0.0
        # dominant kernel: Quality Score -- algorithmic bytes = the QUAL bytes it must read
        q = timing.get("qual", {"launches": 0, "total_ms": 0.0, "algo_bytes": 0})
        roofline = None
        if q["launches"] and q["total_ms"] > 0:
            avg_ms = q["total_ms"] / q["launches"]
            achieved = (q["algo_bytes"] / q["launches"]) / (avg_ms * 1e-3) / 1e9
            roofline = {"bound": "hbm", "kernel": "k_qual_ragged" if mixed else "k_qual_perm", "achieved": round(achieved, 2),
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                        "avg_launch_ms": round(avg_ms, 4),
                        "algo_bytes_per_launch": q["algo_bytes"] // q["launches"]}
            if live[0] is not None and not mixed:
                roofline["traffic"], roofline["traffic_source"] = live
                roofline["traffic_measured_in_this_run"] = True   # by this command: separate rocprofv3 --pmc passes of the same workload
            else:
                roofline["traffic"], roofline["traffic_source"] = pmc_traffic(n, args)
                roofline["traffic_measured_in_this_run"] = False  # the committed PMC summary (tools/profile_round.sh)
                if live[1]:
                    roofline["live_traffic_unavailable"] = live[1]
        kernels = kernel_table(timing)
        out = {
            "metric": "BAM records/sec (whole node), all qc facets, 150 bp reads",
            "value": round(value, 1), "unit": "records/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "ms_per_step_each_loop": [round(t / args.steps * 1e3, 3) for t in loops],
            "loops": "%d timed loops of %d steps each; value and ms_per_step are the median loop's" % (len(loops), args.steps),
            "gpu_state": clocks.summary(), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8/u32 integer (u64 accumulators)",
            "data": "synthetic",
            "config": {"workload": ("BASELINE configs[2]: %d M synthetic %s reads per GPU resident in HBM, "
                                    "all default facets incl. CIGAR coverage over chr1 (L=248956422)"
                                    % (n // 1_000_000, "50-300 bp mixed-CIGAR" if mixed else f"{args.read_len} bp")),
                       "records_per_gpu": n, "read_len": max_len if mixed else args.read_len,
                       "facets": ",".join(nm for b_, nm in ((1, "General"), (2, "Template Length"), (4, "GC Content"),
                                                            (8, "Quality Score"), (16, "Coverage"), (32, "Edits"),
                                                            (64, "Genomic Features")) if args.facets & b_),
                       "sharding": ("contiguous record (BGZF block) ranges; ngsq_exchange over %s: all-reduce of counters, "
                                    "owner-computes coverage teardown with point-to-point halos (mode %s, %d halo bytes from "
                                    "rank 0, %d host syncs per step)"
                                    % (comm.kind + (" (RCCL asked for, not available: %s)" % comm.fallback_reason
                                                    if comm.fallback_reason else ""), report.get("mode"), report.get("halo_bytes", 0), report.get("host_syncs", 0)))
                       if comm is not None else "single GPU",
                       "coverage": ("streamed from the coordinate-sorted records (sorted_input)" if args.coverage == "stream"
                                    else "difference arrays + teardown scan"),
                       "algorithmic_bytes_per_record": round(algo_rec, 2),
                       "hbm_frac_whole_pass": round(value / world * algo_rec / (HBM_PEAK_GBS * 1e9), 4)},
            "roofline": roofline, "cpu_baseline": None, "kernels": kernels,
            "ms_per_step_outside_kernels": round(ms_per_step - sum(v["avg_ms"] for v in kernels.values()), 3),
            "parity_check": parity, "generate_s": round(t_gen, 2),
        }
        if coverage_paths:
            out["coverage_paths_ms_per_step"] = coverage_paths
        if comm is not None:
            out["comm"] = {"kind": comm.kind, "world": comm.world, "rccl_version": int(lib.ngsq_comm_rccl_version()),
                           "fallback_reason": comm.fallback_reason}
    ctx.free_batch(db)
    ctx.close()
    comm_kind = comm.kind if comm is not None else None
    shared_file = None
    if comm is not None and world > 1 and args.file_records > 0 and not mixed and not args.emulate_shard:
        # ---- every rank: ONE BAM file, each rank streaming its BGZF block range of it inside this process (HIP is up, the
        # communicator exists): the streaming rate of N GPUs on one file, without the process starts of `ngs qc --gpus N`
        try:
            shared_file = leg_file_sharded_in_process(lib, host, ffi, np, args, comm, rank, world, device)
        except Exception as e:  # noqa: BLE001 -- reported, never required
            shared_file = {"in_process": {"failed": f"{type(e).__name__}: {e}"}}
    if comm is not None:
        comm.barrier()
        comm.destroy()
    if rank == 0:
        # ---- the legs beside the headline number: rank 0 at N = 1 only, each bounded to seconds
        if world == 1 and not args.emulate_shard and not args.pmc_child:
            if args.cpu_sample > 0:
                out["cpu_baseline"] = cpu_baseline(lib, host, ffi, scfg, min(args.cpu_sample, n), max_len)
            if args.mixed_records > 0 and not mixed:
                out["mixed"] = guarded(leg_mixed, lib, host, ffi, np, args, device)
            if args.h2d_batch > 0 and not mixed:
                out["h2d_inclusive"] = guarded(leg_h2d, lib, host, ffi, args)
                out["stager"] = guarded(leg_stager, lib, host, ffi, args)
            if args.file_records > 0 and not mixed:
                fe = guarded(leg_file, lib, host, ffi, args)
                out["ingest_roofline"] = fe.pop("ingest_roofline", None) if isinstance(fe, dict) else None
                out["file_end_to_end"] = fe
            if args.extra_facet_legs and not mixed and args.facets == 0x1F:
                out["extra_facets"] = guarded(leg_extra_facets, lib, host, ffi, np, args.extra_facet_records)
            if args.all_facets_records > 0 and not mixed and args.facets == 0x1F:
                out["all_facets"] = guarded(leg_all_facets, lib, host, ffi, np, args, device)
            if args.whole_genome_records > 0 and not mixed and args.facets == 0x1F:
                out["whole_genome"] = guarded(leg_whole_genome, lib, host, ffi, np, args, device)
        elif world > 1 and args.file_records > 0 and not mixed and not args.emulate_shard:
            # the number the metric is named after, on N GPUs: ONE BAM file scanned by `ngs qc --gpus N` (the other ranks
            # of this launch have released their devices and are on their way out)
            out["file_end_to_end"] = guarded(leg_file_sharded, lib, host, ffi, args, world, comm_kind, shared_file)
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    return rc


def effective_cores() -> int:
    """Cores this process may really use: the cgroup's CPU quota (cpu.max) when there is one, else the online count.
    (The MI355X boxes of this pool show 256 online CPUs under a quota of 16.)"""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(int(parts[0]) / int(parts[1]))))
            else:
                q = int(parts[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                        n = min(n, max(1, q // int(f.read().split()[0])))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def guarded(fn, *a):
    try:
        return fn(*a)
    except Exception as e:  # these legs are reported, never required
        import traceback
        traceback.print_exc()
        return {"failed": f"{type(e).__name__}: {e}"}


def kernel_table(timing):
    return {k: {"avg_ms": round(v["total_ms"] / v["launches"], 4),
                "GBps": round(v["algo_bytes"] / max(v["total_ms"], 1e-9) / 1e6, 1)}
            for k, v in timing.items() if v["launches"] and v["total_ms"] > 0}


def check_invariants(ctx, ffi, total: int, args, mixed: bool, emulated: bool) -> str:
    """Size-independent properties of the whole-file result (tests/test_parity_gpu.py::test_full_size_properties):
    every rank holds the whole-file integers after ngsq_exchange + ngsq_finalize."""
    bad = []

    def need(cond, what):
        if not cond:
            bad.append(what)
    f = args.facets
    checked = 0
    if f & ffi.FACET_GENERAL:
        g = ctx.general()
        need(g["total"] == total, "general.total == records")
        need(g["primary"] + g["secondary"] + g["supplementary"] == total, "designations partition the records")
        checked += 2
    if f & ffi.FACET_QUALITY_SCORE:
        q = ctx.quality_scores()
        rows = q.sum(axis=1)
        if mixed:
            need(int(rows[0]) == total and (rows[:-1] >= rows[1:]).all(), "quality rows start at records and never grow")
        else:
            need((rows[:args.read_len] == total).all(), "every record reaches every cycle exactly once")
        checked += 1
    if f & ffi.FACET_TEMPLATE_LENGTH:
        h, processed, ignored = ctx.template_length()
        need(processed + ignored == total and int(h.sum()) == processed, "template length conserves records")
        checked += 1
    if f & ffi.FACET_GC_CONTENT:
        gc = ctx.gc_content()
        need(gc["processed"] + gc["ignored_flags"] + gc["ignored_too_short"] == total, "GC conserves records")
        need(int(gc["histogram"].sum()) == gc["processed"], "GC histogram sums to processed")
        need(gc["total_gc_count"] + gc["total_at_count"] + gc["total_other_count"] == 100 * gc["processed"],
             "GC window is 100 bases per processed read")
        checked += 3
    if f & ffi.FACET_COVERAGE and not emulated:
        seen, hist, ign, bins = ctx.coverage_sequence(0)
        need(seen and int(hist.sum()) + ign == CHR1 + 1, "depth histogram counts L+1 positions")
        need(ctx.coverage_nonsensical() == 0, "no position beyond the sequence")
        need(not ctx.coverage_sequence(1)[0], "chr2 has no entry")
        if not mixed and f & ffi.FACET_GENERAL:
            need(int(bins.sum()) == args.read_len * (total - g["unmapped"]), "depth total == read_len x mapped reads")
            checked += 1
        checked += 3
    return ("ok: %d full-size invariants" % checked) if not bad else "FAILED: " + "; ".join(bad)


def ingest_traffic(args, algo_bytes_per_launch: int):
    """ingest_roofline.traffic: measured by this run (live_inflate_traffic) when it could be, else the committed summary."""
    li = getattr(args, "live_inflate", None)
    if li and "per_algo_byte_raw" in li:
        return {"traffic": round(li["per_algo_byte_raw"] * algo_bytes_per_launch),
                "traffic_read_side_doubled": round(li["per_algo_byte_read_side_doubled"] * algo_bytes_per_launch),
                "traffic_per_algorithmic_byte": [round(li["per_algo_byte_raw"], 3), round(li["per_algo_byte_read_side_doubled"], 3)],
                "traffic_source": li["traffic_source"], "traffic_measured_in_this_run": True}
    out = inflate_traffic(algo_bytes_per_launch)
    if li:
        out["live_traffic_unavailable"] = li.get("traffic_source")
    return out


def inflate_traffic(algo_bytes_per_launch: int):
    """HBM bytes per launch of k_bgzf_inflate, scaled from the committed PMC summary of tools/bench_inflate.py (one launch over
    1.5 GB of algorithmic bytes) to this run's launch size.  Raw and with the read side doubled: the doubling is the guide's
    correction for wide streaming reads, which the decoder's dword / byte loads are not -- the truth lies between the two."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_inflate_traffic.json")))
    try:
        with open(files[-1]) as f:
            doc = json.load(f)
        k = next(v for name, v in doc["kernels"].items() if "k_bgzf_inflate" in name)
        ref_algo = 404_000_000 + 1_095_000_000   # the workload line of that file: 404 MB of blocks -> 1095 MB
        scale = algo_bytes_per_launch / ref_algo
        return {"traffic": round((k["fetch_raw"] + k["write_raw"]) * scale), "traffic_read_side_doubled": round(k["hbm_bytes_corrected"] * scale),
                "traffic_source": os.path.relpath(files[-1], ROOT) + " scaled by the launch's algorithmic bytes",
                "traffic_measured_in_this_run": False}
    except Exception as e:  # noqa: BLE001
        return {"traffic": None, "traffic_source": f"unavailable: {e}"}


def live_pmc_traffic(args, mixed=False):
    """HBM bytes per launch of the dominant kernel from two rocprofv3 --pmc passes of this same workload (3 steps each), run as
    child processes BEFORE this process initialises HIP.  MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE in separate passes;
    on gfx950 the read side is doubled (checked on kernels with known byte counts: profiles/r04_fetch_calib.txt).  Returns
    (bytes or None, source text)."""
    import csv
    import glob
    import shutil
    import tempfile
    if not shutil.which("rocprofv3"):
        return None, "rocprofv3 not on PATH"
    # Not under a profiler: a tool library preloaded into THIS process has initialised the GPU already, and starting another
    # program from such a process is what the GPU boxes of this pool refuse (and a profiler inside a profiler measures nothing).
    for k, v in os.environ.items():
        if k in ("HSA_TOOLS_LIB", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_REGISTER_FORCE_LOAD") or k.startswith("ROCPROF") or \
                (k == "LD_PRELOAD" and "rocprof" in v.lower()):
            return None, f"running under a profiler ({k} is set)"
    tmp = tempfile.mkdtemp(prefix="ngsq_pmc_", dir=os.environ.get("TMPDIR", "/tmp"))
    vals = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out_dir = os.path.join(tmp, counter)
            cmd = ["rocprofv3", "--pmc", counter, "--kernel-trace", "-d", out_dir, "-o", "out", "--output-format", "csv", "--",
                   sys.executable, os.path.abspath(__file__), "--pmc-child", "--steps", "3", "--warmup", "1", "--no-timing",
                   "--records", str(args.mixed_records if mixed else args.records), "--read-len", str(args.read_len),
                   "--facets", hex(args.facets & 0x1F if mixed else args.facets), "--coverage", "auto" if mixed else args.coverage]
            if mixed:
                cmd += ["--workload", "mixed", "--mixed-max-len", str(args.mixed_max_len)]
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=90, cwd=tmp)   # (a pass takes ~5 s)
            files = glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None, f"rocprofv3 --pmc {counter} failed (rc {r.returncode}): {(r.stderr or '')[-200:]}"
            tot, cnt = 0.0, 0
            with open(files[0], newline="") as f:
                for row in csv.DictReader(f):
                    if row["Counter_Name"] == counter and any(k in row["Kernel_Name"] for k in (("k_qual_ragged",) if mixed else ("k_qual_perm", "k_qual_win"))):
                        tot += float(row["Counter_Value"])
                        cnt += 1
            if not cnt:
                return None, f"no {counter} rows for the quality kernel"
            vals[counter] = tot / cnt * 1024.0
        return int(round(2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"])), ("two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of this workload, 3 steps "
                                                                            "each, run by this command before its own timed steps; read side doubled (gfx950)")
    except Exception as e:  # noqa: BLE001 -- never required
        return None, f"{type(e).__name__}: {e}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def pmc_inflate_child(args, json_fd) -> int:
    """Under rocprofv3 --pmc: one in-process scan of the BAM the parent wrote; prints the algorithmic bytes of its inflate launches."""
    import ctypes as C
    from ngs_amd import ffi, host
    lib = ffi.load_library()
    path = args.pmc_inflate_child
    ctx = host.QcContext([CHR1, CHR2], [1, 1], max_read_len=1024, gc_seed=GC_SEED, sorted_input=True, timing=False, lib=lib)
    h = C.c_void_p()
    if lib.ngsq_bam_open(path.encode(), 0, C.byref(h)) != 0:
        return 4
    n = 0
    while True:
        b = ffi.Batch()
        if lib.ngsq_bam_next_batch_device(h, ctx._ctx, 1 << 22, C.byref(b)) != 0 or b.n_records == 0:
            break
        n += int(b.n_records)
        lib.ngsq_process_batch(ctx._ctx, C.byref(b), ffi.PASS_BOTH)
    lib.ngsq_bam_close(h)
    ctx.finalize()
    t = ctx.kernel_timing()["bgzf_inflate"]
    ctx.close()
    os.write(json_fd, (json.dumps({"records": n, "inflate_algo_bytes": t["algo_bytes"], "launches": t["launches"]}) + "\n").encode())
    return 0


def live_inflate_traffic(args):
    """HBM bytes per ALGORITHMIC byte of k_bgzf_inflate, measured by this command: a small file of the bench's plain style written here
    (host code, no HIP), scanned once by a child under `rocprofv3 --pmc FETCH_SIZE` and once under `--pmc WRITE_SIZE` (separate
    passes, MI355X_MICROARCH.md).  Returns a dict for ingest_roofline, or {"traffic_source": why not}."""
    import csv
    import ctypes as C
    import glob
    import shutil
    import tempfile
    if not shutil.which("rocprofv3"):
        return {"traffic_source": "unavailable: rocprofv3 not on PATH"}
    for k, v in os.environ.items():
        if k in ("HSA_TOOLS_LIB", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_REGISTER_FORCE_LOAD") or k.startswith("ROCPROF") or \
                (k == "LD_PRELOAD" and "rocprof" in v.lower()):
            return {"traffic_source": f"unavailable: running under a profiler ({k} is set)"}
    tmp = tempfile.mkdtemp(prefix="ngsq_pmci_", dir=os.environ.get("TMPDIR", "/tmp"))
    try:
        from ngs_amd import ffi, host
        lib = ffi.load_library()           # (loading the library and writing a file with it touch no HIP)
        n = 4_000_000
        bam = os.path.join(tmp, "small.bam")
        cfg = host.synth_config(n, read_len=args.read_len, ref_len=CHR1, n_refs=2)
        if lib.ngsq_synth_write_bam(C.byref(cfg), bam.encode(), n, args.file_level, 0) != 0:
            return {"traffic_source": "unavailable: could not write the small file"}
        vals, algo = {}, None
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out_dir = os.path.join(tmp, counter)
            cmd = ["rocprofv3", "--pmc", counter, "--kernel-trace", "-d", out_dir, "-o", "out", "--output-format", "csv", "--",
                   sys.executable, os.path.abspath(__file__), "--pmc-inflate-child", bam]
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=120, cwd=tmp)
            files = glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return {"traffic_source": f"unavailable: rocprofv3 --pmc {counter} failed (rc {r.returncode}): {(r.stderr or '')[-200:]}"}
            line = next((ln for ln in r.stdout.splitlines() if ln.startswith("{") and "inflate_algo_bytes" in ln), None)
            if line is None:
                return {"traffic_source": "unavailable: the profiled child printed no line"}
            algo = json.loads(line)["inflate_algo_bytes"]
            tot = 0.0
            with open(files[0], newline="") as f:
                for row in csv.DictReader(f):
                    if row["Counter_Name"] == counter and "k_bgzf_inflate" in row["Kernel_Name"]:
                        tot += float(row["Counter_Value"])
            if not tot:
                return {"traffic_source": f"unavailable: no {counter} rows for k_bgzf_inflate"}
            vals[counter] = tot * 1024.0
        return {"per_algo_byte_raw": (vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) / algo,
                "per_algo_byte_read_side_doubled": (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) / algo,
                "traffic_source": "two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of one scan of a 4 M-record file of the same style, run by "
                                  "this command before its own timed steps; scaled by this run's algorithmic bytes per launch"}
    except Exception as e:  # noqa: BLE001 -- never required
        return {"traffic_source": f"unavailable: {type(e).__name__}: {e}"}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def pmc_traffic(n: int, args):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC summary
    (profiles/*_traffic.json, made by tools/collect_traffic.py from separate --pmc FETCH_SIZE /
    WRITE_SIZE passes of this same command).  Only valid for the workload it was collected on."""
    import glob
    mixed = args.workload == "mixed"
    if n != 100_000_000 or (not mixed and args.read_len != 150) or (mixed and args.mixed_max_len != 300):
        return None, "no PMC summary for this workload"
    files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_mixed_traffic.json" if mixed else "r[0-9][0-9]_traffic.json")))
    if not files:
        return None, "profiles/*_traffic.json absent"
    try:
        with open(files[-1]) as f:
            doc = json.load(f)
        for name, v in doc["kernels"].items():
            if "k_qual_perm" in name or "k_qual_win" in name or "k_qual_ragged" in name:
                return v["hbm_bytes_corrected"], os.path.relpath(files[-1], ROOT)
    except Exception as e:  # noqa: BLE001
        return None, f"unreadable: {e}"
    return None, "kernel not in summary"


# ---------------------------------------------------------------------------------------------
# CPU baseline: the oracle (test infrastructure) timed on the host cores -- a reported baseline
# ---------------------------------------------------------------------------------------------
def cpu_baseline(lib, host, ffi, scfg, sample: int, max_len: int):
    """The CPU restatement of the reference loop (oracle/, single thread, record at a time, two passes) on a
    bounded sample of the same workload: 1 warm + 3 timed runs of a 1/4 sample for the scan rate (median), one run
    of the full sample including the O(L) chr1 teardown.  Beside it, the same scan spread over all host cores
    (private state per thread, one teardown) -- NOT reference behaviour: the reference is one thread
    (src/qc/command.rs:226-421)."""
    try:
        from concurrent.futures import ThreadPoolExecutor
        from oracle import oracle_py
        kw = dict(facets=ffi.FACETS_DEFAULT, max_read_len=max_len, gc_seed=GC_SEED)
        chunk = 250_000

        def scan(orc, hb):
            for lo in range(0, hb.n, chunk):
                orc.process_batch(hb.slice(lo, min(hb.n, lo + chunk)))
            return orc.elapsed_seconds()

        hb = host.synth_host_batch(scfg, 0, sample, lib)
        orc = oracle_py.Oracle([CHR1, CHR2], [1, 1], **kw)
        t_scan = scan(orc, hb)
        orc.finalize()
        t_all = orc.elapsed_seconds()
        orc.close()
        # repeatability of the scan rate: 3 runs on a quarter of the sample
        quarter = hb.slice(0, max(1, sample // 4))
        rates = []
        for _ in range(3):
            o = oracle_py.Oracle([CHR1, CHR2], [1, 1], **kw)
            rates.append(quarter.n / scan(o, quarter))
            o.close()
        rates.sort()
        # all cores: threads (the C calls release the GIL), private state each, disjoint contiguous slices
        cores = effective_cores()
        workers = max(1, min(cores, 64))
        per = max(100_000, min(1_000_000, sample // 2))

        def work(w):
            b = host.synth_host_batch(scfg, w * per, per, lib)
            o = oracle_py.Oracle([CHR1, CHR2], [1, 1], **kw)
            t0 = time.perf_counter()
            scan(o, b)
            dt = time.perf_counter() - t0
            o.close()
            return dt
        t0 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=workers) as ex:
            list(ex.map(work, range(workers)))
        t_par = time.perf_counter() - t0
        return {"value": round(sample / t_all, 1), "unit": "records/s", "cores": 1, "kind": "port",
                "sample": ("first %d records of the workload, all default facets; %.1f s facet loop + %.1f s chr1 "
                           "coverage teardown (O(L), not amortised over the full file)"
                           % (sample, t_scan, t_all - t_scan)),
                "scan_only_value": round(sample / t_scan, 1),
                "scan_only_runs_median_of_3": round(rates[1], 1),
                "all_cores": {"value": round(workers * per / t_par, 1), "unit": "records/s", "cores": workers,
                              "host_cpus_online": os.cpu_count(), "cpu_quota_cores": cores,
                              "note": ("NOT reference behaviour (the reference is single-threaded): %d threads x %d records "
                                       "each with private facet state, wall clock incl. batch generation, no merge and no "
                                       "teardown" % (workers, per))}}
    except Exception as e:  # the baseline is reported, never required
        return {"value": None, "unit": "records/s", "cores": 1, "kind": "port", "sample": f"failed: {e}"}


# ---------------------------------------------------------------------------------------------
# the other two rates of SURVEY 8d: H2D-inclusive and from the BAM file
# ---------------------------------------------------------------------------------------------
def pinned_copy(lib, host, np, C, hb):
    """Copy a HostBatch's columns into hipHostMalloc'd memory (the pointers are returned for freeing)."""
    keep, cols = [], {}
    for k, a in hb.cols.items():
        if a is None:
            cols[k] = None
            continue
        p = C.c_void_p()
        assert lib.ngsq_host_malloc_pinned(max(a.nbytes, 64), C.byref(p)) == 0
        dst = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(max(a.nbytes, 64),))
        dst[:a.nbytes] = a.view(np.uint8).reshape(-1)
        cols[k] = dst[:a.nbytes].view(a.dtype)
        keep.append(p)
    return host.HostBatch(hb.n, cols, hb.seq_stride, hb.qual_stride, hb.cigar_stride, hb.first_record_index), keep


def leg_h2d(lib, host, ffi, args):
    """Pinned host SoA batches through ngsq_process_batch: H2D copy + all default facet kernels per batch."""
    import ctypes as C
    import numpy as np
    scfg = host.synth_config(100_000_000, read_len=args.read_len, ref_len=CHR1, n_refs=2)
    hb = host.synth_host_batch(scfg, 0, args.h2d_batch, lib)
    pb, keep = pinned_copy(lib, host, np, C, hb)
    # the same batch is re-sent: not a sorted stream, so Coverage runs on the difference arrays here
    ctx = host.QcContext([CHR1, CHR2], [1, 1], max_read_len=args.read_len, gc_seed=GC_SEED, lib=lib)
    try:
        ctx.process_batch(pb)
        ctx.synchronize()
        ctx.reset()
        reps = 8
        t0 = time.perf_counter()
        for _ in range(reps):
            ctx.process_batch(pb)
        ctx.synchronize()
        dt = time.perf_counter() - t0
        ctx.finalize()
        bytes_rec = 254.0 if args.read_len == 150 else 25 + 4 + (args.read_len + 1) // 2 + args.read_len
        return {"value": round(reps * pb.n / dt, 1), "unit": "records/s", "GB_per_s": round(reps * pb.n * bytes_rec / dt / 1e9, 1),
                "batch_records": pb.n, "batches": reps, "memory": "pinned host SoA (hipHostMalloc), the same batch re-sent",
                "note": "PCIe Gen5 x16 bound (~63 GB/s); never `value`"}
    finally:
        ctx.close()
        for p in keep:
            lib.ngsq_host_free_pinned(p)


def leg_stager(lib, host, ffi, args):
    """The per-record adapter (include/ngsq_stage.h) on one host core: every record of a host batch through
    ngsq_stager_push_packed (in C: ngsq_stager_push_records), a flush per full stager -- what a host that keeps the reference's
    `process(&Record)` loop gets."""
    import ctypes as C
    scfg = host.synth_config(100_000_000, read_len=args.read_len, ref_len=CHR1, n_refs=2)
    n = min(args.h2d_batch, 4_000_000)
    hb = host.synth_host_batch(scfg, 0, n, lib)
    src = hb.struct()
    ctx = host.QcContext([CHR1, CHR2], [1, 1], max_read_len=args.read_len, gc_seed=GC_SEED, lib=lib)
    st = C.c_void_p()
    try:
        if lib.ngsq_stager_create(1 << 20, ffi.STAGE_PINNED, C.byref(st)) != 0:
            raise RuntimeError(lib.ngsq_stager_last_error(None).decode())
        took = C.c_uint64(0)

        def one_pass():
            push_s, first = 0.0, 0
            while first < n:
                t = time.perf_counter()
                if lib.ngsq_stager_push_records(st, C.byref(src), first, n - first, C.byref(took)) != 0:
                    raise RuntimeError(lib.ngsq_stager_last_error(st).decode())
                push_s += time.perf_counter() - t
                first += took.value
                if lib.ngsq_stager_flush(st, ctx._ctx, ffi.PASS_BOTH) != 0:
                    raise RuntimeError((lib.ngsq_last_error(ctx._ctx) or b"ngsq_stager_flush failed").decode())
            return push_s
        one_pass()
        ctx.synchronize()
        ctx.reset()
        reps = 3
        t0 = time.perf_counter()
        push_s = sum(one_pass() for _ in range(reps))
        ctx.synchronize()
        dt = time.perf_counter() - t0
        ctx.finalize()
        return {"value": round(reps * n / dt, 1), "unit": "records/s", "push_only_records_per_s": round(reps * n / push_s, 1), "cores": 1,
                "records": reps * n, "stager_capacity": 1 << 20,
                "note": "one thread pushes record by record and flushes; a flush queues its copies and the pushes go on into the stager's second set of columns (round 6: until then the thread sat out every copy)"}
    finally:
        if st:
            lib.ngsq_stager_destroy(st)
        ctx.close()


def fs_type_of(path: str) -> str:
    best, kind = "", "?"
    try:
        with open("/proc/mounts") as f:
            for line in f:
                parts = line.split()
                if len(parts) >= 3 and os.path.abspath(path).startswith(parts[1]) and len(parts[1]) > len(best):
                    best, kind = parts[1], parts[2]
    except OSError:
        pass
    return kind


def drop_from_page_cache(path: str) -> bool:
    """posix_fadvise(DONTNEED) on a synced file: the next read comes from storage (a no-op on tmpfs)."""
    try:
        fd = os.open(path, os.O_RDONLY)
        try:
            os.fsync(fd)
            os.posix_fadvise(fd, 0, 0, os.POSIX_FADV_DONTNEED)
        finally:
            os.close(fd)
        return True
    except (OSError, AttributeError):
        return False


def raw_read_rate(path: str, threads: int = 8) -> float:
    """GB/s of plain parallel pread()s over the whole file (what the storage under it delivers to this process)."""
    from concurrent.futures import ThreadPoolExecutor
    size = os.path.getsize(path)
    piece = 64 << 20
    fd = os.open(path, os.O_RDONLY)

    def rd(off):
        left, o = min(piece, size - off), off
        while left > 0:
            got = len(os.pread(fd, min(left, 8 << 20), o))
            if not got:
                break
            left -= got
            o += got
    t0 = time.perf_counter()
    try:
        with ThreadPoolExecutor(max_workers=threads) as ex:
            list(ex.map(rd, range(0, size, piece)))
    finally:
        os.close(fd)
    return size / max(time.perf_counter() - t0, 1e-9) / 1e9


FILE_SCANS = 5   # timed scans of every file leg (behind the first scan of the process); the legs report all of them and the median


def scan_file_in_process(lib, host, ffi, ctx, bam, n, reps, names=("chr1", "chr2")):
    """ngsq_bam_open -> ngsq_bam_next_batch_device -> ngsq_process_batch -> ngsq_finalize, `reps` times; returns
    (seconds of every scan, kernel timing of the best one, (rate after the first batch, seconds to it), document)."""
    import ctypes as C
    times, best, timing, after_first, doc = [], None, None, None, None
    for rep in range(reps):
        ctx.reset()
        ctx.kernel_timing_reset()
        t0 = time.perf_counter()
        h = C.c_void_p()
        if lib.ngsq_bam_open(bam.encode(), 0, C.byref(h)) != 0:
            raise RuntimeError(lib.ngsq_bam_last_error().decode())
        got, first = 0, None
        while True:
            b = ffi.Batch()
            if lib.ngsq_bam_next_batch_device(h, ctx._ctx, 1 << 22, C.byref(b)) != 0:
                raise RuntimeError(lib.ngsq_bam_last_error().decode())
            if b.n_records == 0:
                break
            got += int(b.n_records)
            if first is None:   # start-up (buffers allocated and pinned, first chunk read and inflated) ends here
                first = (time.perf_counter(), got)
            if lib.ngsq_process_batch(ctx._ctx, C.byref(b), ffi.PASS_BOTH) != 0:
                raise RuntimeError(lib.ngsq_last_error(ctx._ctx).decode())
        lib.ngsq_bam_close(h)
        ctx.finalize()
        dt = time.perf_counter() - t0
        assert got == n, (got, n)
        times.append(round(dt, 3))
        if best is None or dt < best:
            best, timing = dt, ctx.kernel_timing()
            after_first = ((n - first[1]) / max(t0 + dt - first[0], 1e-9), first[0] - t0) if first and n > first[1] else None
    doc = ctx.results(list(names))
    return times, best, timing, after_first, doc


def leg_file(lib, host, ffi, args):
    """A synthetic BGZF BAM written once -> (a) `ngs qc` as a child process, wall clock including process start and HIP
    initialisation, device ingest (the default) and host ingest; (b) the same file through the same entry points inside
    this process (HIP already initialised): the steady-state rate.  JSON of (a) device == (a) host == (b)."""
    import ctypes as C
    import tempfile
    from ngs_amd import build
    n = args.file_records
    tmp = tempfile.mkdtemp(prefix="ngsq_bench_", dir=os.environ.get("TMPDIR", "/tmp"))
    bam = os.path.join(tmp, "synth.bam")
    out = {"records": n, "zlib_level": args.file_level, "host_cores": effective_cores(),
           "source": "page cache (the file was written seconds before it is read; see cold_cache for storage)",
           "filesystem": fs_type_of(tmp)}
    try:
        fcfg = host.synth_config(n, read_len=args.read_len, ref_len=CHR1, n_refs=2)
        t0 = time.perf_counter()
        assert lib.ngsq_synth_write_bam(C.byref(fcfg), bam.encode(), n, args.file_level, 0) == 0, lib.ngsq_bam_last_error()
        out["bam_write_s"] = round(time.perf_counter() - t0, 2)
        out["bam_bytes"] = os.path.getsize(bam)
        t0 = time.perf_counter()
        os.sync()   # let the write-back of the fresh file finish: it otherwise competes with the timed reads a few seconds later
        out["sync_s"] = round(time.perf_counter() - t0, 2)
        out["page_cache_settling_reads_GB_per_s"] = settle_page_cache(bam)
        ngs = build.build_cli(verbose=False)
        docs = {}
        for ingest, runs in (("device", 3), ("host", 1)):
            best, each = None, []
            for _ in range(runs):
                t0 = time.perf_counter()
                r = subprocess.run([ngs, "-q", "qc", bam, "GRCh38_no_alt_AnalysisSet", "-o", tmp, "--ingest", ingest],
                                   capture_output=True, text=True)
                dt = time.perf_counter() - t0
                if r.returncode != 0:
                    raise RuntimeError(f"ngs qc --ingest {ingest}: {r.stderr[-400:]}")
                each.append(round(dt, 3))
            best = median(each)
            with open(os.path.join(tmp, "synth.bam.results.json")) as f:
                docs[ingest] = json.load(f)
            out[f"cli_{ingest}_ingest"] = {"seconds": round(best, 3), "seconds_each_run": each, "records_per_s": round(n / best, 1),
                                           "compressed_GB_per_s": round(out["bam_bytes"] / best / 1e9, 2),
                                           "includes": "process start, HIP initialisation, header + index checks, JSON write"}
            if ingest == "device":   # one more run with the command's own milestones (NGSQ_INGEST_TRACE=1): where its wall clock goes
                t0 = time.perf_counter()
                r = subprocess.run([ngs, "-q", "qc", bam, "GRCh38_no_alt_AnalysisSet", "-o", tmp, "--ingest", ingest],
                                   capture_output=True, text=True, env=dict(os.environ, NGSQ_INGEST_TRACE="1"))
                wall = (time.perf_counter() - t0) * 1e3
                marks = []
                for ln in r.stderr.splitlines():
                    if ln.startswith("[ngs]") and " ms " in ln:
                        ms, what = ln[5:].split(" ms ", 1)
                        marks.append((what.strip(), float(ms)))
                if r.returncode == 0 and marks:
                    ph, prev = {}, 0.0
                    for what, ms in marks[1:]:
                        ph["until " + what] = round(ms - prev, 1)
                        prev = ms
                    ph["process start + exit (wall clock of the command - its own last milestone)"] = round(wall - marks[-1][1], 1)
                    out["cli_device_ingest"]["phases_ms"] = ph
                    out["cli_device_ingest"]["phases_wall_ms"] = round(wall, 1)
        # (b) in process: ngsq_bam_open -> ngsq_bam_next_batch_device -> ngsq_process_batch -> ngsq_finalize
        ctx = host.QcContext([CHR1, CHR2], [1, 1], max_read_len=1024, gc_seed=GC_SEED, sorted_input=True, timing=True, lib=lib)
        try:
            lib.ngsq_release_cached_memory()   # the first file of a process finds no cached blocks
            times, best, timing, after_first, doc = scan_file_in_process(lib, host, ffi, ctx, bam, n, 1 + FILE_SCANS)
            docs["in_process"] = doc
            # scan 0 allocates and pins the pipeline's buffers; `value` = the MEDIAN of the five scans behind it (all in the line)
            med = median(times[1:])
            out["value"] = round(n / med, 1)
            out["unit"] = "records/s"
            out["value_is"] = "median of scans 1..%d of seconds_each_scan (scan 0: the first file of the process)" % FILE_SCANS
            out["in_process_device_ingest"] = {"seconds": round(med, 3), "records_per_s": round(n / med, 1),
                                               "seconds_best_scan": round(best, 3),
                                               "scan_spread_pct": round(100 * (max(times[1:]) / min(times[1:]) - 1), 1),
                                               "compressed_GB_per_s": round(out["bam_bytes"] / med / 1e9, 2),
                                               "includes": "file open, reads, H2D of the compressed bytes, inflate, parse, "
                                                           "all default facets, finalize, close",
                                               "seconds_each_scan": times,
                                               "note": "scan 0 allocates and pins the pipeline's buffers (and frees them in close "
                                                       "without the cache); later files of a process take them from the block "
                                                       "cache (mem_pool.cpp; NGSQ_POOL_MB=0 disables it)",
                                               "kernels": kernel_table(timing)}
            if after_first:
                out["in_process_device_ingest"]["first_batch_after_s"] = round(after_first[1], 3)
                out["in_process_device_ingest"]["records_per_s_after_first_batch"] = round(after_first[0], 1)
            inf = timing.get("bgzf_inflate")
            if inf and inf["launches"] and inf["total_ms"] > 0:
                gbs = inf["algo_bytes"] / inf["total_ms"] / 1e6
                out["ingest_roofline"] = {"bound": "hbm", "kernel": "k_bgzf_inflate", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS,
                                          "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                                          **ingest_traffic(args, inf["algo_bytes"] // inf["launches"]),
                                          "avg_launch_ms": round(inf["total_ms"] / inf["launches"], 3),
                                          "algo_bytes_per_launch": inf["algo_bytes"] // inf["launches"],
                                          "note": "algorithmic bytes = compressed bytes read + inflated bytes written; the kernel is "
                                                  "bound by instruction issue (24 persistent decoder waves per CU, one BGZF block at a "
                                                  "time each: ~24 instructions per DEFLATE symbol keep the scalar and vector ports "
                                                  "of a CU 60-70 % busy), not by HBM; launches of the first chunks of a scan are "
                                                  "smaller (32, 64, ... MiB)"}
            # ---- the same scan with the file dropped from the page cache first (storage -> pinned memory -> GPU), and what
            # the storage under the file delivers to plain parallel pread()s
            if drop_from_page_cache(bam):
                t_c, best_c, _, _, doc_c = scan_file_in_process(lib, host, ffi, ctx, bam, n, 1)
                cold = {"seconds": round(best_c, 3), "records_per_s": round(n / best_c, 1),
                        "compressed_GB_per_s": round(out["bam_bytes"] / best_c / 1e9, 2),
                        "how": "os.sync + fsync + posix_fadvise(POSIX_FADV_DONTNEED) on the file, then one in-process scan",
                        "same_document": json.dumps(doc_c, sort_keys=True) == json.dumps(doc, sort_keys=True)}
                if drop_from_page_cache(bam):
                    cold["storage_pread_GB_per_s"] = round(raw_read_rate(bam), 2)
                cold["page_cache_pread_GB_per_s"] = round(raw_read_rate(bam), 2)   # the file is cached again by now
                out["cold_cache"] = cold
            # ---- a larger file beside it (1 GiB chunks, whose inflate launches run fuller): in process only
            if args.file_big_records > n:
                big = os.path.join(tmp, "big.bam")
                try:
                    # (the writer is zlib on the host cores, ~1.25 M records/s on the 16 cores these boxes grant: 200 M records take
                    # it 160 s -- the longest single item of the default run; a slower writer or a small disk gets a smaller file)
                    nb = int(min(args.file_big_records, max(n, n / max(out["bam_write_s"], 1e-3) * 175.0)))
                    try:
                        import shutil
                        nb = int(min(nb, max(n, shutil.disk_usage(tmp).free / 3 / 105)))
                    except OSError:
                        pass
                    bcfg = host.synth_config(nb, read_len=args.read_len, ref_len=CHR1, n_refs=2)
                    t0 = time.perf_counter()
                    assert lib.ngsq_synth_write_bam(C.byref(bcfg), big.encode(), nb, args.file_level, 0) == 0, lib.ngsq_bam_last_error()
                    tw = time.perf_counter() - t0
                    os.sync()
                    settle_page_cache(big)
                    t_b, best_b, timing_b, after_b, doc_b = scan_file_in_process(lib, host, ffi, ctx, big, nb, FILE_SCANS)
                    best_b = median(t_b)
                    out["big_file"] = {"records": nb, "bam_bytes": os.path.getsize(big), "bam_write_s": round(tw, 1),
                                       "seconds_each_scan": t_b, "records_per_s": round(nb / best_b, 1), "value_is": "median scan",
                                       "compressed_GB_per_s": round(os.path.getsize(big) / best_b / 1e9, 2),
                                       "records_per_s_after_first_batch": round(after_b[0], 1) if after_b else None,
                                       "check_total": doc_b["general"]["records"]["total"], "source": "page cache"}
                except Exception as e:  # noqa: BLE001
                    out["big_file"] = {"failed": f"{type(e).__name__}: {e}"}
            # ---- what an aligner writes: long names, a dozen tags per record, multi-operation CIGARs, real mate positions --
            # 370 bytes per record where the plain file has 273, and a screen that has to tell records from their tails
            if args.file_realistic_records > 0:
                out["realistic"] = guarded(leg_file_realistic, lib, host, ffi, args, ctx, tmp, ngs, out["bam_write_s"] / max(n, 1))
                if isinstance(out["realistic"], dict) and "cli_all_facets" in out["realistic"]:
                    out["cli_all_facets"] = out["realistic"].pop("cli_all_facets")   # (file_end_to_end.cli_all_facets: on the realistic file)
        finally:
            ctx.close()
        def strip(d):  # the GC window offsets are a function of (seed, record id = virtual offset): identical across the runs
            return json.dumps(d, sort_keys=True)
        out["json_equal_device_host_inprocess"] = strip(docs["device"]) == strip(docs["host"]) == strip(docs["in_process"])
        out["check_total"] = docs["device"]["general"]["records"]["total"]
        return out
    finally:
        for f in os.listdir(tmp):
            os.remove(os.path.join(tmp, f))
        os.rmdir(tmp)


def write_bench_fasta(np, lib, host, scfg, names, lens, path):
    """The FASTA the reads of `scfg` (seq_model FROM_REFERENCE) were sampled from, as the analysis set comes: 60 bases per line,
    soft-masked (runs of a few hundred lower-case bases, about half of it), a .fai-less plain file of the genome's size."""
    lut = np.frombuffer(b"=ACMGRSVTWYHKDBN", dtype=np.uint8)
    rng = np.random.default_rng(3)
    with open(path, "wb") as f:
        for r, (name, L) in enumerate(zip(names, lens)):
            s_ = lut[host.synth_reference(scfg, r, L, lib)]
            # soft-masking: alternate runs, lengths 50-700, lower-cased with probability one half
            edges = np.cumsum(rng.integers(50, 700, L // 375 + 2))
            edges = edges[edges < L]
            run = np.zeros(L, dtype=np.uint8)
            run[edges] = 1
            lower = (np.cumsum(run, dtype=np.uint32) & 1).astype(bool)
            s_ = np.where(lower, s_ | 0x20, s_).astype(np.uint8)
            f.write(f">{name}  AC:stand-in  gi:0  LN:{L}  rl:Chromosome  M5:0  AS:GRCh38\n".encode())
            full = L // 60 * 60
            if full:
                lines = np.empty((full // 60, 61), dtype=np.uint8)
                lines[:, :60] = s_[:full].reshape(-1, 60)
                lines[:, 60] = 10
                f.write(lines.tobytes())
            if L > full:
                f.write(s_[full:].tobytes() + b"\n")
    return os.path.getsize(path)


def write_bench_gff(np, names, lens, primary, path, n_rows=3_400_000):
    """A GENCODE-shaped GFF3: ~3.4 M rows (gene / transcript / exon / CDS / UTR / codon rows with their attribute strings, ~330
    bytes each), on the 24 chromosomes and chrM, both strands.  Returns (bytes, the model the facet keeps of it)."""
    rng = np.random.default_rng(11)
    types = ["gene", "transcript", "exon", "CDS", "five_prime_UTR", "three_prime_UTR", "start_codon", "stop_codon"]
    tp = rng.choice(len(types), n_rows, p=[.02, .08, .48, .27, .06, .05, .02, .02])
    seq = rng.choice(25, n_rows)
    seq.sort()
    L = np.array(lens, dtype=np.int64)[seq]
    start = (rng.random(n_rows) * (L - 6000)).astype(np.int64) + 1
    stop = start + np.where(tp == 0, rng.integers(1000, 5000, n_rows), rng.integers(30, 900, n_rows))
    strand = rng.integers(0, 2, n_rows)
    role = {"five_prime_UTR": 0, "three_prime_UTR": 1, "CDS": 2, "exon": 3, "gene": 4}
    keep = np.array([types[t] in role and primary[q] for t, q in zip(tp, seq)], dtype=bool)
    model = (seq[keep].astype(np.uint32), np.array([role.get(types[t], 0) for t in tp[keep]], dtype=np.uint32), start[keep].astype(np.uint32),
             stop[keep].astype(np.uint32))
    attr = ("ID={0}:ENST00000{1:06d}.{2};Parent=ENST00000{1:06d}.{2};gene_id=ENSG00000{3:06d}.{2};transcript_id=ENST00000{1:06d}.{2};"
            "gene_type=protein_coding;gene_name=GENE{3};transcript_type=protein_coding;transcript_name=GENE{3}-20{2};exon_number={4};"
            "exon_id=ENSE0000{1:07d}.1;level=2;protein_id=ENSP00000{1:06d}.{2};transcript_support_level=1;tag=basic,Ensembl_canonical,MANE_Select,appris_principal_1,CCDS")
    with open(path, "w") as f:
        f.write("##gff-version 3\n#description: stand-in for a GENCODE comprehensive annotation\n#provider: bench.py\n")
        chunk = []
        for k in range(n_rows):
            t = types[tp[k]]
            chunk.append(f"{names[seq[k]]}\tHAVANA\t{t}\t{start[k]}\t{stop[k]}\t.\t{'+-'[strand[k]]}\t{'.' if t != 'CDS' else k % 3}\t" +
                         attr.format(t, k % 1000000, 1 + k % 9, k % 60000, 1 + k % 30))
            if len(chunk) == 200_000:
                f.write("\n".join(chunk) + "\n")
                chunk = []
        if chunk:
            f.write("\n".join(chunk) + "\n")
    return os.path.getsize(path), model


def run_cli_traced(ngs, argv, tmp, env_extra=None):
    """`ngs qc ...` as a child with its own milestones (NGSQ_INGEST_TRACE=1): wall clock, phases, and what it says of the set-up."""
    t0 = time.perf_counter()
    r = subprocess.run([ngs, "-q", "qc", *argv, "-o", tmp], capture_output=True, text=True, env=dict(os.environ, NGSQ_INGEST_TRACE="1", **(env_extra or {})))
    wall = (time.perf_counter() - t0) * 1e3
    if r.returncode != 0:
        raise RuntimeError(f"ngs qc {' '.join(argv[2:])}: {r.stderr[-600:]}")
    marks, notes = [], {}
    for ln in r.stderr.splitlines():
        if ln.startswith("[ngs] gene model:") or ln.startswith("[ngs] reference:"):
            notes[ln[6:].split(":")[0]] = ln.split(":", 1)[1].strip()
        elif ln.startswith("[ngs]") and " ms " in ln:
            ms, what = ln[5:].split(" ms ", 1)
            marks.append((what.strip(), float(ms)))
    ph, prev = {}, 0.0
    for what, ms in marks[1:]:
        ph["until " + what] = round(ms - prev, 1)
        prev = ms
    ph["process start + exit (wall clock of the command - its own last milestone)"] = round(wall - (marks[-1][1] if marks else 0.0), 1)
    return wall, ph, notes


def leg_file_realistic(lib, host, ffi, args, ctx, tmp, ngs, write_s_per_record):
    """An aligner-style file (include/ngsq_shared.h NGSQ_SYNTH_FILE_REALISTIC) on the header a user has -- the 195 @SQ of the GRCh38
    no-alt analysis set at full length, the records over all of it, their bases sampled from the reference (round 6: chr1 + chr2
    and independent bases until then) -- through the same entry points: in process five times, once through `ngs qc --ingest
    host` for the document it must equal, and then THE COMMAND A USER RUNS: `ngs qc -r <the 3.1 GB soft-masked FASTA> -f <a 3.4 M-row
    GFF>` -- all seven facets, wall clock with phases -- beside the same command without -r / -f."""
    import ctypes as C
    import numpy as np
    from ngs_amd.genome_shape import grch38_no_alt
    names, lens, primary = grch38_no_alt()
    # (the writer is zlib level 6 on the host cores, ~1.1 M aligner-style records/s on the 16 these boxes grant: 150 M records -- a
    # file on which the GPU's time, not the pipeline's start, is what is measured -- take it ~140 s; a slower host gets a smaller file)
    nr = int(min(args.file_realistic_records, max(10_000_000, args.file_realistic_budget / max(write_s_per_record * 1.4, 1e-9))))
    path = os.path.join(tmp, "realistic.bam")
    cfg = host.synth_config(nr, read_len=args.read_len, genome=lens, file_style=ffi.SYNTH_FILE_REALISTIC, seq_model=ffi.SYNTH_SEQ_FROM_REFERENCE, lib=lib)
    name_arr = (C.c_char_p * len(names))(*[x.encode() for x in names])
    t0 = time.perf_counter()
    assert lib.ngsq_synth_write_bam_named(C.byref(cfg), name_arr, path.encode(), nr, args.file_level, 0) == 0, lib.ngsq_bam_last_error()
    tw = time.perf_counter() - t0
    os.sync()
    size = os.path.getsize(path)
    settled = settle_page_cache(path)
    gctx = host.QcContext(lens, primary, max_read_len=1024, gc_seed=GC_SEED, sorted_input=True, timing=True, lib=lib)
    try:
        times, best, timing, after, doc = scan_file_in_process(lib, host, ffi, gctx, path, nr, FILE_SCANS, names)
    finally:
        gctx.close()
    lib.ngsq_release_cached_memory()        # (the child processes below find the device as a user's command would)
    best_scan, best = best, median(times)   # everything below is quoted on the MEDIAN scan
    inf = timing.get("bgzf_inflate")
    raw = (inf["algo_bytes"] - size) if inf else None
    out = {"records": nr, "bam_bytes": size, "bam_write_s": round(tw, 1),
           "header": "195 @SQ: the GRCh38 no-alt analysis set at full length (3.1 Gbp), records on all of it, bases sampled from the reference",
           "style": "Illumina read names, NM MD MC AS XS MQ RG on every mapped record, SA / XA / a B,S array on some, "
                    "15 % CIGARs of 2-5 operations (clips, insertions, deletions), real mate positions",
           "compressed_bytes_per_record": round(size / nr, 1), "inflated_bytes_per_record": round(raw / nr, 1) if raw else None,
           "seconds_each_scan": times, "value": round(nr / best, 1), "unit": "records/s", "value_is": "median of the %d scans" % FILE_SCANS,
           "seconds_best_scan": round(best_scan, 3), "scan_spread_pct": round(100 * (max(times) / min(times) - 1), 1),
           "page_cache_settling_reads_GB_per_s": settled,
           "compressed_GB_per_s": round(size / best / 1e9, 2), "inflated_GB_per_s": round(raw / best / 1e9, 2) if raw else None,
           "records_per_s_after_first_batch": round(after[0], 1) if after else None,
           "kernels": kernel_table(timing), "check_total": doc["general"]["records"]["total"], "source": "page cache"}
    if inf and inf["launches"]:
        gbs = inf["algo_bytes"] / inf["total_ms"] / 1e6
        out["inflate"] = {"achieved_GB_per_s": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4),
                          "avg_launch_ms": round(inf["total_ms"] / inf["launches"], 3)}
    t0 = time.perf_counter()
    r = subprocess.run([ngs, "-q", "qc", path, "GRCh38_no_alt_AnalysisSet", "-o", tmp, "--ingest", "host"], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"ngs qc --ingest host: {r.stderr[-400:]}")
    out["cli_host_ingest_s"] = round(time.perf_counter() - t0, 2)
    with open(os.path.join(tmp, "realistic.bam.results.json")) as f:
        out["same_document_as_host_reader"] = json.dumps(json.load(f), sort_keys=True) == json.dumps(doc, sort_keys=True)
    # ---- the command a user runs: all seven facets from the files they have
    try:
        fa, gff = os.path.join(tmp, "GRCh38_stand_in.fa"), os.path.join(tmp, "annotation.gff3")
        t0 = time.perf_counter()
        fa_bytes = write_bench_fasta(np, lib, host, cfg, names, lens, fa)
        gff_bytes, model = write_bench_gff(np, names, lens, primary, gff)
        os.sync()
        for f_ in (fa, gff):
            settle_page_cache(f_)
        made_s = time.perf_counter() - t0
        base = [path, "GRCh38_no_alt_AnalysisSet"]
        runs = {"default": [], "all": []}
        phases = {}
        for rep in range(3):
            for key, extra in (("default", []), ("all", ["-r", fa, "-f", gff])):
                # (a command that has just left is still being torn down by the driver -- tens of GB of device memory it left to the
                # kernel -- and the next process's allocations wait for that: back to back the runs measured each other, 0.25-1.5 s
                # of "HIP initialisation")
                time.sleep(2.0)
                wall, ph, notes = run_cli_traced(ngs, base + extra, tmp)
                runs[key].append(round(wall, 1))
                phases[key] = (ph, notes)
                if key == "all":
                    with open(os.path.join(tmp, "realistic.bam.results.json")) as f:
                        cli_doc = json.load(f)
        # ... and returning when the document is on disk (NGSQ_RETURN_WHEN_DONE=1: the scan in a forked child whose teardown -- the
        # driver taking 17-40 GB of device memory apart -- goes on behind the command; opt-in, see ngs_main.cpp)
        early = {}
        for key, extra in (("default", []), ("all", ["-r", fa, "-f", gff])):
            ws = []
            for _ in range(3):
                time.sleep(2.0)
                ws.append(round(run_cli_traced(ngs, base + extra, tmp, {"NGSQ_RETURN_WHEN_DONE": "1"})[0], 1))
            early[key] = ws
        time.sleep(2.0)
        # the same seven facets inside this process (HIP up, FASTA through the same loader): the document the command must give
        t0 = time.perf_counter()
        actx = host.QcContext(lens, primary, facets=0x7F, max_read_len=1024, gc_seed=GC_SEED, sorted_input=True, timing=True, lib=lib,
                              ref_fasta=fa, ref_names=names)
        try:
            actx.set_features(*model)
            t_all, _, timing_all, _, doc_all = scan_file_in_process(lib, host, ffi, actx, path, nr, 1, names)
            ref_stats = actx.reference_wait()
        finally:
            actx.close()
        wall_default, wall_all = median(runs["default"]), median(runs["all"])
        ph_all, notes = phases["all"]
        scan_ms = ph_all.get("until records scanned", 0.0)
        out["cli_all_facets"] = {
            "command": "ngs qc realistic.bam GRCh38_no_alt_AnalysisSet -r <FASTA> -f <GFF>  (General, Template Length, GC Content, Quality Score, "
                       "Genomic Features, Coverage, Edits)",
            "fasta_bytes": fa_bytes, "fasta": "195 sequences, 60 bases per line, soft-masked (half of it lower case), no .fai",
            "gff_bytes": gff_bytes, "gff_rows": 3_400_000, "gff_intervals_kept": int(len(model[0])), "files_written_s": round(made_s, 1),
            "wall_ms": wall_all, "wall_ms_each_run": runs["all"], "records_per_s": round(nr / wall_all * 1e3, 1),
            "phases_ms": ph_all, "set_up": notes,
            "default_facets_same_file": {"wall_ms": wall_default, "wall_ms_each_run": runs["default"], "phases_ms": phases["default"][0],
                                         "records_per_s": round(nr / wall_default * 1e3, 1)},
            "extra_wall_for_edits_and_features_ms": round(wall_all - wall_default, 1),
            "returning_when_the_document_is_on_disk": {"env": "NGSQ_RETURN_WHEN_DONE=1", "wall_ms": median(early["all"]), "wall_ms_each_run": early["all"],
                                                       "default_facets_wall_ms": median(early["default"]), "default_facets_wall_ms_each_run": early["default"]},
            "in_process_all_facets": {"seconds": t_all[0], "records_per_s": round(nr / t_all[0], 1), "reference_load": {k: (round(v, 4) if isinstance(v, float) else v) for k, v in ref_stats.items()},
                                      "kernels": kernel_table(timing_all)},
            "same_document_as_in_process": json.dumps(cli_doc, sort_keys=True) == json.dumps(doc_all, sort_keys=True),
            "edits_reads": int(sum(cli_doc["edits"]["read_one_edits"]["values"]) + sum(cli_doc["edits"]["read_two_edits"]["values"])),
            "mean_edits_read_one": cli_doc["edits"]["summary"]["mean_edits_read_one"],
            "features_processed": cli_doc["features"]["records"]["processed"],
        }
    except Exception as e:  # noqa: BLE001 -- reported, never required
        out["cli_all_facets"] = {"failed": f"{type(e).__name__}: {e}"}
    return out


def leg_mixed(lib, host, ffi, np, args, device):
    """The 50-300 bp mixed-CIGAR workload (ragged SEQ / QUAL / CIGAR columns with offsets) on the same facets, beside the
    headline line: the same loop as main()'s, fewer steps."""
    import types
    n = args.mixed_records
    scfg = host.synth_config(n, mode=ffi.SYNTH_MIXED, read_len=args.read_len, max_len=args.mixed_max_len, ref_len=CHR1, n_refs=2)
    ctx = host.QcContext([CHR1, CHR2], [1, 1], facets=args.facets & 0x1F, device=device, max_read_len=args.mixed_max_len, gc_seed=GC_SEED,
                         timing=True, sorted_input=n / CHR1 <= 0.5, lib=lib)
    db = ctx.synth_device_batch(scfg, 0, n)
    try:
        def step():
            ctx.reset(); ctx.process_batch(db); ctx.finalize()
        for _ in range(3):
            step()
        ctx.synchronize()
        ctx.kernel_timing_reset()
        t0 = time.perf_counter()
        for _ in range(args.mixed_steps):
            step()
        ctx.synchronize()
        dt = time.perf_counter() - t0
        margs = types.SimpleNamespace(facets=args.facets & 0x1F, read_len=args.read_len, workload="mixed", mixed_max_len=args.mixed_max_len)
        parity = check_invariants(ctx, ffi, n, margs, True, False)
        timing = ctx.kernel_timing()
        algo_rec = 25.0 + 4.0 * db.cigar_ops / n + db.seq_bytes / n + db.qual_bytes / n
        out = {"workload": "%d M synthetic 50-%d bp reads, 1-5 CIGAR operations, ragged columns" % (n // 1_000_000, args.mixed_max_len),
               "value": round(n * args.mixed_steps / dt, 1), "unit": "records/s", "steps": args.mixed_steps,
               "ms_per_step": round(dt / args.mixed_steps * 1e3, 3), "algorithmic_bytes_per_record": round(algo_rec, 2),
               "hbm_frac_whole_pass": round(n * args.mixed_steps / dt * algo_rec / (HBM_PEAK_GBS * 1e9), 4),
               "parity_check": parity, "kernels": kernel_table(timing)}
        q = timing.get("qual")
        if q and q["launches"] and q["total_ms"] > 0:
            avg_ms = q["total_ms"] / q["launches"]
            achieved = (q["algo_bytes"] / q["launches"]) / (avg_ms * 1e-3) / 1e9
            traffic, src = pmc_traffic(n, margs)
            lm = getattr(args, "live_mixed", (None, None))
            measured = lm[0] is not None
            if measured:
                traffic, src = lm
            out["roofline"] = {"bound": "hbm", "kernel": "k_qual_ragged", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": src,
                               "traffic_measured_in_this_run": measured, "avg_launch_ms": round(avg_ms, 4),
                               "algo_bytes_per_launch": q["algo_bytes"] // q["launches"]}
        return out
    finally:
        ctx.free_batch(db)
        ctx.close()


# ---------------------------------------------------------------------------------------------
# Edits and Genomic Features (optional facets of the reference: -r FASTA, -f GFF)
# ---------------------------------------------------------------------------------------------
def synthetic_gene_model(np):
    """A GENCODE-shaped gene model on chr1/chr2: ~20 k genes per sequence, ~10 exons each, CDS inside exons, UTRs
    at the ends (columns of ngsq_features: sequence index, name id = role, GFF start, GFF end)."""
    rng = np.random.default_rng(0x47464633)
    ref, name, start, stop = [], [], [], []
    for r, L in enumerate((CHR1, CHR2)):
        n_genes = 20_000
        g0 = np.sort(rng.integers(1, L - 200_000, n_genes))
        glen = rng.integers(2_000, 150_000, n_genes)
        for a, ln in zip(g0.tolist(), glen.tolist()):
            ref.append(r); name.append(4); start.append(a); stop.append(a + ln)          # gene
            k = int(rng.integers(2, 18))
            cuts = np.sort(rng.integers(a, a + ln, 2 * k))
            for e in range(k):
                s, t = int(cuts[2 * e]), int(cuts[2 * e + 1])
                ref.append(r); name.append(3); start.append(s); stop.append(t)            # exon
                if 0 < e < k - 1:
                    ref.append(r); name.append(2); start.append(s); stop.append(t)        # CDS
            ref.append(r); name.append(0); start.append(int(cuts[0])); stop.append(int(cuts[1]))      # 5' UTR
            ref.append(r); name.append(1); start.append(int(cuts[-2])); stop.append(int(cuts[-1]))   # 3' UTR
    return (np.asarray(ref, dtype=np.uint32), np.asarray(name, dtype=np.uint32), np.asarray(start, dtype=np.uint32),
            np.asarray(stop, dtype=np.uint32))


def leg_extra_facets(lib, host, ffi, np, n=100_000_000):
    """Kernel times of the two optional facets on the first n records of the workload (reference bases of chr1
    and chr2 and a 400 k-interval gene model resident in HBM).  The reads of this leg are SAMPLED FROM the reference
    (ngsq_shared.h NGSQ_SYNTH_SEQ_FROM_REFERENCE: the reference's bases under every `M`, one substitution in 200) --
    what Edits sees on aligner output (edits.rs:276-291: a mismatch is rare); `edits_iid_reads` keeps round 3's input
    beside it, independent bases that differ from the reference three times in four."""
    scfg = host.synth_config(100_000_000, ref_len=CHR1, n_refs=2, seq_model=ffi.SYNTH_SEQ_FROM_REFERENCE)
    out = {"records": n, "reads": "sampled from the reference, 0.5 % substitutions"}
    bases = [host.synth_reference(scfg, r, L, lib) for r, L in enumerate((CHR1, CHR2))]
    ctx = host.QcContext([CHR1, CHR2], [1, 1], facets=ffi.FACET_EDITS | ffi.FACET_FEATURES, max_read_len=150, gc_seed=GC_SEED,
                         timing=True, ref_bases=bases, lib=lib)
    del bases
    try:
        ctx.set_features(*synthetic_gene_model(np))
        db = ctx.synth_device_batch(scfg, 0, n)
        for rep in range(2):
            ctx.reset()
            ctx.kernel_timing_reset()
            ctx.process_batch(db)
            ctx.finalize()
        t = ctx.kernel_timing()
        f = ctx.features()
        out["kernels"] = kernel_table(t)
        for k in ("edits", "edits_vaf", "features"):
            if k in out["kernels"]:
                out["kernels"][k]["hbm_frac"] = round(out["kernels"][k]["GBps"] / HBM_PEAK_GBS, 4)
        out["features_processed"] = f["processed"]
        r1, r2, vaf = ctx.edits()
        out["edits_reads"] = int(r1.sum() + r2.sum())
        tot = np.arange(r1.size, dtype=np.float64)
        out["mean_edits_per_read"] = round(float(((r1 + r2) * tot).sum() / max(1, out["edits_reads"])), 4)
        out["vaf_positions"] = int(vaf.sum())
        ctx.free_batch(db)
        # the same reads with an aligner's CIGARs: 9 % soft-clipped, 3 % with an insertion, 3 % with a deletion (the 6 % with
        # an insertion or a deletion are walked operation by operation, beside the fast path's 94 %)
        acfg = host.synth_config(100_000_000, ref_len=CHR1, n_refs=2, seq_model=ffi.SYNTH_SEQ_FROM_REFERENCE, file_style=ffi.SYNTH_FILE_CIGAR_MIX)
        db = ctx.synth_device_batch(acfg, 0, n)
        for rep in range(2):
            ctx.reset()
            ctx.kernel_timing_reset()
            ctx.process_batch(db)
            ctx.finalize()
        ka = kernel_table(ctx.kernel_timing()).get("edits")
        if ka:
            ka["hbm_frac"] = round(ka["GBps"] / HBM_PEAK_GBS, 4)
            r1, r2, _ = ctx.edits()
            ka["mean_edits_per_read"] = round(float(((r1 + r2) * tot).sum() / max(1, int(r1.sum() + r2.sum()))), 4)
            ka["cigars"] = "85 % 150M, 9 % soft-clipped at one end, 3 % an insertion, 3 % a deletion (1-8 bases)"
            out["edits_aligner_cigars"] = ka
        ctx.free_batch(db)
        # reads that DIFFER from the reference (edits.rs:276-291 costs the same per base whatever the bases are): 5 % and 25 % of the
        # compared bases substituted -- bisulfite-converted, cross-species, noisy reads -- and round 3's input, independent random
        # bases (three in four differ)
        # ... and the 50-300 base reads of `mixed` (the offsets layout: since round 5 through the same window lanes, a byte map in LDS
        # saying which record a window belongs to; 10 % M (I|D) M, 5 % M (N of 100-5000) M)
        for key, model, what, mode in (("edits_subst_5pct", ffi.synth_seq_subst(0.05), "sampled from the reference, 5 % substitutions", ffi.SYNTH_FIXED),
                                       ("edits_subst_25pct", ffi.synth_seq_subst(0.25), "sampled from the reference, 25 % substitutions", ffi.SYNTH_FIXED),
                                       ("edits_iid_reads", ffi.SYNTH_SEQ_IID, "independent random bases", ffi.SYNTH_FIXED),
                                       ("edits_mixed_reads", ffi.SYNTH_SEQ_FROM_REFERENCE, "50-300 bases, 1-3 CIGAR operations, offsets layout; sampled from the reference, 0.5 % substitutions", ffi.SYNTH_MIXED),
                                       ("edits_mixed_subst_5pct", ffi.synth_seq_subst(0.05), "50-300 bases, 1-3 CIGAR operations, offsets layout; 5 % substitutions", ffi.SYNTH_MIXED)):
            icfg = host.synth_config(100_000_000, mode=mode, ref_len=CHR1, n_refs=2, seq_model=model)
            db = ctx.synth_device_batch(icfg, 0, n)
            for rep in range(2):
                ctx.reset()
                ctx.kernel_timing_reset()
                ctx.process_batch(db)
                ctx.finalize()
            ki = kernel_table(ctx.kernel_timing()).get("edits")
            if ki:
                ki["hbm_frac"] = round(ki["GBps"] / HBM_PEAK_GBS, 4)
                r1, r2, _ = ctx.edits()
                ki["mean_edits_per_read"] = round(float(((r1 + r2) * tot).sum() / max(1, int(r1.sum() + r2.sum()))), 2)
                ki["reads"] = what
                out[key] = ki
            ctx.free_batch(db)
        return out
    finally:
        ctx.close()


def leg_all_facets(lib, host, ffi, np, args, device):
    """ALL seven facets as ONE scan (get_qc_facets with -r and -f: qc.rs:44-126 builds General, Template Length, GC Content, Quality
    Score, Genomic Features, Coverage and Edits for one pass pair): reads sampled from the synthetic reference (0.5 % substitutions),
    the reference's bases and the 400 k-interval gene model resident in HBM, Coverage streamed.  ms per step, the fraction of the
    HBM roofline on the pass's algorithmic bytes, the kernels, and the full-size invariants of every facet."""
    import types
    n = args.all_facets_records
    scfg = host.synth_config(n, read_len=args.read_len, ref_len=CHR1, n_refs=2, seq_model=ffi.SYNTH_SEQ_FROM_REFERENCE)
    bases = [host.synth_reference(scfg, r, L, lib) for r, L in enumerate((CHR1, CHR2))]
    ctx = host.QcContext([CHR1, CHR2], [1, 1], facets=0x7F, device=device, max_read_len=args.read_len, gc_seed=GC_SEED, timing=True,
                         sorted_input=n / CHR1 <= 0.5, ref_bases=bases, lib=lib)
    del bases
    db = None
    try:
        ctx.set_features(*synthetic_gene_model(np))
        db = ctx.synth_device_batch(scfg, 0, n)

        def step():
            ctx.reset(); ctx.process_batch(db); ctx.finalize()
        for _ in range(3):
            step()
        ctx.synchronize()
        ctx.kernel_timing_reset()
        loops = []
        for _ in range(3):
            ctx.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.all_facets_steps):
                step()
            ctx.synchronize()
            loops.append((time.perf_counter() - t0) / args.all_facets_steps * 1e3)
        ms = median(loops)
        fargs = types.SimpleNamespace(facets=0x1F, read_len=args.read_len, workload="fixed", mixed_max_len=args.mixed_max_len)
        parity = check_invariants(ctx, ffi, n, fargs, False, False)
        bad = []
        g, f = ctx.general(), ctx.features()
        r1, r2, vaf = ctx.edits()
        reads = int(r1.sum() + r2.sum())
        if not (n - g["unmapped"] - g["duplicate"] <= reads <= n - max(g["unmapped"], g["duplicate"])):
            bad.append("Edits counts the mapped, non-duplicate reads")
        if not (0 < int(vaf.sum()) <= CHR1 + CHR2):
            bad.append("VAF histogram counts covered positions")
        if f["processed"] + f["ignored_flags"] + f["ignored_nonprimary_chromosome"] != n:
            bad.append("Genomic Features conserves records")
        if bad:
            parity = ("FAILED: " if parity.startswith("ok") else parity + "; ") + "; ".join(bad)
        elif parity.startswith("ok"):
            parity = "ok: %d full-size invariants" % (int(parity.split()[1]) + 3)
        timing = ctx.kernel_timing()
        # algorithmic bytes of the pass: the record's own 254 bytes (every column once) + the teardown scans; the packed reference
        # bytes Edits compares with (75 per read, mostly served by the L2: neighbouring reads overlap) are NOT counted
        algo_rec = 25.0 + 4.0 * db.cigar_ops / n + db.seq_bytes / n + db.qual_bytes / n
        kt = kernel_table(timing)
        return {"workload": "%d M synthetic %d bp reads sampled from the reference, facets 0x7F (all seven), chr1 + chr2 bases and a 400 k-interval "
                            "gene model resident in HBM" % (n // 1_000_000, args.read_len),
                "ms_per_step": round(ms, 3), "ms_per_step_each_loop": [round(x, 3) for x in loops], "steps": args.all_facets_steps,
                "value": round(n / ms * 1e3, 1), "unit": "records/s", "algorithmic_bytes_per_record": round(algo_rec, 2),
                "hbm_frac_whole_pass": round(n / ms * 1e3 * algo_rec / (HBM_PEAK_GBS * 1e9), 4),
                "seq_column_reads": "twice (k_gc and k_edits_rows)" if "gc" in kt else "once (the GC window is tallied by k_edits_rows)",
                "ms_per_step_outside_kernels": round(ms - sum(v["avg_ms"] * timing[k]["launches"] / max(1, timing["qual"]["launches"]) for k, v in kt.items()), 3),
                "parity_check": parity, "kernels": kt}
    finally:
        if db is not None:
            ctx.free_batch(db)
        ctx.close()


def leg_whole_genome(lib, host, ffi, np, args, device):
    """The header `ngs qc` meets in practice -- the 195 @SQ of the GRCh38 no-alt analysis set at their real lengths (3.1 Gbp; the
    reference loops over every one of them: command.rs:258-272,356) -- with the records spread over all of it (GENOME mode of the
    synthetic generator).  Until round 6 every leg had chr1 + chr2.  Reported: what 3.1 Gbp of Coverage and Edits state cost to
    create, reset and tear down, ms per pass of the default facets (streamed Coverage and depth arrays) and of all seven, the
    launches of one finalize, and full-size invariants in place of the oracle."""
    from ngs_amd.genome_shape import grch38_no_alt
    names, lens, primary = grch38_no_alt()
    n = args.whole_genome_records
    G = int(sum(lens))
    scfg = host.synth_config(n, read_len=args.read_len, genome=lens, seq_model=ffi.SYNTH_SEQ_FROM_REFERENCE, lib=lib)
    out = {"workload": "%d M synthetic %d bp reads over the 195 sequences of the GRCh38 no-alt analysis set (%.2f Gbp; the 169 contigs' "
                       "lengths are stand-ins in the real range: ngs_amd/genome_shape.py)" % (n // 1_000_000, args.read_len, G / 1e9),
           "sequences": len(lens), "primary": int(sum(primary)), "depth": round(n * args.read_len / G, 2)}
    seen_primary = None
    db = None
    keep = None   # the context the batch's columns live in
    try:
        for label, kw in (("default_streamed", dict(facets=0x1F, sorted_input=True)), ("default_arrays", dict(facets=0x1F, sorted_input=False)),
                          ("all_seven_streamed", dict(facets=0x7F, sorted_input=True))):
            row = {}
            t0 = time.perf_counter()
            bases = None
            if kw["facets"] & ffi.FACET_EDITS:
                bases = [host.synth_reference(scfg, r, L, lib) for r, L in enumerate(lens)]
                row["reference_generated_s"] = round(time.perf_counter() - t0, 2)
                t0 = time.perf_counter()
            ctx = host.QcContext(lens, primary, device=device, max_read_len=args.read_len, gc_seed=GC_SEED, timing=True, ref_bases=bases, lib=lib, **kw)
            ctx.synchronize()
            row["create_ms"] = round((time.perf_counter() - t0) * 1e3, 1)
            del bases
            try:
                if kw["facets"] & ffi.FACET_FEATURES:
                    rng = np.random.default_rng(7)
                    m = 3_000_000   # a GENCODE-sized model: ~3 M rows of the five feature types
                    fr = rng.integers(0, 24, m).astype(np.uint32)
                    fs = (rng.random(m) * (np.array(lens, dtype=np.float64)[fr] - 5000)).astype(np.uint32) + 1
                    ctx.set_features(fr, rng.choice(5, m, p=[.05, .05, .3, .5, .1]).astype(np.uint32), fs, fs + rng.integers(0, 4000, m).astype(np.uint32))
                if db is None:
                    t0 = time.perf_counter()
                    db = ctx.synth_device_batch(scfg, 0, n)
                    keep = ctx
                    out["generate_s"] = round(time.perf_counter() - t0, 2)

                def timed(fn):
                    ctx.synchronize()
                    t = time.perf_counter()
                    fn()
                    ctx.synchronize()
                    return (time.perf_counter() - t) * 1e3
                for _ in range(2):
                    ctx.reset(); ctx.process_batch(db); ctx.finalize()
                ctx.kernel_timing_reset()
                steps = args.whole_genome_steps
                parts = {"reset": [], "process": [], "finalize": []}
                for _ in range(steps):
                    parts["reset"].append(timed(ctx.reset))
                    parts["process"].append(timed(lambda: ctx.process_batch(db)))
                    parts["finalize"].append(timed(ctx.finalize))
                timing = ctx.kernel_timing()
                row["ms_per_pass"] = round(sum(median(v) for v in parts.values()), 3)
                row["of_which_ms"] = {k: round(median(v), 3) for k, v in parts.items()}
                row["records_per_s"] = round(n / row["ms_per_pass"] * 1e3, 1)
                row["kernels"] = kernel_table(timing)
                row["launches_per_pass"] = {k: round(v["launches"] / steps, 1) for k, v in timing.items() if v["launches"]}
                # ---- invariants
                bad = []
                g = ctx.general()
                if g["total"] != n:
                    bad.append("general.total == records")
                cov_seen, pos_sum, depth_sum = [], 0, 0
                for r in range(len(lens)):
                    seen, hist, ign, bins = ctx.coverage_sequence(r)
                    if seen and not primary[r]:
                        bad.append("a sequence outside the primary assembly has a Coverage entry")
                    if seen:
                        cov_seen.append(r)
                        pos_sum += int(hist.sum()) + ign - (lens[r] + 1)
                        depth_sum += int(bins.sum())
                if pos_sum != 0:
                    bad.append("every covered sequence's depth histogram counts its L+1 positions")
                if seen_primary is None:
                    seen_primary = cov_seen
                elif cov_seen != seen_primary:
                    bad.append("the same sequences have entries on both Coverage paths")
                if ctx.coverage_nonsensical() != 0:
                    bad.append("no read crosses the end of its sequence")
                row["coverage_entries"] = len(cov_seen)
                row["depth_total"] = depth_sum
                if kw["facets"] & ffi.FACET_EDITS:
                    r1, r2, vaf = ctx.edits()
                    reads = int(r1.sum() + r2.sum())
                    if not (n - g["unmapped"] - g["duplicate"] <= reads <= n - max(g["unmapped"], g["duplicate"])):
                        bad.append("Edits counts the mapped, non-duplicate reads")
                    if int(r1[:8].sum() + r2[:8].sum()) < 0.99 * reads:
                        bad.append("reads sampled from the reference have a handful of edits at most (every sequence's bases are the right ones)")
                    f = ctx.features()
                    if f["processed"] + f["ignored_flags"] + f["ignored_nonprimary_chromosome"] != n:
                        bad.append("Genomic Features conserves records")
                    row["state_GB"] = round((12 * G) / 1e9, 1)   # u32 depth + 2 x u32 Edits arrays per position
                row["parity_check"] = "ok" if not bad else "FAILED: " + "; ".join(bad)
            finally:
                if ctx is not keep:
                    t0 = time.perf_counter()
                    ctx.close()
                    row["destroy_ms"] = round((time.perf_counter() - t0) * 1e3, 1)
            out[label] = row
        d0, d1 = out["default_streamed"].get("depth_total"), out["default_arrays"].get("depth_total")
        out["streamed_equals_arrays"] = d0 == d1 and d0 is not None
        out["parity_check"] = "ok" if all(out[k].get("parity_check") == "ok" for k in ("default_streamed", "default_arrays", "all_seven_streamed")) and out["streamed_equals_arrays"] else "FAILED"
        return out
    finally:
        if keep is not None:
            if db is not None:
                keep.free_batch(db)
            keep.close()


def write_sharded_bam(lib, host, args, world, tmp, bam):
    """The synthetic BAM of the N > 1 file legs: as many records as the writer (zlib on the host cores) manages inside
    --file-write-budget seconds (probed), at most N x --file-records and a third of the free space."""
    import ctypes as C
    probe_n = 2_000_000
    pcfg = host.synth_config(probe_n, read_len=args.read_len, ref_len=CHR1, n_refs=2)
    t0 = time.perf_counter()
    assert lib.ngsq_synth_write_bam(C.byref(pcfg), bam.encode(), probe_n, args.file_level, 0) == 0, lib.ngsq_bam_last_error()
    rate = probe_n / max(time.perf_counter() - t0, 1e-6)
    n = int(min(world * args.file_records, max(world * 5_000_000, rate * args.file_write_budget)))
    try:    # and no more than a third of the free space under the temporary directory (~100 bytes per record)
        import shutil
        n = int(min(n, max(world * 1_000_000, shutil.disk_usage(tmp).free / 3 / 105)))
    except OSError:
        pass
    fcfg = host.synth_config(n, read_len=args.read_len, ref_len=CHR1, n_refs=2)
    t0 = time.perf_counter()
    assert lib.ngsq_synth_write_bam(C.byref(fcfg), bam.encode(), n, args.file_level, 0) == 0, lib.ngsq_bam_last_error()
    tw = time.perf_counter() - t0
    os.sync()
    settle_page_cache(bam)
    return n, tw


def leg_file_sharded_in_process(lib, host, ffi, np, args, comm, rank, world, device):
    """N > 1, EVERY rank: rank 0 writes one synthetic BAM (the others wait for a marker file -- no collective is left
    waiting while zlib runs for a minute); then each rank streams its BGZF block range of it through the chunked
    pipeline (ngs_amd/shard.py: Comm.scan_file_shard = ngsq_bam_shard_open / next_batch_device / process_batch /
    ngsq_bam_shard_verify), one ngsq_exchange, ngsq_finalize -- bracketed by barriers, so the time is the slowest
    rank's.  Returns {"tmp", "bam", "records", "bam_write_s", "in_process": {...}} (rank 0 keeps the file for the
    `ngs qc --gpus N` leg); every rank returns the same verdict."""
    import tempfile
    # the path: rank 0's choice, learnt by everyone in one small all-gather
    buf = np.zeros(512, dtype=np.uint8)
    tmp = None
    if rank == 0:
        tmp = tempfile.mkdtemp(prefix="ngsq_bench_", dir=os.environ.get("TMPDIR", "/tmp"))
        raw = tmp.encode()
        buf[:len(raw)] = np.frombuffer(raw, dtype=np.uint8)
    tmp = bytes(comm.allgather(buf)[0]).rstrip(b"\0").decode()
    bam, marker = os.path.join(tmp, "synth.bam"), os.path.join(tmp, "written.json")
    if rank == 0:
        try:
            n, tw = write_sharded_bam(lib, host, args, world, tmp, bam)
            note = {"ok": True, "records": n, "bam_write_s": round(tw, 2), "bam_bytes": os.path.getsize(bam)}
        except Exception as e:  # noqa: BLE001 -- the others must hear of it
            note = {"ok": False, "why": f"{type(e).__name__}: {e}"}
        with open(marker + ".tmp", "w") as f:
            json.dump(note, f)
        os.rename(marker + ".tmp", marker)
    else:
        deadline = time.time() + 4 * args.file_write_budget + 180
        while not os.path.exists(marker):
            if time.time() > deadline:
                raise RuntimeError("rank 0 did not write the shared BAM in time")
            time.sleep(0.05)
        with open(marker) as f:
            note = json.load(f)
    out = {"tmp": tmp, "bam": bam, "records": note.get("records", 0), "bam_write_s": note.get("bam_write_s"),
           "bam_bytes": note.get("bam_bytes")}
    if not note["ok"]:
        out["in_process"] = {"failed": note["why"]}
        return out
    n = note["records"]
    # (every rank says whether it is ready before the first collective of the scan: one that is not must not leave the
    # others waiting inside it)
    ctx, why = None, ""
    try:
        ctx = host.QcContext([CHR1, CHR2], [1, 1], device=device, max_read_len=max(256, args.read_len), gc_seed=GC_SEED, sorted_input=True,
                             timing=False, lib=lib)
    except Exception as e:  # noqa: BLE001
        why = f"{type(e).__name__}: {e}"
    ready = [r[0] for r in comm.allgather_ints([1 if ctx is not None else 0])]
    if not all(ready):
        if ctx is not None:
            ctx.close()
        out["in_process"] = {"failed": why or "rank(s) %s could not create a context" % [i for i, r in enumerate(ready) if not r]}
        return out
    # reader threads per rank: the host cores this launch may use, less two per rank for the scan loops, shared out
    # (what `ngs qc --gpus N` gives its workers)
    reader_threads = max(2, (effective_cores() - 2 * world) // world)
    try:
        times, mine, rounds = [], 0, 0
        for rep in range(4):   # the first scan of a process allocates and pins the pipeline's buffers; the median of the other three counts
            ctx.reset()
            comm.barrier()
            t0 = time.perf_counter()
            info, rounds, mine = comm.scan_file_shard(ctx, bam, batch_records=1 << 22, threads=reader_threads)
            comm.exchange(ctx)
            ctx.finalize()
            comm.barrier()
            times.append(round(time.perf_counter() - t0, 3))
        total = ctx.results(["chr1", "chr2"])["general"]["records"]["total"]
        per_rank = [c[0] for c in comm.allgather_ints([mine])]
        best = median(times[1:])
        out["in_process"] = {"gpus": world, "seconds_each_scan": times, "value_is": "median of scans 1..3", "records_per_s": round(n / best, 1),
                             "compressed_GB_per_s": round(note["bam_bytes"] / best / 1e9, 2),
                             "records_per_s_per_gpu": round(n / best / world, 1), "records_per_rank": per_rank,
                             "rescans": rounds, "check_total": total, "reader_threads_per_rank": reader_threads,
                             "includes": "every rank: file open, reads of its BGZF block range, H2D, inflate, parse, all default "
                                         "facets, the boundary check, ngsq_exchange, finalize, close; slowest rank (barriers)"}
        if rank == 0:
            out["in_process"]["document"] = ctx.results(["chr1", "chr2"])
        return out
    finally:
        ctx.close()


def leg_file_sharded(lib, host, ffi, args, world, comm_kind, shared=None):
    """N > 1: ONE synthetic BGZF BAM of (up to) N x --file-records records scanned by `ngs qc --gpus N` -- one worker
    process per GPU, each streaming its BGZF block range through the chunked pipeline, one exchange before the teardown
    -- wall clock of the command (process starts and HIP initialisation included), and the document against the one
    `ngs qc` writes for the same file on one GPU."""
    import tempfile
    from ngs_amd import build
    reuse = isinstance(shared, dict) and shared.get("records") and os.path.exists(shared.get("bam", ""))
    tmp = shared["tmp"] if reuse else tempfile.mkdtemp(prefix="ngsq_bench_", dir=os.environ.get("TMPDIR", "/tmp"))
    bam = os.path.join(tmp, "synth.bam")
    out = {"zlib_level": args.file_level, "host_cores": effective_cores(), "source": "page cache", "filesystem": fs_type_of(tmp)}
    try:
        if reuse:   # the file the ranks have just scanned in process
            n = shared["records"]
            out["bam_write_s"] = shared["bam_write_s"]
        else:
            n, tw = write_sharded_bam(lib, host, args, world, tmp, bam)
            out["bam_write_s"] = round(tw, 2)
        out["records"] = n
        out["records_wanted"] = world * args.file_records
        out["bam_bytes"] = os.path.getsize(bam)
        inproc = dict(shared.get("in_process", {})) if isinstance(shared, dict) else {}
        inproc_doc = inproc.pop("document", None)
        if inproc:
            out["in_process"] = inproc
        ngs = build.build_cli(verbose=False)
        flags = ["--same-device"] if args.same_gpu else []
        if comm_kind == "shm" or args.transport == "shm":
            flags += ["--transport", "shm"]
        docs = {}
        for label, gp, runs in (("sharded", world, 3), ("one_gpu", 1, 1)):
            best, each = None, []
            for _ in range(runs):
                # (stderr into a file, not a pipe: the command returns when its workers have reported, while they are still being
                # torn down by the kernel -- a pipe would keep this process reading until the last of them is gone)
                errp = os.path.join(tmp, "ngs.stderr")
                with open(errp, "w") as ef:
                    t0 = time.perf_counter()
                    r = subprocess.run([ngs, "qc", bam, "GRCh38_no_alt_AnalysisSet", "-o", tmp] + (["--gpus", str(gp)] + flags if gp > 1 else []),
                                       stdout=subprocess.DEVNULL, stderr=ef)
                    dt = time.perf_counter() - t0   # (the command returns when its last worker has exited: round 4)
                with open(errp) as ef:
                    err_text = ef.read()
                if r.returncode != 0:
                    raise RuntimeError(f"ngs qc --gpus {gp}: {err_text[-600:]}")
                each.append(round(dt, 3))
            best = median(each)
            with open(os.path.join(tmp, "synth.bam.results.json")) as f:
                docs[label] = json.load(f)
            out[label] = {"gpus": gp, "seconds": round(best, 3), "seconds_each_run": each, "value_is": "median run", "records_per_s": round(n / best, 1),
                          "compressed_GB_per_s": round(out["bam_bytes"] / best / 1e9, 2)}
            if gp > 1:
                # the opt-in of callers that only want the document (NGSQ_RETURN_WHEN_DONE=1: the command returns when every
                # worker has written its part, while the kernel still unmaps their memory): beside the headline, never it
                with open(errp, "w") as ef:
                    t0 = time.perf_counter()
                    r2 = subprocess.run([ngs, "qc", bam, "GRCh38_no_alt_AnalysisSet", "-o", tmp, "--gpus", str(gp)] + flags,
                                        stdout=subprocess.DEVNULL, stderr=ef, env=dict(os.environ, NGSQ_RETURN_WHEN_DONE="1"))
                    dt2 = time.perf_counter() - t0
                time.sleep(0.5)  # (those workers' memory goes back to the driver: not part of the next run)
                if r2.returncode == 0:
                    out[label]["seconds_until_document_NGSQ_RETURN_WHEN_DONE"] = round(dt2, 3)
                out[label]["records_per_s_per_worker"] = round(n / best / gp, 1)
                out[label]["transport"] = [ln for ln in err_text.splitlines() if "exchange over" in ln][:1]
        out["value"] = out["sharded"]["records_per_s"]
        out["unit"] = "records/s"
        out["includes"] = "launcher + N worker process starts, HIP initialisation, header + index checks, the exchange, JSON write"
        out["json_equal_sharded_one_gpu"] = json.dumps(docs["sharded"], sort_keys=True) == json.dumps(docs["one_gpu"], sort_keys=True)
        if inproc_doc is not None:
            out["in_process"]["json_equal_one_gpu"] = json.dumps(inproc_doc, sort_keys=True) == json.dumps(docs["one_gpu"], sort_keys=True)
        out["check_total"] = docs["sharded"]["general"]["records"]["total"]
        return out
    finally:
        for f in os.listdir(tmp):
            os.remove(os.path.join(tmp, f))
        os.rmdir(tmp)


def _leave(code: int) -> None:
    """A rank whose ncclCommInitRank never returned (ngsq_comm_create_rccl gave up after NGSQ_RCCL_INIT_TIMEOUT_S and the
    ranks agreed on the shared-memory transport) still has a thread inside RCCL: the interpreter's and the runtime's exit
    handlers may wait for it.  Everything has been printed: leave without them."""
    try:
        from ngs_amd import ffi
        stuck = ffi.load_library().ngsq_comm_rccl_stuck()
    except Exception:  # noqa: BLE001 -- the launcher process never loads the library
        stuck = 0
    sys.stdout.flush()
    sys.stderr.flush()
    if stuck:
        os._exit(code)
    sys.exit(code)


if __name__ == "__main__":
    _leave(main())
