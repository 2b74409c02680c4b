#!/usr/bin/env python3
"""bench.py -- `ngs qc` record-scanning hot path on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2], the configuration the metric is quoted on
that fits one GPU): per GPU 100 M synthetic 150 bp reads resident in HBM as SoA
columns, coordinate-sorted over a chr1-sized reference (L = 248 956 422), ALL
default QC facets (General, Template Length, GC Content, Quality Score,
Coverage).  One step = one full pass of the hot path over the resident shard:
reset -> the facet kernels over every record -> (N > 1: RCCL sum of the shard
states) -> coverage teardown scan -> integer results on the host.

Prints ONE JSON line on rank 0 (contract in the task statement) with the
`roofline` of the dominant kernel (Quality Score: 150 of the 254 algorithmic
bytes per record) and the `cpu_baseline` (the C oracle, 1 core, bounded sample).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CHR1 = 248_956_422
CHR2 = 242_193_529
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def algorithmic_bytes_per_record(read_len: int, n_ops: float) -> float:
    """25 B fixed + 4 B per CIGAR op + ceil(l/2) SEQ + l QUAL."""
    return 25.0 + 4.0 * n_ops + (read_len + 1) // 2 + read_len


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--records", type=int, default=100_000_000, help="records per GPU")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--workload", choices=["fixed", "mixed"], default="fixed")
    ap.add_argument("--mixed-max-len", type=int, default=300, help="longest read of --workload mixed (50..N bp)")
    ap.add_argument("--cpu-sample", type=int, default=10_000_000,
                    help="records of the workload timed on the CPU oracle (0 = skip)")
    ap.add_argument("--no-timing", action="store_true", help="no per-kernel HIP event brackets")
    ap.add_argument("--force-dist", action="store_true",
                    help="take the torch.distributed/RCCL path even with one rank (plumbing check)")
    ap.add_argument("--coverage", choices=["auto", "stream", "array"], default="auto",
                    help="stream: sorted_input context, Coverage finishes positions while the sorted records stream by; "
                         "array: difference arrays + teardown scan (any record order); auto: stream up to 0.5 records per "
                         "reference position in the whole file (whole-genome depths: measured faster there), array above "
                         "(the weak-scaling runs pile N x 100 M reads on chr1: DESIGN.md section 5.4)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend of the N > 1 exchange (nccl = RCCL; "
                    "gloo stages the small collectives through the host: test boxes)")
    ap.add_argument("--same-gpu", action="store_true", help="all ranks on GPU 0 (boxes with one GPU; needs --backend gloo)")
    ap.add_argument("--emulate-shard", default="",
                    help="R/W: on ONE GPU, scan the records shard R of a W-GPU run would scan (the W x --records file's "
                         "slice, W times the depth, head guard as for rank R) -- kernel cost of a shard without the exchange")
    ap.add_argument("--facets", type=lambda x: int(x, 0), default=0x1F,
                    help="facet mask (default 0x1F = all default facets; 0x0E = BASELINE configs[1])")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            print(f"bench.py: --gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks", file=sys.stderr)
            return 2
        args.gpus = world

    from ngs_amd import build, ffi, host

    if rank == 0:
        build.build(verbose=False)
    lib = None
    dist = None
    torch = None
    use_dist = world > 1 or args.force_dist
    if use_dist:
        import torch  # noqa: F811
        import torch.distributed as dist  # noqa: F811
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if args.same_gpu:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
        dist.barrier()
    lib = ffi.load_library()
    if lib.ngsq_device_count() < 1:
        print("bench.py: no HIP device visible; the hot path has no CPU fallback", file=sys.stderr)
        return 3

    n = args.records
    mixed = args.workload == "mixed"
    max_len = args.mixed_max_len if mixed else args.read_len
    # the whole synthetic file has world * n records; this rank owns the contiguous
    # record range [rank*n, (rank+1)*n) = a contiguous BGZF block range of a sorted BAM
    # (weak scaling: ngs_amd.shard.shard_range(n * world, rank, world) == (rank * n, n))
    emu_rank, emu_world = (int(x) for x in args.emulate_shard.split("/")) if args.emulate_shard else (rank, world)
    scfg = host.synth_config(n * emu_world, mode=ffi.SYNTH_MIXED if mixed else ffi.SYNTH_FIXED,
                             read_len=args.read_len, max_len=args.mixed_max_len, ref_len=CHR1, n_refs=2)
    if args.coverage == "auto":
        args.coverage = "stream" if emu_world * n / CHR1 <= 0.5 else "array"
    ctx = host.QcContext([CHR1, CHR2], [1, 1], facets=args.facets, device=local_rank,
                         max_read_len=max_len, gc_seed=0x4E4753, timing=not args.no_timing,
                         sorted_input=args.coverage == "stream",
                         # shards behind the first: positions a read of the shard in front may still cover
                         # (the synthetic reads span at most 5.3 kb; real files: tools/qc_sharded.py keeps 1 Mi)
                         cov_head_guard=(1 << 16) if emu_rank > 0 else 0, lib=lib)
    t_gen = time.perf_counter()
    db = ctx.synth_device_batch(scfg, emu_rank * n, n)
    t_gen = time.perf_counter() - t_gen

    views = None
    if use_dist:
        from ngs_amd import shard
        views = shard.device_views(ctx, torch, local_rank)

    def sync():
        ctx.synchronize()
        if use_dist:
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    def step():
        ctx.reset()
        ctx.process_batch(db)
        if use_dist:
            # every facet state is an integer sum over records -> one RCCL
            # sum of the packed counter block and of the coverage difference arrays
            # (ngs_amd/shard.py: counters all-reduced; coverage by owner-computes halo exchange)
            shard.owner_teardown(ctx, dist, torch, views, coll_device=None if args.backend == "nccl" else "cpu")
        ctx.finalize()

    for _ in range(args.warmup):
        step()
    sync()
    ctx.kernel_timing_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    total_records = n * world
    ok = True
    if args.facets & ffi.FACET_GENERAL:
        ok = ctx.general()["total"] == total_records
    elif args.facets & ffi.FACET_QUALITY_SCORE:
        ok = int(ctx.quality_scores()[0].sum()) == total_records
    timing = ctx.kernel_timing()

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
            roofline = {"bound": "hbm", "kernel": "k_qual_perm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                        "avg_launch_ms": round(avg_ms, 4),
                        "algo_bytes_per_launch": q["algo_bytes"] // q["launches"]}
        kernels = {k: {"avg_ms": round(v["total_ms"] / v["launches"], 4),
                       "GBps": round(v["algo_bytes"] / max(v["total_ms"], 1e-9) / 1e6, 1)}
                   for k, v in timing.items() if v["launches"] and v["total_ms"] > 0}
        if roofline is not None:
            roofline["traffic"], roofline["traffic_source"] = pmc_traffic(n, args)
        cpu = None
        if world == 1 and args.cpu_sample > 0:
            cpu = cpu_baseline(lib, host, ffi, scfg, min(args.cpu_sample, n), max_len)
        out = {
            "metric": "BAM records/sec (whole node), all qc facets, 150 bp reads",
            "value": round(value, 1), "unit": "records/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8/u32 integer (u64 accumulators)",
            "data": "synthetic",
            "config": {"workload": ("BASELINE configs[2]: %d M synthetic %s reads per GPU resident in HBM, "
                                    "all default facets incl. CIGAR coverage over chr1 (L=248956422)"
                                    % (n // 1_000_000, "50-300 bp mixed-CIGAR" if mixed else f"{args.read_len} bp")),
                       "records_per_gpu": n, "read_len": max_len if mixed else args.read_len,
                       "facets": ",".join(n for b_, n in ((1, "General"), (2, "Template Length"), (4, "GC Content"),
                                                           (8, "Quality Score"), (16, "Coverage")) if args.facets & b_),
                       "sharding": ("contiguous record (BGZF block) ranges; RCCL all-reduce of counters, owner-computes "
                                    "coverage teardown with halo exchange") if world > 1 else "single GPU",
                       "coverage": ("streamed from the coordinate-sorted records (sorted_input)" if args.coverage == "stream"
                                    else "difference arrays + teardown scan"),
                       "algorithmic_bytes_per_record": round(algo_rec, 2),
                       "hbm_frac_whole_pass": round(value / world * algo_rec / (HBM_PEAK_GBS * 1e9), 4)},
            "roofline": roofline, "cpu_baseline": cpu, "kernels": kernels,
            "parity_check": "total==records" if ok else "FAILED total!=records",
            "generate_s": round(t_gen, 2),
        }
        print(json.dumps(out), flush=True)
    ctx.free_batch(db)
    ctx.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    return 0 if ok else 1


def pmc_traffic(n: int, args):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC summary
    (profiles/*_traffic.json, made by tools/collect_traffic.py from separate --pmc FETCH_SIZE /
    WRITE_SIZE passes of this same command).  Only valid for the workload it was collected on."""
    import glob
    if n != 100_000_000 or args.read_len != 150 or args.workload != "fixed":
        return None, "no PMC summary for this workload"
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")))
    if not files:
        return None, "profiles/*_traffic.json absent"
    try:
        with open(files[-1]) as f:
            doc = json.load(f)
        for name, v in doc["kernels"].items():
            if "k_qual_perm" in name or "k_qual_win" in name:
                return v["hbm_bytes_corrected"], os.path.relpath(files[-1], ROOT)
    except Exception as e:  # noqa: BLE001
        return None, f"unreadable: {e}"
    return None, "kernel not in summary"


def cpu_baseline(lib, host, ffi, scfg, sample: int, max_len: int):
    """The CPU restatement of the reference loop (oracle/, single thread, record at a
    time, two passes) on a bounded sample of the same workload.  Reported baseline only."""
    try:
        from oracle import oracle_py
        hb = host.synth_host_batch(scfg, 0, sample, lib)
        orc = oracle_py.Oracle([CHR1, CHR2], [1, 1], facets=ffi.FACETS_DEFAULT, max_read_len=max_len,
                               gc_seed=0x4E4753)
        chunk = 250_000
        for lo in range(0, sample, chunk):
            orc.process_batch(hb.slice(lo, min(sample, lo + chunk)))
        t_scan = orc.elapsed_seconds()
        orc.finalize()
        t_all = orc.elapsed_seconds()
        orc.close()
        return {"value": round(sample / t_all, 1), "unit": "records/s", "cores": 1, "kind": "port",
                "sample": ("first %d records of the workload, all default facets; %.1f s facet loop + %.1f s chr1 "
                           "coverage teardown (O(L), not amortised over the full file)"
                           % (sample, t_scan, t_all - t_scan)),
                "scan_only_value": round(sample / t_scan, 1)}
    except Exception as e:  # the baseline is reported, never required
        return {"value": None, "unit": "records/s", "cores": 1, "kind": "port", "sample": f"failed: {e}"}


if __name__ == "__main__":
    sys.exit(main())
