#!/usr/bin/env python3
"""`ngs qc` of ONE BAM file on several GPUs of a node: every rank ingests its BGZF block range on its
own GPU (include/ngsq_bam.h "sharded device ingest"), the record-facet counters are all-reduced and the
coverage teardown is owner-computes (ngs_amd/shard.py).  Rank 0 writes <out>/<bam name>.results.json.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        tools/qc_sharded.py sample.bam GRCh38_no_alt_AnalysisSet [-o DIR] [--batch-records N]
    (testing on a one-GPU box: add --backend gloo --same-gpu)

Default facets of the reference (General, Template Length, GC Content, Quality Score, Coverage).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def load_genome(name):
    path = os.path.join(ROOT, "ngs_amd", "data", name + ".tsv")
    if not os.path.exists(path):
        raise SystemExit(f"reference genome is not supported: {name}")
    seqs = {}
    for line in open(path):
        if line.startswith("#") or not line.strip():
            continue
        f = line.rstrip("\n").split("\t")
        # get_primary_assembly (src/utils/genome.rs:59-83): autosomes + sex + alt + unlocalized + unplaced
        seqs[f[0]] = f[1] in ("autosome", "sex", "alt", "unlocalized", "unplaced")
    return seqs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("bam")
    ap.add_argument("genome")
    ap.add_argument("-o", "--output-directory", default=".")
    ap.add_argument("--batch-records", type=int, default=1 << 21)
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--coverage", choices=["stream", "array"], default="stream")
    ap.add_argument("--same-gpu", action="store_true", help="all ranks on GPU 0 (test boxes with one GPU)")
    a = ap.parse_args()
    import torch
    import torch.distributed as dist

    from ngs_amd import build, ffi, host, shard

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    gpu = 0 if a.same_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    torch.cuda.set_device(gpu)
    if a.backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", gpu))
    else:
        dist.init_process_group(a.backend, rank=rank, world_size=world)
    coll = None if a.backend == "nccl" else "cpu"
    if rank == 0:
        build.build(verbose=False)
    dist.barrier()
    lib = ffi.load_library()
    t0 = time.perf_counter()
    # header facts (every rank reads the header itself)
    h0 = C.c_void_p()
    if lib.ngsq_bam_open(a.bam.encode(), 2, C.byref(h0)) != 0:
        raise SystemExit(lib.ngsq_bam_last_error().decode())
    if lib.ngsq_bam_check_index(a.bam.encode()) != 0:
        raise SystemExit(lib.ngsq_bam_last_error().decode())
    n_refs = lib.ngsq_bam_n_refs(h0)
    names = [lib.ngsq_bam_ref_name(h0, r).decode() for r in range(n_refs)]
    lens = [lib.ngsq_bam_ref_len(h0, r) for r in range(n_refs)]
    lib.ngsq_bam_close(h0)
    genome = load_genome(a.genome)
    for n in names:
        if n not in genome:
            raise SystemExit(f'Sequence "{n}" not found in specified reference genome. Did you set the correct reference genome?')
    ctx = host.QcContext(lens, [int(genome[n]) for n in names], facets=ffi.FACETS_DEFAULT, device=gpu,
                         max_read_len=1024, gc_seed=0x4E4753,
                         # the file is indexed, i.e. coordinate-sorted: Coverage streams; behind the first shard the
                         # first Mi positions per sequence stay on the exchanged array (reads of the shard in front)
                         sorted_input=a.coverage == "stream", cov_head_guard=(1 << 20) if rank else 0, lib=lib)
    views = shard.device_views(ctx, torch, gpu)
    h, info = shard.open_file_shard(lib, ctx._ctx, a.bam, rank, world, dist, torch, coll_device=coll or f"cuda:{gpu}")
    n = 0
    while True:
        b = ffi.Batch()
        if lib.ngsq_bam_next_batch_device(h, ctx._ctx, a.batch_records, C.byref(b)) != 0:
            raise SystemExit(lib.ngsq_bam_last_error().decode())
        if b.n_records == 0:
            break
        n += int(b.n_records)
        if lib.ngsq_process_batch(ctx._ctx, C.byref(b), ffi.PASS_BOTH) != 0:
            raise SystemExit(lib.ngsq_last_error(ctx._ctx).decode())
    lib.ngsq_bam_close(h)
    if world > 1:
        shard.owner_teardown(ctx, dist, torch, views, coll_device=coll)
    ctx.finalize()
    res = ctx.results(names)
    dt = time.perf_counter() - t0
    if rank == 0:
        os.makedirs(a.output_directory, exist_ok=True)
        out = os.path.join(a.output_directory, os.path.basename(a.bam) + ".results.json")
        with open(out, "w") as f:
            json.dump(res, f, indent=2)
        total = res["general"]["records"]["total"]
        print(json.dumps({"ranks": world, "records": total, "seconds": round(dt, 3),
                          "records_per_s": round(total / dt), "output": out}))
    ctx.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
