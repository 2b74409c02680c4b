"""Build ngs_amd/libngsq.so with hipcc for gfx950 (cross-compiles without a GPU).

    python -m ngs_amd.build [--force]
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libngsq.so")
SOURCES = ["kernels.hip", "qual_kernel.hip", "fields_kernel.hip", "cov_scan.hip", "cov_stream.hip", "synth.hip", "bgzf_inflate.hip",
           "bam_device.hip", "features_kernel.hip", "edits_kernel.hip", "exchange_kernels.hip", "comm.cpp", "exchange.cpp", "mem_pool.cpp", "context.cpp", "stager.cpp", "results.cpp", "bam_reader.cpp", "bam_device_reader.cpp", "synth_bam.cpp",
           "reference.cpp", "reference_kernels.hip"]
HEADERS = ["kernels.h", "context.h", "comm.h", "mem_pool.h", "ingest_kernels.h", "bgzf.h", "reference_kernels.h", "../../include/ngsq.h",
           "../../include/ngsq_reference.h", "../../include/ngsq_comm.h",
           "../../include/ngsq_shared.h", "../../include/ngsq_synth.h", "../../include/ngsq_bam.h", "../../include/ngsq_stage.h"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-result", "-fno-gpu-rdc"]
FLAGS += os.environ.get("NGSQ_EXTRA_FLAGS", "").split()  # measurement builds, e.g. -DNGSQ_INFLATE_PROFILE (use --force)
OBJ_DIR = os.path.join(HERE, "_obj")  # per-source objects (git-ignored): only changed sources recompile


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def up_to_date() -> bool:
    if not os.path.exists(OUT):
        return False
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return all(os.path.getmtime(d) <= t for d in deps)


CLI_SRC = os.path.join(CSRC, "cli", "ngs_main.cpp")
CLI_OUT = os.path.join(HERE, "ngs")


def build_cli(force: bool = False, verbose: bool = True) -> str:
    """The `ngs qc` command line (host C++ only), linked against libngsq.so next to it."""
    deps = [CLI_SRC, OUT, os.path.join(HERE, "..", "include", "ngsq.h"), os.path.join(HERE, "..", "include", "ngsq_bam.h"),
            os.path.join(HERE, "..", "include", "ngsq_reference.h"), os.path.join(CSRC, "cli", "gff_loader.h")]
    if not force and os.path.exists(CLI_OUT) and all(os.path.getmtime(d) <= os.path.getmtime(CLI_OUT) for d in deps):
        return CLI_OUT
    cmd = ["g++", "-O2", "-std=c++17", "-Wall", CLI_SRC, "-L" + HERE, "-lngsq", "-Wl,-rpath,$ORIGIN",
           "-Wl,-rpath-link," + "/opt/rocm/lib", "-lz", "-lpthread", "-o", CLI_OUT + ".tmp"]
    if verbose:
        print("[ngs_amd.build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(CLI_OUT + ".tmp", CLI_OUT)
    return CLI_OUT


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and up_to_date():
        build_cli(False, verbose)
        return OUT
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(OBJ_DIR, exist_ok=True)
    hdr_t = max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS)
    hdr_t = max(hdr_t, os.path.getmtime(os.path.abspath(__file__)))
    cc = hipcc()

    def compile_one(src: str) -> str:
        obj = os.path.join(OBJ_DIR, src.replace("/", "_") + ".o")
        path = os.path.join(CSRC, src)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(hdr_t, os.path.getmtime(path)):
            return obj
        cmd = [cc] + FLAGS + ["-c", path, "-o", obj]
        if verbose:
            print("[ngs_amd.build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True, cwd=CSRC)
        return obj

    with ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [cc, "--offload-arch=gfx950", "-fno-gpu-rdc", "-shared", "-fPIC"] + objs + ["-lz", "-lpthread", "-ldl", "-lrt", "-o", OUT + ".tmp"]
    if verbose:
        print("[ngs_amd.build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=CSRC)
    os.replace(OUT + ".tmp", OUT)
    build_cli(True, verbose)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT)
