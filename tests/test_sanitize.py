"""AddressSanitizer / UndefinedBehaviorSanitizer builds, CPU only (GPU ASan is not available on this pool):
  * the oracle (oracle/Makefile: libngsq_oracle_asan.so) on the hand goldens and on random records with every odd
    shape the generators know;
  * the host C++ of the product that runs without a GPU -- the BGZF/BAM host reader and index reader
    (ngs_amd/csrc/bam_reader.cpp, bgzf.h) and the synthetic BAM writer (synth_bam.cpp) -- on well-formed files,
    on truncated / corrupted / crafted ones (incl. the extra-field subfield that used to be read past its end);
  * the argument and early-error paths of the `ngs qc` command line (ngs_amd/csrc/cli/ngs_main.cpp) that end before
    anything touches a GPU.
Each check runs in a child process with the sanitizer runtime preloaded; a report on stderr or a non-zero exit fails."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = ["-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-fno-sanitize-recover=undefined"]


def libasan():
    p = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(p) or not os.path.exists(p):
        pytest.skip("libasan.so not found")
    return os.path.realpath(p)


def run_child(code, tmp_path, extra_env=None):
    env = dict(os.environ, LD_PRELOAD=libasan(), ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1", PYTHONPATH=ROOT)
    env.update(extra_env or {})
    r = subprocess.run([sys.executable, "-c", textwrap.dedent(code)], capture_output=True, text=True, env=env, cwd=str(tmp_path),
                       timeout=600)
    bad = [l for l in r.stderr.splitlines() if "AddressSanitizer" in l or "runtime error:" in l or "LeakSanitizer" in l]
    assert r.returncode == 0 and not bad, f"rc={r.returncode}\n{r.stdout[-2000:]}\n{r.stderr[-4000:]}"
    return r.stdout


def test_oracle_under_asan_ubsan(tmp_path):
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "libngsq_oracle_asan.so"], check=True, capture_output=True)
    out = run_child("""
        import json, os, numpy as np
        from oracle import oracle_py
        oracle_py.LIB_PATH = os.path.join(os.path.dirname(oracle_py.__file__), "libngsq_oracle_asan.so")
        oracle_py._lib = None
        from ngs_amd import ffi
        from tests.util import batch_from_records, random_batch, json_equal, to_fixed_stride
        from tests.test_oracle_golden import load_gold, load_gold_edits
        for g in (load_gold(), load_gold_edits()):
            cfg = g["config"]
            o = oracle_py.Oracle(cfg["ref_len"], cfg["ref_is_primary"], facets=cfg["facets"], bin_size=cfg["bin_size"],
                                 max_read_len=cfg["max_read_len"], ref_bases=cfg.get("ref_bases"))
            o.process_batch(batch_from_records(g["records"]))
            o.finalize()
            json_equal(o.results(cfg["ref_names"]), g["expected"])
        rng = np.random.default_rng(3)
        ref_len = [30_000, 4_000, 700]
        bases = [rng.choice(np.array([1, 2, 4, 8, 15], dtype=np.uint8), size=n) for n in ref_len]
        for weird in (True, False):
            hb = random_batch(rng, 4000, ref_len, weird=weird)
            for batch in (hb, to_fixed_stride(hb)):
                o = oracle_py.Oracle(ref_len, [1, 0, 1], facets=0x3F, bin_size=777, max_read_len=320, gc_seed=9, ref_bases=bases)
                o.process_batch(batch.slice(0, 1500))
                o.process_batch(batch.slice(1500, batch.n))
                o.finalize(allow_malformed=True)
                o.results_json(["a", "b", "c"])
                o.close()
        print("oracle-asan-ok")
    """, tmp_path)
    assert "oracle-asan-ok" in out


@pytest.fixture(scope="module")
def host_asan_lib(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("asan") / "libngsq_host_asan.so")
    csrc = os.path.join(ROOT, "ngs_amd", "csrc")
    subprocess.run(["g++", "-std=c++17", *SAN, "-fPIC", "-shared", "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__",
                    os.path.join(csrc, "bam_reader.cpp"), os.path.join(csrc, "synth_bam.cpp"), "-lz", "-lpthread", "-o", out],
                   check=True)
    return out


def test_host_reader_and_writer_under_asan_ubsan(host_asan_lib, tmp_path):
    out = run_child(f"""
        import ctypes as C, os, struct, zlib, numpy as np
        from ngs_amd import ffi, host
        lib = C.CDLL({host_asan_lib!r})
        for name in ("ngsq_bam_last_error", "ngsq_bam_open", "ngsq_bam_close", "ngsq_bam_check_index", "ngsq_bam_index_ref_starts",
                     "ngsq_bam_seek", "ngsq_bam_n_refs", "ngsq_bam_ref_name", "ngsq_bam_ref_len", "ngsq_bam_next_batch",
                     "ngsq_bam_records_read", "ngsq_synth_write_bam"):
            f = getattr(lib, name)
            f.restype, f.argtypes = ffi.PROTOTYPES[name]

        def read_all(path, threads=3, per=777):
            h = C.c_void_p()
            rc = lib.ngsq_bam_open(path.encode(), threads, C.byref(h))
            if rc:
                return rc, 0
            n = 0
            while True:
                b = ffi.Batch()
                rc = lib.ngsq_bam_next_batch(h, per, C.byref(b))
                if rc or b.n_records == 0:
                    break
                n += b.n_records
                # touch the columns the reader handed out
                np.ctypeslib.as_array(C.cast(b.flag, C.POINTER(C.c_uint16)), shape=(b.n_records,)).sum()
            lib.ngsq_bam_close(h)
            return rc, n

        for mode, n in ((ffi.SYNTH_FIXED, 20_000), (ffi.SYNTH_MIXED, 12_000)):
            cfg = host.synth_config(n, mode=mode, ref_len=3_000_000)
            path = f"s{{mode}}.bam"
            assert lib.ngsq_synth_write_bam(C.byref(cfg), path.encode(), n, 6, 3) == 0
            assert read_all(path) == (0, n)
            assert lib.ngsq_bam_check_index(path.encode()) == 0
            starts = (C.c_uint64 * 2)(); bins = C.c_uint64()
            assert lib.ngsq_bam_index_ref_starts(path.encode(), 2, starts, C.byref(bins)) == 0 and bins.value > 0
            h = C.c_void_p(); assert lib.ngsq_bam_open(path.encode(), 2, C.byref(h)) == 0
            assert lib.ngsq_bam_seek(h, starts[0]) == 0
            b = ffi.Batch(); assert lib.ngsq_bam_next_batch(h, 10, C.byref(b)) == 0 and b.n_records == 10
            lib.ngsq_bam_close(h)
        good = open("s0.bam", "rb").read()
        # damaged files: every one must fail (or end) cleanly, never read out of bounds
        cases = {{
            "trunc_mid_block": good[:len(good) // 2 + 13],
            "trunc_header": good[:30],
            "bad_magic": b"\\x1f\\x8b\\x08\\x00" + good[4:],
            "flipped_payload": good[:5000] + bytes([good[5000] ^ 0x5A]) + good[5001:],
            "bsize_too_small": good[:16] + b"\\x05\\x00" + good[18:],
            "empty": b"",
        }}
        # an extra field whose BC subfield claims more bytes than the field holds (ADVICE r1): XLEN 6 with SLEN 2 needs 6
        # bytes; here XLEN says 5, and the file ends right behind the field
        hdr = bytes([31, 139, 8, 4, 0, 0, 0, 0, 0, 255]) + struct.pack("<H", 5) + b"BC" + struct.pack("<H", 2) + b"\\x1b"
        cases["subfield_past_extra_field"] = hdr
        cases["subfield_slen_huge"] = bytes([31, 139, 8, 4, 0, 0, 0, 0, 0, 255]) + struct.pack("<H", 6) + b"XX" + struct.pack("<H", 65535) + b"\\x00\\x00" + b"\\x00" * 40
        for name, data in cases.items():
            open(name + ".bam", "wb").write(data)
            rc, n = read_all(name + ".bam")
            assert rc != 0 or n < 20_000, (name, rc, n)
        # a BAI cut short / with a wrong magic
        bai = open("s0.bam.bai", "rb").read()
        for k, data in enumerate((bai[:7], bai[:len(bai) // 2], b"BAJ\\x01" + bai[4:], bai + b"\\x00" * 3)):
            open("s0.bam.bai", "wb").write(data)
            lib.ngsq_bam_check_index(b"s0.bam")
        print("host-asan-ok")
    """, tmp_path)
    assert "host-asan-ok" in out


def test_cli_argument_paths_under_asan_ubsan(tmp_path):
    """`ngs qc` up to the point where it would create a context: usage errors, unknown genome, format sniffing, missing
    index, header / genome concordance, GFF and FASTA loading.  (Linked against the regular libngsq.so: only the
    command line's own code is instrumented.)"""
    from ngs_amd import build
    from tests import bamio
    from tests.test_cli import GENOME, LENS, NAMES, sorted_batch
    build.build(verbose=False)
    exe = str(tmp_path / "ngs_asan")
    lib_dir = os.path.join(ROOT, "ngs_amd")
    subprocess.run(["g++", "-std=c++17", *SAN, os.path.join(lib_dir, "csrc", "cli", "ngs_main.cpp"), "-L" + lib_dir, "-lngsq",
                    "-Wl,-rpath," + lib_dir, "-Wl,-rpath-link,/opt/rocm/lib", "-lz", "-o", exe], check=True)
    hb = sorted_batch(1, 50)
    bam = str(tmp_path / "a.bam")
    bamio.write_bam(bam, hb, NAMES, LENS)
    gff = tmp_path / "m.gff"
    gff.write_text("##gff-version 3\nchr1\tx\tgene\t100\t900\t.\t+\t.\tID=g\nchr1\tx\texon\t100\t300\t.\t+\t.\tID=e\n")
    fa = tmp_path / "r.fa"
    fa.write_text(">chr1\n" + "ACGT" * 10 + "\n")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1",
               NGSQ_DATA_DIR=os.path.join(lib_dir, "data"))
    runs = [
        (["--help"], 0), ([], 1), (["qc"], 1), (["qc", bam], 1), (["qc", bam, "hg19"], 1), (["qc", "--bogus", bam, GENOME], 1),
        (["qc", str(tmp_path / "x.sam"), GENOME], 1), (["qc", str(tmp_path / "none.bam"), GENOME], 1),
        (["qc", bam, GENOME, "--only", "Nope", "-o", str(tmp_path / "o1")], 1),
        (["qc", bam, GENOME, "-f", str(gff), "--only", "Nope", "-o", str(tmp_path / "o2")], 1),
        (["qc", bam, GENOME, "-r", str(fa), "-o", str(tmp_path / "o3")], 1),      # FASTA lacks chr2 / wrong length
        (["qc", bam, GENOME, "--gpus", "99"], 1), (["qc", bam, GENOME, "--gpus", "2", "-n", "5"], 1),
        (["qc", bam, GENOME, "--worker", "garbage"], 1), (["qc", bam, GENOME, "--coverage", "sideways"], 2),
        (["qc", bam, GENOME, "-n"], 1),
    ]
    for args, want in runs:
        r = subprocess.run([exe, *args], capture_output=True, text=True, env=env, timeout=120)
        bad = [l for l in r.stderr.splitlines() if "AddressSanitizer" in l or "runtime error:" in l]
        assert not bad, (args, r.stderr[-3000:])
        assert r.returncode == want, (args, r.returncode, r.stderr[-600:])


def test_gff_loader_under_asan_ubsan(tmp_path):
    """ngs_amd/csrc/cli/gff_loader.h (the parallel parser behind `-f`, round 6) as a stand-alone program under ASan / UBSan:
    well-formed text cut into pieces at every thread count, lines without a terminator, CRLF, a tenth column, numbers that
    overflow, an empty file, `##FASTA` first, gzip -- the intervals kept and the FIRST offending line must not depend on how
    the text was cut."""
    import gzip
    src = tmp_path / "drive.cpp"
    src.write_text(r'''
#include <cstdio>
#include <cstdint>
#include "gff_loader.h"
int main(int argc, char **argv) {
    const std::string names[5] = {"five_prime_UTR", "three_prime_UTR", "CDS", "exon", "gene"};
    std::set<std::string> primary = {"chr1", "chr2", "chrX"};
    std::map<std::string, uint32_t> idx = {{"chr1", 0}, {"chr2", 1}, {"chrM", 2}};
    const std::string path = argv[1];
    const bool gz = path.size() > 3 && path.compare(path.size() - 3, 3, ".gz") == 0;
    GeneModel m = load_gff_parallel(path, gz, names, primary, idx, atoi(argv[2]));
    unsigned long long sum = 0;
    for (size_t i = 0; i < m.ref.size(); i++) sum = sum * 1000003ull + m.ref[i] * 7 + m.name[i] * 3 + m.start[i] + 5ull * m.stop[i];
    printf("%zu %llu %llu|%s\n", m.ref.size(), sum, (unsigned long long)m.lines, m.error.c_str());
    return 0;
}''')
    exe = str(tmp_path / "drive")
    subprocess.run(["g++", "-std=c++17", *SAN, "-I", os.path.join(ROOT, "ngs_amd", "csrc", "cli"), str(src), "-lz", "-lpthread", "-o", exe], check=True)
    import random
    rnd = random.Random(5)
    types = ["gene", "exon", "CDS", "five_prime_UTR", "three_prime_UTR", "transcript"]
    lines = ["##gff-version 3", "#c"]
    for k in range(30_000):
        s = rnd.randrange(1, 10 ** 6)
        lines.append(f"{rnd.choice(['chr1', 'chr2', 'chrX', 'chrM', 'scaffold_9'])}\tsrc\t{rnd.choice(types)}\t{s}\t{s + rnd.randrange(0, 5000)}\t.\t{rnd.choice('+-')}\t.\tID=x{k};note=" + "y" * rnd.randrange(0, 300))
    good = "\n".join(lines) + "\n"
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1")

    def run(text, threads, name="t.gff"):
        p = tmp_path / name
        if name.endswith(".gz"):
            with gzip.open(p, "wb") as f:
                f.write(text if isinstance(text, bytes) else text.encode())
        else:
            p.write_bytes(text if isinstance(text, bytes) else text.encode())
        r = subprocess.run([exe, str(p), str(threads)], capture_output=True, text=True, env=env, timeout=120)
        bad = [l for l in r.stderr.splitlines() if "AddressSanitizer" in l or "runtime error:" in l]
        assert r.returncode == 0 and not bad, r.stderr[-3000:]
        return r.stdout.strip()

    want = run(good, 1)
    n_kept = int(want.split()[0])
    assert n_kept > 5000 and want.endswith("|")
    for t in (2, 3, 8, 31):
        assert run(good, t) == want
    assert run(good, 4, "t.gff3.gz") == want
    assert run(good[:-1], 5).split("|")[0].split()[:2] == want.split("|")[0].split()[:2]          # no newline at the end
    assert run(good.replace("\n", "\r\n"), 3).split()[:2] == want.split()[:2]                      # CRLF
    assert run(good + "##FASTA\n>x\nACGT\tnot\ta\trecord\n", 4).split()[:2] == want.split()[:2]   # ##FASTA ends the records
    assert run("##FASTA\n" + good, 2).startswith("0 0 ")
    assert run("", 3).startswith("0 0 0|")
    # the first offending line, wherever the pieces are cut
    bad_at = 20_001
    broken = lines[:]
    broken[bad_at] = "chr1\tsrc\tgene\t12\t11\t.\t+\t.\tID=stop_before_start"
    broken[bad_at + 3000] = "chr1\tsrc\tgene\tx\t11\t.\t+\t.\tID=later"
    for t in (1, 2, 7):
        out = run("\n".join(broken) + "\n", t)
        assert out.endswith(f"|invalid GFF record on line {bad_at + 1} of {tmp_path / 't.gff'}"), out
    for bad_line, msg in (("chr1\tsrc\tgene\t1\t2\t.\t.\t.\tID=s", "attempted to parse strand from value: ."),
                          ("chr1\tsrc\tgene\t1\t2\t.\t+\t.\tID=s\ttenth", "invalid GFF record on line 3"),
                          ("chr1\tsrc\tgene\t1\t99999999999999999999\t.\t+\t.\tID=s", "invalid GFF record on line 3"),
                          ("chr1\tsrc\tgene\t0\t2\t.\t+\t.\tID=s", "invalid GFF record on line 3"),
                          ("chr1\tsrc\tgene", "invalid GFF record on line 3"),
                          ("chrM\tsrc\tgene\t1\t2\t.\t?\t.\tID=not_primary_strand_never_parsed", "")):
        out = run("##gff-version 3\n#c\n" + bad_line + "\n", 2)
        assert msg in out.split("|", 1)[1] and (msg or out.endswith("|")), (bad_line, out)
