"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads and
exports every symbol include/*.h declares; structure layouts match the
binding; and without a GPU the product fails loudly instead of falling back."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from ngs_amd import ffi, host

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = set()
    for h in sorted(os.listdir(os.path.join(ROOT, "include"))):
        if not h.endswith(".h") or h == "ngsq_shared.h":  # ngsq_shared.h: inline host/device functions, no exports
            continue
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(ngsq_[a-z0-9_]+)\s*\(", text))
    return names


def test_every_declared_symbol_is_exported(lib):
    decl = declared_symbols()
    assert len(decl) >= 40
    for name in sorted(decl):
        assert hasattr(lib, name), f"libngsq.so does not export {name}"
    # and the binding covers exactly the declared surface
    assert decl == set(ffi.PROTOTYPES), decl ^ set(ffi.PROTOTYPES)


def test_struct_layouts_match_the_header(tmp_path):
    """Compile a C probe against include/ngsq.h and compare sizeof/offsetof with ctypes."""
    src = tmp_path / "probe.c"
    src.write_text(r'''
#include <stdio.h>
#include <stddef.h>
#include "ngsq.h"
#include "ngsq_synth.h"
#include "ngsq_comm.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu %zu\n", sizeof(ngsq_config), sizeof(ngsq_batch), sizeof(ngsq_general_metrics),
         sizeof(ngsq_gc_metrics), sizeof(ngsq_error_counts), sizeof(ngsq_kernel_time), sizeof(ngsq_synth_config));
  printf("%zu %zu %zu %zu\n", offsetof(ngsq_config, gc_seed), offsetof(ngsq_config, stream),
         offsetof(ngsq_batch, seq_stride), offsetof(ngsq_batch, cigar_ops));
  printf("%zu %zu %zu %zu %zu %zu %zu\n", sizeof(ngsq_p2p), sizeof(ngsq_comm_ops), sizeof(ngsq_exchange_report),
         sizeof(ngsq_shard_state), offsetof(ngsq_shard_state, touched), offsetof(ngsq_shard_state, teardown_range),
         offsetof(ngsq_exchange_report, host_syncs));
  return 0;
}''')
    exe = tmp_path / "probe"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    sizes = [int(x) for x in out]
    assert sizes[:7] == [C.sizeof(ffi.Config), C.sizeof(ffi.Batch), C.sizeof(ffi.GeneralMetrics),
                         C.sizeof(ffi.GcMetrics), C.sizeof(ffi.ErrorCounts), C.sizeof(ffi.KernelTime),
                         C.sizeof(ffi.SynthConfig)]
    assert sizes[7:11] == [ffi.Config.gc_seed.offset, ffi.Config.stream.offset, ffi.Batch.seq_stride.offset,
                           ffi.Batch.cigar_ops.offset]
    assert sizes[11:] == [C.sizeof(ffi.P2P), C.sizeof(ffi.CommOps), C.sizeof(ffi.ExchangeReport), C.sizeof(ffi.ShardState),
                          ffi.ShardState.touched.offset, ffi.ShardState.teardown_range.offset,
                          ffi.ExchangeReport.host_syncs.offset]


def test_abi_version_and_pure_functions(lib):
    assert lib.ngsq_abi_version() == ffi.ABI_VERSION
    # gc_content.rs:69-74: offset 0 when l_seq <= 100, else in 0..l_seq-100 (exclusive)
    assert lib.ngsq_gc_offset(1, 2, 100) == 0 and lib.ngsq_gc_offset(1, 2, 50) == 0
    offs = {lib.ngsq_gc_offset(7, i, 150) for i in range(2000)}
    assert offs == set(range(50))
    assert {lib.ngsq_gc_offset(7, i, 101) for i in range(50)} == {0}


@pytest.mark.skipif(ffi.load_library().ngsq_device_count() > 0, reason="checks the no-GPU failure mode")
def test_no_gpu_fails_loudly(lib):
    with pytest.raises(host.NgsqError) as ei:
        host.QcContext([1000])
    assert ei.value.code == ffi.ERR_NO_DEVICE
    assert "no CPU fallback" in ei.value.message


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises(ffi.LibraryNotBuilt):
        ffi.load_library(str(tmp_path / "libngsq.so"))


def test_synthetic_host_generator_is_a_pure_function_of_the_index(lib):
    """Any shard regenerates its slice: records [a,b) generated alone equal the
    slice of a larger generation, in both modes."""
    for mode in (ffi.SYNTH_FIXED, ffi.SYNTH_MIXED):
        cfg = host.synth_config(5000, mode=mode, ref_len=200_000)
        whole = host.synth_host_batch(cfg, 0, 3000, lib)
        part = host.synth_host_batch(cfg, 1000, 500, lib)
        sl = whole.slice(1000, 1500)
        for k, a in part.cols.items():
            if a is None:
                assert sl.cols[k] is None
            else:
                np.testing.assert_array_equal(a, sl.cols[k], err_msg=k)
        # coordinate-sorted by construction, inside the reference
        pos = whole.cols["pos"]
        assert (np.diff(pos.astype(np.int64)) >= 0).all() and pos.min() >= 1 and pos.max() < 200_000 - 5000


def test_synthetic_distributions(lib):
    cfg = host.synth_config(200_000, ref_len=10_000_000)
    hb = host.synth_host_batch(cfg, 0, 200_000, lib)
    f = hb.cols["flag"]
    frac = lambda m: float(((f & m) != 0).mean())
    assert abs(frac(0x1) - 0.98) < 0.005 and abs(frac(0x4) - 0.01) < 0.003
    assert abs(frac(0x400) - 0.05) < 0.005 and abs(frac(0x100) - 0.01) < 0.003
    q = hb.cols["qual"].reshape(-1, 150)
    assert set(np.unique(q)) == {2, 11, 25, 37}
    assert abs((q[:, 0] == 37).mean() - 0.9) < 0.01 and abs((q[:, 149] == 37).mean() - 0.6) < 0.01
    s = hb.cols["seq"]
    hi, lo = s >> 4, s & 15
    codes = np.concatenate([hi, lo])
    assert set(np.unique(codes)) == {1, 2, 4, 8, 15}
    assert abs(((codes == 2) | (codes == 4)).mean() - 0.41) < 0.01
    t = hb.cols["tlen"][(f & 0x41) == 0x41]
    t = t[(t > 0) & (t <= 1024)]
    assert abs(t.mean() - 350) < 2 and abs(t.std() - 50) < 2


def test_fixed_shapes_are_the_reference_s(lib):
    """tests/golden/reference_constants.json is read out of the reference's source by
    tests/golden/make_reference_constants.py: facet names, histogram capacities, the GC window, the coverage bin, the
    MAPQ that counts as high quality.  include/ngsq.h, the library and the oracle must carry the same numbers."""
    import json
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_constants.json")))
    header = open(os.path.join(ROOT, "include", "ngsq.h")).read()

    def macro(name):
        return int(re.search(rf"#define {name}\s+(\d+)", header).group(1))

    assert macro("NGSQ_MAX_SCORE") == want["max_quality_score"]
    assert macro("NGSQ_GC_WINDOW") == want["gc_truncation_length"]
    assert macro("NGSQ_GC_BINS") == 101                                   # percentages 0..=100 (gc_content.rs:129-138)
    assert macro("NGSQ_EDITS_BINS") == want["default_histogram_capacity"] + 1
    assert macro("NGSQ_VAF_BINS") == want["vaf_histogram_capacity"] + 1
    names = want["facet_names"]
    for bit, key in ((ffi.FACET_GENERAL, "general"), (ffi.FACET_TEMPLATE_LENGTH, "template_length"),
                     (ffi.FACET_GC_CONTENT, "gc_content"), (ffi.FACET_QUALITY_SCORE, "quality_score"),
                     (ffi.FACET_COVERAGE, "coverage"), (ffi.FACET_EDITS, "edits"), (ffi.FACET_FEATURES, "features")):
        assert lib.ngsq_facet_name(bit).decode() == names[key]
    # defaults of ngsq_create (0 = the reference's value) as documented in the header, and in the sources that apply them
    ctx_src = open(os.path.join(ROOT, "ngs_amd", "csrc", "context.cpp")).read()
    assert f"bin_size = {want['coverage_bin_size']}" in ctx_src and f"tlen_cap = {want['template_length_capacity']}" in ctx_src
    assert f"cov_cap = {want['coverage_histogram_capacity']}" in ctx_src
    res_src = open(os.path.join(ROOT, "ngs_amd", "csrc", "results.cpp")).read()
    assert all(f"{c}" in res_src for c in want["genome_covered_by"])
    fields = open(os.path.join(ROOT, "ngs_amd", "csrc", "fields_kernel.hip")).read()
    assert f"mq >= {want['high_quality_mapq']}u" in fields
    oracle_src = open(os.path.join(ROOT, "oracle", "oracle.c")).read()
    assert f">= {want['high_quality_mapq']}" in oracle_src


def test_oracle_defaults_are_the_reference_s(oracle_mod):
    """The same constants, observed in the oracle's behaviour: an empty run reports the reference's histogram shapes."""
    import json
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_constants.json")))
    orc = oracle_mod.Oracle([1000], facets=ffi.FACETS_DEFAULT)
    orc.finalize()
    doc = orc.results(["chr1"])
    assert doc["template_length"]["histogram"]["range_stop"] == want["template_length_capacity"]
    assert len(doc["template_length"]["histogram"]["values"]) == want["template_length_capacity"] + 1
    assert doc["gc_content"]["histogram"]["range_stop"] == 100
    assert doc["coverage"]["coverage_distribution"]["range_stop"] == want["coverage_histogram_capacity"]
    assert sorted(doc["coverage"]["genome_covered_by"]) == sorted(f"{c}x" for c in want["genome_covered_by"])


def test_rec_fixed_ticket_is_ordered_behind_the_tallies(tmp_path):
    """k_rec_fixed's last block publishes the batch's layout statistics (longest read, most CIGAR operations,
    sums) that size the CIGAR column -- every block's tallies must have reached the L2 before its ticket is drawn.  The
    ordering is an explicit `s_waitcnt vmcnt(0)`; this reads the gfx950 ISA the product's flags produce and requires (a) one
    such wait between the last tally atomic and the ticket (the returning add), (b) one in front of the block's barrier."""
    from ngs_amd import build
    src = os.path.join(build.CSRC, "bam_device.hip")
    out = tmp_path / "bam_device.s"
    flags = [f for f in build.FLAGS if f != "-fPIC"]
    subprocess.run([build.hipcc()] + flags + ["--cuda-device-only", "-S", src, "-o", str(out)], check=True, cwd=build.CSRC,
                   stderr=subprocess.DEVNULL)
    body, on = [], False
    for line in open(out):
        if re.match(r"^_ZN4ngsq11k_rec_fixed\w*:", line):
            on = True
        if on:
            body.append(line.strip())
            if line.strip().startswith("s_endpgm"):
                break
    assert body, "k_rec_fixed not found in the ISA"
    ops = [l for l in body if re.match(r"(global_atomic|s_waitcnt vmcnt\(0\)|s_barrier)", l)]
    barrier = ops.index("s_barrier")
    assert "s_waitcnt vmcnt(0)" in ops[:barrier], "no vmcnt(0) in front of the block's barrier"
    after = ops[barrier + 1:]
    # the four tallies (two max, two add, none returning), a wait, then the returning ticket add
    ticket = next(k for k, l in enumerate(after) if l.startswith("global_atomic_add_x2") and "sc0" in l)
    tallies = [l for l in after[:ticket] if l.startswith("global_atomic")]
    assert len(tallies) == 4 and sum(l.startswith("global_atomic_umax_x2") for l in tallies) == 2, tallies
    assert after[ticket - 1] == "s_waitcnt vmcnt(0)", after[:ticket + 1]
