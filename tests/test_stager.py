"""The per-record side of the boundary (include/ngsq_stage.h): a stager under the reference's
`process(&mut self, &Record)` calls (src/qc.rs:165,203-219; src/qc/command.rs:305-316,356-397).

CPU: the columns a stager assembles from records pushed one by one are the batch the records came from
(values, layout choice, the oracle's results on them); argument and lifecycle errors.
GPU (-m gpu): a C program (tests/c/stager_drive.c) drives the C ABI in the reference's call shape -- pass 1 record by record,
`summarize`, pass 2 per sequence setup -> query -> process -> teardown, both -n rules -- and its document equals the oracle's
and the batch path's."""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

from ngs_amd import build, ffi, host
from tests import bamio
from tests.util import json_equal, random_batch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LENS = [60_000, 20_000, 16_569, 2_274]
NAMES = ["chr1", "chr2", "chrM", "chrUn_KI270302v1"]
PRIMARY = [1, 1, 0, 1]


def unpack_bases(packed, l):
    codes = np.empty(2 * len(packed), dtype=np.uint8)
    codes[0::2] = packed >> 4
    codes[1::2] = packed & 15
    return np.ascontiguousarray(codes[:l])


def push_record(lib, st, hb, i, packed, rid=ffi.STAGE_NO_ID):
    c = hb.cols
    l = int(c["l_seq"][i])
    sq = np.ascontiguousarray(c["seq"][int(c["seq_off"][i]):int(c["seq_off"][i + 1])])
    ql = np.ascontiguousarray(c["qual"][int(c["qual_off"][i]):int(c["qual_off"][i + 1])])
    cg = np.ascontiguousarray(c["cigar"][int(c["cigar_off"][i]):int(c["cigar_off"][i + 1])])
    fixed = (int(c["flag"][i]), int(c["mapq"][i]), int(c["ref_id"][i]), int(c["pos"][i]), int(c["mate_ref_id"][i]), int(c["tlen"][i]), l)
    if packed:
        # the record's own BAM bytes: absent qualities are l_seq bytes of 0xFF
        qb = ql if len(ql) else np.full(l, 0xFF, dtype=np.uint8)
        return lib.ngsq_stager_push_packed(st, *fixed, sq.ctypes.data if l else None, qb.ctypes.data if l else None,
                                           cg.ctypes.data if len(cg) else None, len(cg), rid)
    bases = unpack_bases(sq, l)
    return lib.ngsq_stager_push(st, *fixed, bases.ctypes.data if l else None, ql.ctypes.data if len(ql) else None, len(ql),
                                cg.ctypes.data if len(cg) else None, len(cg), rid)


def view(lib, st):
    """The staged batch as a HostBatch (numpy copies of the stager's columns)."""
    b = ffi.Batch()
    assert lib.ngsq_stager_view(st, C.byref(b)) == 0
    n = int(b.n_records)

    def arr(ptr, dt, count):
        if not ptr or not count:
            return np.zeros(0, dtype=dt)
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(count * np.dtype(dt).itemsize,)).view(dt).copy()
    cols = {k: arr(getattr(b, k), host.COLUMN_DTYPES[k], n) for k in host.FIXED_COLUMNS}
    cols["seq"] = arr(b.seq, np.uint8, int(b.seq_bytes))
    cols["qual"] = arr(b.qual, np.uint8, int(b.qual_bytes))
    cols["cigar"] = arr(b.cigar, np.uint32, int(b.cigar_ops))
    cols["seq_off"] = arr(b.seq_off, np.uint64, n + 1) if b.seq_off else None
    cols["qual_off"] = arr(b.qual_off, np.uint64, n + 1) if b.qual_off else None
    cols["cigar_off"] = arr(b.cigar_off, np.uint64, n + 1) if b.cigar_off else None
    cols["record_id"] = arr(b.record_id, np.uint64, n) if b.record_id else None
    hb = host.HostBatch(n, cols, int(b.seq_stride), int(b.qual_stride), int(b.cigar_stride), int(b.first_record_index))
    return hb, b


def new_stager(lib, cap, flags=ffi.STAGE_PAGEABLE):
    st = C.c_void_p()
    assert lib.ngsq_stager_create(cap, flags, C.byref(st)) == 0, lib.ngsq_stager_last_error(None)
    return st


@pytest.mark.parametrize("packed", [False, True])
def test_pushed_records_are_the_batch_they_came_from(lib, oracle_mod, packed):
    """Random records of every odd shape (no bases, no qualities, no CIGAR, all nine operations, unplaced) pushed one by one:
    the staged columns equal the source batch's, and the oracle gives the same document for both."""
    rng = np.random.default_rng(11)
    hb = random_batch(rng, 700, LENS, max_len=260, first_record_index=5)
    st = new_stager(lib, 1000)
    assert lib.ngsq_stager_rewind(st, 5) == 0
    for i in range(hb.n):
        assert push_record(lib, st, hb, i, packed) == 0, lib.ngsq_stager_last_error(st)
    assert lib.ngsq_stager_len(st) == hb.n == lib.ngsq_stager_pushed(st)
    got, b = view(lib, st)
    assert got.first_record_index == 5 and not b.record_id and b.max_l_seq == int(hb.cols["l_seq"].max())
    assert got.cols["seq_off"] is not None and got.cols["cigar_off"] is not None      # ragged: the offsets layout
    for k in list(host.FIXED_COLUMNS) + ["seq_off", "qual_off", "cigar_off", "qual", "cigar"]:
        assert np.array_equal(got.cols[k], hb.cols[k]), k
    # SEQ: equal but for the unused low nibble of an odd read's last byte, which the stager clears
    want = hb.cols["seq"].copy()
    odd = np.nonzero(hb.cols["l_seq"] & 1)[0]
    want[hb.cols["seq_off"][odd + 1].astype(np.int64) - 1] &= 0xF0
    assert np.array_equal(got.cols["seq"], want)
    docs = []
    for batch in (hb, got):
        o = oracle_mod.Oracle(LENS, PRIMARY, facets=ffi.FACETS_DEFAULT, max_read_len=300, gc_seed=3)
        o.process_batch(batch)
        try:
            o.finalize()
        except Exception:   # noqa: BLE001 -- random records hold conditions the reference aborts on; the integers still compare
            pass
        docs.append(o.results(NAMES))
        o.close()
    json_equal(docs[1], docs[0])
    lib.ngsq_stager_destroy(st)


@pytest.mark.parametrize("rows", [False, True])
def test_push_records_is_push_packed_per_record(lib, rows):
    """ngsq_stager_push_records over a host batch (offsets layout; fixed-pitch rows) stages what one ngsq_stager_push_packed
    per record stages, stops at a full stager, and refuses a range outside the batch."""
    rng = np.random.default_rng(23)
    if rows:
        cfg = host.synth_config(300, ffi.SYNTH_FIXED, read_len=150, seed=4)
        hb = host.synth_host_batch(cfg, 0, 300, lib)
    else:
        hb = random_batch(rng, 300, LENS, max_len=200)
    st = new_stager(lib, 256)
    took = C.c_uint64(0)
    src = hb.struct()
    assert lib.ngsq_stager_push_records(st, C.byref(src), 10, 500, C.byref(took)) == ffi.ERR_INVALID_ARGUMENT
    assert lib.ngsq_stager_push_records(st, C.byref(src), 10, 290, C.byref(took)) == 0, lib.ngsq_stager_last_error(st)
    assert took.value == 256 == lib.ngsq_stager_len(st)                      # full: the caller flushes and goes on from first + took
    got, b = view(lib, st)
    want = hb.slice(10, 266)
    assert bool(b.seq_off) == (not rows) and (int(b.seq_stride) == 75) == rows
    for k in host.FIXED_COLUMNS:
        assert np.array_equal(got.cols[k], want.cols[k]), k
    if rows:
        for k in ("seq", "qual"):
            assert np.array_equal(got.cols[k][:len(want.cols[k])], want.cols[k]), k
        # (an unmapped record of the generator has no operation: its slot of the fixed-pitch column is not staged)
        assert np.array_equal(got.cols["cigar"], want.cols["cigar"][want.cols["n_cigar"] == 1])
    else:
        for k in ("seq_off", "qual_off", "cigar_off", "qual", "cigar"):
            assert np.array_equal(got.cols[k], want.cols[k]), k
        seq = want.cols["seq"].copy()
        odd = np.nonzero(want.cols["l_seq"] & 1)[0]
        seq[want.cols["seq_off"][odd + 1].astype(np.int64) - 1] &= 0xF0
        assert np.array_equal(got.cols["seq"], seq)
    # the records keep their ordinal in the source batch as their identity
    assert np.array_equal(got.cols["record_id"], np.arange(10, 266, dtype=np.uint64) + np.uint64(hb.first_record_index))
    lib.ngsq_stager_destroy(st)


def test_layout_follows_the_records(lib):
    """Reads of one length with qualities and one operation each are handed over as fixed-pitch rows (the fast kernels); the
    first record that differs turns the same columns into the offsets layout -- nothing is moved."""
    rng = np.random.default_rng(5)
    hb = random_batch(rng, 64, LENS, max_len=150, min_len=150, weird=False)
    one = np.array([150 << 4], dtype=np.uint32)
    st = new_stager(lib, 100)
    for i in range(40):
        c = hb.cols
        sq = np.ascontiguousarray(c["seq"][int(c["seq_off"][i]):int(c["seq_off"][i + 1])])
        ql = np.ascontiguousarray(c["qual"][int(c["qual_off"][i]):int(c["qual_off"][i + 1])])
        assert lib.ngsq_stager_push_packed(st, 0, 60, 0, 100 + i, -1, 0, 150, sq.ctypes.data, ql.ctypes.data, one.ctypes.data, 1, 1000 + i) == 0
    got, b = view(lib, st)
    assert (b.seq_stride, b.qual_stride, b.cigar_stride) == (75, 150, 1) and not b.seq_off and not b.qual_off and not b.cigar_off
    assert np.array_equal(got.cols["record_id"], np.arange(1000, 1040, dtype=np.uint64))
    assert np.array_equal(got.cols["qual"], hb.cols["qual"][:40 * 150])
    # a record without qualities: offsets for SEQ / QUAL, still one operation each
    assert lib.ngsq_stager_push_packed(st, 4, 0, -1, -1, -1, 0, 150, hb.cols["seq"].ctypes.data, None, one.ctypes.data, 1, 2000) == 0
    got, b = view(lib, st)
    assert b.seq_off and b.qual_off and b.cigar_stride == 1 and not b.cigar_off
    assert int(got.cols["qual_off"][41]) == int(got.cols["qual_off"][40]) == 40 * 150
    # two operations: the CIGAR column goes to offsets as well; 70 000 operations saturate the 16-bit column
    big = np.full(70_000, (1 << 4) | 0, dtype=np.uint32)
    assert lib.ngsq_stager_push_packed(st, 0, 0, 0, 5, -1, 0, 0, None, None, big.ctypes.data, len(big), 2001) == 0
    got, b = view(lib, st)
    assert b.cigar_off and int(got.cols["n_cigar"][41]) == 65535 and int(got.cols["cigar_off"][42] - got.cols["cigar_off"][41]) == 70_000
    # NGSQ_STAGE_OFFSETS_ONLY never hands over rows
    lib.ngsq_stager_destroy(st)
    st = new_stager(lib, 10, ffi.STAGE_PAGEABLE | ffi.STAGE_OFFSETS_ONLY)
    assert lib.ngsq_stager_push_packed(st, 0, 60, 0, 100, -1, 0, 150, hb.cols["seq"].ctypes.data, hb.cols["qual"].ctypes.data, one.ctypes.data, 1, ffi.STAGE_NO_ID) == 0
    _, b = view(lib, st)
    assert b.seq_off and b.qual_off and b.cigar_off
    lib.ngsq_stager_destroy(st)


def test_all_0xff_scores_are_no_qualities_in_either_layout(lib, oracle_mod):
    """ngsq_stager_push with l_seq scores of 0xFF is BAM's encoding of "no qualities": staged as n_quals = 0 whatever else the
    flush holds (fixed-pitch rows read such a row as absent, the offsets layout counted l_seq decode errors)."""
    bases = np.array([1, 2, 4, 8] * 5, dtype=np.uint8)
    ff = np.full(20, 0xFF, dtype=np.uint8)
    q = np.full(20, 30, dtype=np.uint8)
    one, two = np.array([20 << 4], dtype=np.uint32), np.array([10 << 4, 10 << 4 | 4], dtype=np.uint32)
    docs = []
    for ragged in (False, True):
        st = new_stager(lib, 8)
        assert lib.ngsq_stager_push(st, 0, 60, 0, 10, -1, 0, 20, bases.ctypes.data, q.ctypes.data, 20, one.ctypes.data, 1, ffi.STAGE_NO_ID) == 0
        assert lib.ngsq_stager_push(st, 0, 60, 0, 11, -1, 0, 20, bases.ctypes.data, ff.ctypes.data, 20, one.ctypes.data, 1, ffi.STAGE_NO_ID) == 0
        if ragged:  # a record of another shape turns the flush into the offsets layout
            assert lib.ngsq_stager_push(st, 0, 60, 0, 12, -1, 0, 20, bases.ctypes.data, q.ctypes.data, 20, two.ctypes.data, 2, ffi.STAGE_NO_ID) == 0
        got, b = view(lib, st)
        assert int(got.cols["qual_off"][2] - got.cols["qual_off"][1]) == 0     # the 0xFF record holds no score bytes
        orc = oracle_mod.Oracle(LENS, PRIMARY, facets=ffi.FACET_QUALITY_SCORE, max_read_len=32)
        orc.process_batch(got.slice(0, 2))
        assert orc.finalize(allow_malformed=True) == 0                          # no bad_quality_score
        docs.append(orc.results(NAMES))
        lib.ngsq_stager_destroy(st)
    json_equal(docs[0], docs[1])
    # a read longer than BAM's l_seq field is refused before anything is sized from it
    st = new_stager(lib, 2)
    assert lib.ngsq_stager_push_packed(st, 0, 0, -1, -1, -1, 0, 0xFFFFFFFF, bases.ctypes.data, None, None, 0, ffi.STAGE_NO_ID) == ffi.ERR_INVALID_ARGUMENT
    assert b"l_seq" in lib.ngsq_stager_last_error(st)
    lib.ngsq_stager_destroy(st)


def test_errors_and_lifecycle(lib):
    st = C.c_void_p()
    assert lib.ngsq_stager_create(0, ffi.STAGE_PAGEABLE, C.byref(st)) == ffi.ERR_INVALID_ARGUMENT
    assert lib.ngsq_stager_create(8, 0x80, C.byref(st)) == ffi.ERR_INVALID_ARGUMENT
    if lib.ngsq_device_count() < 1:   # pinned columns need the device: no silent fallback to ordinary memory
        assert lib.ngsq_stager_create(8, ffi.STAGE_PINNED, C.byref(st)) == ffi.ERR_NO_DEVICE
        assert b"NGSQ_STAGE_PAGEABLE" in lib.ngsq_stager_last_error(None)
    st = new_stager(lib, 2)
    bases = np.array([1, 2, 4, 8, 15], dtype=np.uint8)
    q = np.array([30, 31, 32, 33, 34], dtype=np.uint8)
    op = np.array([5 << 4], dtype=np.uint32)
    push = lambda b_=bases, q_=q, nq=5, rid=ffi.STAGE_NO_ID: lib.ngsq_stager_push(  # noqa: E731
        st, 0, 60, 0, 10, -1, 0, 5, b_.ctypes.data, q_.ctypes.data, nq, op.ctypes.data, 1, rid)
    assert push(b_=np.array([1, 2, 16, 8, 15], dtype=np.uint8)) == ffi.ERR_INVALID_ARGUMENT          # not a 4-bit code
    assert b"4-bit" in lib.ngsq_stager_last_error(st)
    assert push(nq=3) == ffi.ERR_INVALID_ARGUMENT                                                       # 5 bases, 3 scores
    assert push() == 0
    assert push(rid=77) == ffi.ERR_INVALID_ARGUMENT                                                     # ids: all or none per flush
    assert push() == 0
    assert push() == ffi.ERR_STATE and b"full" in lib.ngsq_stager_last_error(st)                       # flush first
    assert lib.ngsq_stager_rewind(st, 0) == ffi.ERR_STATE                                               # not with records staged
    got, b = view(lib, st)
    assert got.n == 2 and np.array_equal(got.cols["seq"][:3], np.array([0x12, 0x48, 0xF0], dtype=np.uint8))
    assert lib.ngsq_stager_len(st) == 2 and lib.ngsq_stager_capacity(st) == 2 and lib.ngsq_stager_pushed(st) == 2
    lib.ngsq_stager_destroy(st)
    lib.ngsq_stager_destroy(None)


# ---------------------------------------------------------------------------------------------------------------------
# GPU: the C ABI driven in the reference's call shape by a C program
# ---------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def driver(lib, tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("stager") / "stager_drive")
    lib_dir = os.path.dirname(build.OUT)
    subprocess.run(["gcc", "-std=c99", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c", "stager_drive.c"),
                    "-L", lib_dir, "-lngsq", "-Wl,-rpath," + lib_dir, "-Wl,-rpath-link,/opt/rocm/lib", "-o", exe], check=True)
    return exe


def test_the_driver_is_plain_c_against_the_headers(driver):
    """(CPU) tests/c/stager_drive.c compiles as C99 with -Wall -Werror against include/*.h and links libngsq.so."""
    assert os.access(driver, os.X_OK)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [None, 0, 1, 7, 300])
@pytest.mark.parametrize("shape", ["uniform150", "ragged"])
def test_reference_call_shape(driver, gpu_lib, oracle_mod, tmp_path, shape, n):
    """pass 1: push per record with NGSQ_PASS_RECORD, flush at capacity and at `summarize`; pass 2: per sequence setup -> region
    query through the index -> push per record with NGSQ_PASS_SEQUENCE -> flush at `teardown`; -n: the counter rules of
    command.rs:305-316 (display.rs:58-63) and :354,384-388.  Against the oracle's emulation of the same driver, and -- without
    -n -- against the batch path (`ngs qc`) on the same file."""
    from tests.test_cli import sorted_batch
    hb = sorted_batch(3, 4000) if shape == "uniform150" else sorted_batch(4, 4000, max_len=260, min_len=30)
    bam = str(tmp_path / "s.bam")
    hb = bamio.with_ids(hb, bamio.write_bam(bam, hb, NAMES, LENS, block_payload=3000, real_index=True))
    out = str(tmp_path / "drive.json")
    args = [driver, bam, out, "".join(str(p) for p in PRIMARY), "977"] + ([str(n)] if n is not None else [])
    r = subprocess.run(args, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = json.load(open(out))
    c = hb.cols
    if n is None:
        pass1, picks = hb, None
    else:
        pass1 = hb.slice(0, min(max(n, 1), hb.n))
        picks, counter = [], 0
        for ref in range(len(NAMES)):
            for i in range(hb.n):
                if c["ref_id"][i] != ref or c["pos"][i] < 0:
                    continue
                ops = c["cigar"][int(c["cigar_off"][i]):int(c["cigar_off"][i + 1])]
                span = sum(int(x) >> 4 for x in ops if (int(x) & 15) in (0, 2, 3, 7, 8))
                s = int(c["pos"][i]) + 1
                if s + span - 1 == 0 or s > LENS[ref]:
                    continue
                picks.append(i)
                counter += 1
                if counter >= n:
                    break
    o = oracle_mod.Oracle(LENS, PRIMARY, facets=ffi.FACETS_DEFAULT, max_read_len=1024, gc_seed=0x4E4753)
    if picks is None:
        o.process_batch(hb)
    else:
        o.process_batch(pass1, ffi.PASS_RECORD)
        for i in picks:
            o.process_batch(hb.slice(i, i + 1), ffi.PASS_SEQUENCE)
    o.finalize()
    json_equal(got, o.results(NAMES))
    o.close()
    assert "flushes" in r.stderr and int(r.stderr.split("flushes")[0].split()[-1]) >= (5 if n is None else 1)
    if n is None:   # and the batch path on the same file
        ngs = build.build_cli(verbose=False)
        r2 = subprocess.run([ngs, "-q", "qc", bam, "GRCh38_no_alt_AnalysisSet", "-o", str(tmp_path)], capture_output=True, text=True)
        assert r2.returncode == 0, r2.stderr
        json_equal(got, json.load(open(tmp_path / "s.bam.results.json")))
