"""Host ingest (include/ngsq_bam.h, no GPU needed): BGZF inflate + BAM parse -> SoA batches.
The staged batches must hold exactly the records that were written, in both layouts."""
import ctypes as C
import os

import numpy as np
import pytest

from ngs_amd import ffi, host
from tests import bamio
from tests.util import random_batch, to_fixed_stride


def read_all(lib, path, max_records, threads=3):
    h = C.c_void_p()
    rc = lib.ngsq_bam_open(path.encode(), threads, C.byref(h))
    assert rc == 0, lib.ngsq_bam_last_error()
    refs = [(lib.ngsq_bam_ref_name(h, i).decode(), lib.ngsq_bam_ref_len(h, i)) for i in range(lib.ngsq_bam_n_refs(h))]
    batches = []
    while True:
        b = ffi.Batch()
        rc = lib.ngsq_bam_next_batch(h, max_records, C.byref(b))
        assert rc == 0, lib.ngsq_bam_last_error()
        if b.n_records == 0:
            break
        batches.append(copy_batch(b))
    n = lib.ngsq_bam_records_read(h)
    lib.ngsq_bam_close(h)
    return refs, batches, n


def copy_batch(b: ffi.Batch) -> host.HostBatch:
    n = int(b.n_records)

    def arr(ptr, count, dt):
        if not ptr or count == 0:
            return np.zeros(0, dtype=dt) if ptr else None
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(np.ctypeslib.as_ctypes_type(dt))), shape=(count,)).copy()

    cols = {k: arr(getattr(b, k), n, host.COLUMN_DTYPES[k]) for k in host.FIXED_COLUMNS}
    cols["record_id"] = arr(b.record_id, n, np.uint64)
    for data, off, stride, tot in (("seq", "seq_off", b.seq_stride, b.seq_bytes), ("qual", "qual_off", b.qual_stride, b.qual_bytes),
                                   ("cigar", "cigar_off", b.cigar_stride, b.cigar_ops)):
        cols[off] = arr(getattr(b, off), n + 1, np.uint64) if getattr(b, off) else None
        cols[data] = arr(getattr(b, data), int(tot), host.COLUMN_DTYPES[data])
    return host.HostBatch(n, cols, b.seq_stride, b.qual_stride, b.cigar_stride, int(b.first_record_index))


def records_of(hb):
    """Layout-independent view: list of (fixed fields, seq bytes, qual bytes, cigar ops)."""
    out = []
    c = hb.cols
    for i in range(hb.n):
        l = int(c["l_seq"][i])
        if c["seq_off"] is not None:
            s = c["seq"][int(c["seq_off"][i]):int(c["seq_off"][i + 1])].tobytes()
            q = c["qual"][int(c["qual_off"][i]):int(c["qual_off"][i + 1])].tobytes()
        else:
            s = c["seq"][i * hb.seq_stride:i * hb.seq_stride + (l + 1) // 2].tobytes()
            q = c["qual"][i * hb.qual_stride:i * hb.qual_stride + l].tobytes()
            if q == b"\xff" * l:
                q = b""
        nc = int(c["n_cigar"][i])
        if c["cigar_off"] is not None:
            g = tuple(int(x) for x in c["cigar"][int(c["cigar_off"][i]):int(c["cigar_off"][i + 1])])
        else:
            g = tuple(int(x) for x in c["cigar"][i * hb.cigar_stride:i * hb.cigar_stride + nc])
        out.append((tuple(int(c[k][i]) for k in host.FIXED_COLUMNS), s, q, g))
    return out


def dressed(rng, hb, n_ref, adversarial):
    """Read names and auxiliary data for the records of `hb`: what an aligner writes, or (adversarial) payloads that read
    as BAM records themselves (tests/bamio.py: fake_record_chain)."""
    names = [bamio.aligner_name(rng) for _ in range(hb.n)]
    aux = []
    for i in range(hb.n):
        a = bamio.aligner_aux(rng, int(hb.cols["l_seq"][i]), mapped=not (int(hb.cols["flag"][i]) & 4))
        if adversarial and rng.random() < 0.7:
            a += bamio.adversarial_aux(rng, n_ref)
        aux.append(a)
    return names, aux


@pytest.mark.parametrize("case", ["ragged", "uniform150", "long", "aligner150", "adversarial"])
def test_round_trip(lib, tmp_path, case):
    rng = np.random.default_rng(9)
    ref_len = [50_000, 7_000]
    names = aux = None
    if case in ("aligner150", "adversarial"):     # names of 37-39 characters and tags behind the qualities: skipped unread
        hb = random_batch(rng, 3000, ref_len, max_len=150, min_len=150 if case == "aligner150" else 20, weird=case == "adversarial")
        names, aux = dressed(rng, hb, 2, case == "adversarial")
    elif case == "ragged":
        hb = random_batch(rng, 3000, ref_len, max_len=300, weird=True)
    elif case == "uniform150":
        hb = random_batch(rng, 3000, ref_len, max_len=150, min_len=150, weird=False)
    else:
        hb = random_batch(rng, 400, ref_len, max_len=900, min_len=321, weird=False)
    # qualities == 0xFF for a whole read mean "absent" in BAM: keep real scores <= 93 (random_batch does)
    path = str(tmp_path / "t.bam")
    voff = bamio.write_bam(path, hb, ["chr1", "chr2"], ref_len, block_payload=5000, names=names, aux=aux)
    assert lib.ngsq_bam_check_index(path.encode()) == 0
    want = records_of(hb)
    for max_records in (1 << 20, 257):
        refs, batches, n = read_all(lib, path, max_records)
        assert refs == [("chr1", 50_000), ("chr2", 7_000)] and n == hb.n
        got = [r for b in batches for r in records_of(b)]
        assert got == want
        # ngsq_batch.record_id: every record's BAM virtual offset, as the writer laid the records out
        assert np.array_equal(np.concatenate([b.cols["record_id"] for b in batches]), voff)
        assert [b.first_record_index for b in batches] == list(np.cumsum([0] + [b.n for b in batches[:-1]]))
    if case == "uniform150":
        assert batches[0].seq_stride == 75 and batches[0].qual_stride == 150
    if case == "long":
        assert batches[0].qual_stride == 0 and batches[0].cols["qual_off"] is not None


def test_header_of_many_blocks(lib, tmp_path):
    """20 000 @SQ lines: the header (text + reference table, ~1.3 MB) spans twenty BGZF blocks."""
    rng = np.random.default_rng(41)
    n_refs = 20_000
    names = [f"contig_{i:05d}_of_a_fragmented_assembly" for i in range(n_refs)]
    ref_len = [3000] * n_refs
    hb = random_batch(rng, 3000, ref_len[:7], max_len=120, weird=False)
    path = str(tmp_path / "h.bam")
    bamio.write_bam(path, hb, names, ref_len, block_payload=30_000)
    refs, batches, n = read_all(lib, path, 1000)
    assert refs == list(zip(names, ref_len)) and n == hb.n
    assert [r for b in batches for r in records_of(b)] == records_of(hb)


def test_errors(lib, tmp_path):
    p = str(tmp_path / "x.bam")
    h = C.c_void_p()
    assert lib.ngsq_bam_open(p.encode(), 1, C.byref(h)) != 0 and b"opening BAM file" in lib.ngsq_bam_last_error()
    open(p, "wb").write(b"not a bam at all, definitely not....")
    assert lib.ngsq_bam_open(p.encode(), 1, C.byref(h)) != 0
    assert lib.ngsq_bam_check_index(p.encode()) != 0 and b"reading BAM index" in lib.ngsq_bam_last_error()
    rng = np.random.default_rng(1)
    hb = random_batch(rng, 200, [9000], max_len=80)
    bamio.write_bam(p, hb, ["chr1"], [9000])
    data = open(p, "rb").read()
    open(p, "wb").write(data[:len(data) // 2])  # truncated file
    rc = lib.ngsq_bam_open(p.encode(), 2, C.byref(h))
    if rc == 0:
        b = ffi.Batch()
        rc = lib.ngsq_bam_next_batch(h, 1 << 20, C.byref(b))
        lib.ngsq_bam_close(h)
    assert rc != 0


def test_empty_bam(lib, tmp_path):
    p = str(tmp_path / "e.bam")
    bamio.write_bam(p, random_batch(np.random.default_rng(0), 0, [100]), ["chr1"], [100])
    refs, batches, n = read_all(lib, p, 100)
    assert refs == [("chr1", 100)] and batches == [] and n == 0


def test_synthetic_bam_writer_round_trip(lib, tmp_path):
    """ngsq_synth_write_bam -> ngsq_bam_next_batch gives back exactly the generator's records."""
    for mode in (ffi.SYNTH_FIXED, ffi.SYNTH_MIXED):
        cfg = host.synth_config(20_000, mode=mode, ref_len=3_000_000)
        p = str(tmp_path / f"s{mode}.bam")
        assert lib.ngsq_synth_write_bam(C.byref(cfg), p.encode(), 20_000, 1, 4) == 0
        assert lib.ngsq_bam_check_index(p.encode()) == 0
        refs, batches, n = read_all(lib, p, 7000)
        assert n == 20_000 and refs[0] == ("chr1", 3_000_000)
        got = [r for b in batches for r in records_of(b)]
        assert got == records_of(host.synth_host_batch(cfg, 0, 20_000, lib))
        if mode == ffi.SYNTH_FIXED:
            assert batches[0].qual_stride == 150 and batches[0].cigar_stride == 1
        # the same records dressed as an aligner's output (names of 37-39 characters, NM MD MC AS XS MQ RG [SA XA B]): the
        # batches are those of the plain file, the file is a third larger
        acfg = host.synth_config(20_000, mode=mode, ref_len=3_000_000, file_style=ffi.SYNTH_FILE_ALIGNER)
        q = str(tmp_path / f"a{mode}.bam")
        assert lib.ngsq_synth_write_bam(C.byref(acfg), q.encode(), 20_000, 1, 4) == 0
        assert lib.ngsq_bam_check_index(q.encode()) == 0
        _, abatches, an = read_all(lib, q, 7000)
        assert an == 20_000 and [r for b in abatches for r in records_of(b)] == got
        assert os.path.getsize(q) > 1.15 * os.path.getsize(p)
    # NGSQ_SYNTH_FILE_CIGAR_MIX: 15 % of the mapped fixed-length reads carry a clip or an indel instead of <l>M
    rcfg = host.synth_config(20_000, ref_len=3_000_000, file_style=ffi.SYNTH_FILE_REALISTIC)
    r = str(tmp_path / "r.bam")
    assert lib.ngsq_synth_write_bam(C.byref(rcfg), r.encode(), 20_000, 6, 4) == 0
    _, rb, rn = read_all(lib, r, 1 << 20)
    recs = records_of(rb[0])
    plain = records_of(host.synth_host_batch(rcfg, 0, 20_000, lib))
    assert rn == 20_000 and rb[0].qual_stride == 150 and rb[0].cols["cigar_off"] is not None
    multi = 0
    for a, b in zip(recs, plain):
        fa, fb = dict(zip(host.FIXED_COLUMNS, a[0])), dict(zip(host.FIXED_COLUMNS, b[0]))
        assert a[1:3] == b[1:3] and {k: v for k, v in fa.items() if k != "n_cigar"} == {k: v for k, v in fb.items() if k != "n_cigar"}
        read_bases = sum(c >> 4 for c in a[3] if (c & 15) in (0, 1, 4, 7, 8))
        assert read_bases == (150 if a[3] else 0) and len(a[3]) == fa["n_cigar"]
        multi += len(a[3]) > 1
    assert 0.12 * rn < multi < 0.18 * rn


def test_index_region_query_by_seek(lib, tmp_path):
    """The whole-sequence region query of the sequence pass (command.rs:369-373): the BAI gives the first chunk of a
    sequence, ngsq_bam_seek continues the reader there, and the records that follow are the sequence's, in file order."""
    from tests.util import coordinate_sorted
    rng = np.random.default_rng(12)
    ref_len = [40_000, 500, 25_000, 9_000]              # nothing will map to sequence 1
    hb = random_batch(rng, 6000, ref_len, max_len=120, weird=False)
    hb.cols["ref_id"][hb.cols["ref_id"] == 1] = 2
    hb = coordinate_sorted(hb)
    names = [f"s{i}" for i in range(4)]
    p = str(tmp_path / "q.bam")
    voff = bamio.write_bam(p, hb, names, ref_len, block_payload=2500, real_index=True)   # records straddle many small blocks
    assert lib.ngsq_bam_check_index(p.encode()) == 0
    starts = (C.c_uint64 * 4)()
    bins = C.c_uint64()
    assert lib.ngsq_bam_index_ref_starts(p.encode(), 4, starts, C.byref(bins)) == 0
    assert bins.value > 4 and starts[1] == 0 and all(starts[r] for r in (0, 2, 3))
    ref = hb.cols["ref_id"]
    h = C.c_void_p()
    assert lib.ngsq_bam_open(p.encode(), 2, C.byref(h)) == 0
    for r in (3, 0, 2):                                  # any order: every query is a seek
        assert lib.ngsq_bam_seek(h, starts[r]) == 0, lib.ngsq_bam_last_error()
        first = int(np.argmax((ref == r) & (hb.cols["pos"] >= 0)))
        b = ffi.Batch()
        assert lib.ngsq_bam_next_batch(h, 50, C.byref(b)) == 0
        got = copy_batch(b)
        n_same = min(50, int((ref[first:first + 50] == r).sum()))
        for col in ("ref_id", "pos", "flag", "tlen"):
            np.testing.assert_array_equal(got.cols[col][:n_same], hb.cols[col][first:first + n_same], err_msg=f"{col} of sequence {r}")
        # a record's id does not depend on how the reader got to it
        np.testing.assert_array_equal(got.cols["record_id"][:n_same], voff[first:first + n_same])
    assert lib.ngsq_bam_seek(h, (10 ** 9) << 16) == 0     # beyond the end of the file: nothing follows
    b = ffi.Batch()
    assert lib.ngsq_bam_next_batch(h, 50, C.byref(b)) == 0 and b.n_records == 0
    lib.ngsq_bam_close(h)
    # an index without bins: nothing to look up
    bamio.write_bam(p, hb, names, ref_len)
    assert lib.ngsq_bam_index_ref_starts(p.encode(), 4, starts, C.byref(bins)) == 0 and bins.value == 0


def _parse_bai(path, n_refs):
    import struct
    d = open(path, "rb").read()
    assert d[:4] == b"BAI\1" and struct.unpack_from("<i", d, 4)[0] == n_refs
    q, out = 8, []
    for _ in range(n_refs):
        (n_bin,) = struct.unpack_from("<i", d, q)
        q += 4
        bins = {}
        for _ in range(n_bin):
            b, n_chunk = struct.unpack_from("<Ii", d, q)
            q += 8
            bins[b] = [struct.unpack_from("<QQ", d, q + 16 * k) for k in range(n_chunk)]
            q += 16 * n_chunk
        (n_intv,) = struct.unpack_from("<i", d, q)
        q += 4
        lin = list(struct.unpack_from(f"<{n_intv}Q", d, q))
        q += 8 * n_intv
        out.append((bins, lin))
    assert q + 8 == len(d)
    return out, struct.unpack_from("<Q", d, q)[0]


def test_synthetic_bam_writer_index_follows_the_spec(lib, tmp_path):
    """The BAI of ngsq_synth_write_bam (SAM spec 5.2): every placed record lies inside a chunk of ITS bin
    (reg2bin of [pos, end)), chunks are record-aligned and ordered, and the linear index of a 16 kb window holds the
    virtual offset of the first record that overlaps it."""
    import struct
    import zlib
    n = 12_000
    cfg = host.synth_config(n, mode=ffi.SYNTH_MIXED, ref_len=700_000)
    p = str(tmp_path / "i.bam")
    assert lib.ngsq_synth_write_bam(C.byref(cfg), p.encode(), n, 1, 3) == 0
    # virtual offset of every record: walk the BGZF blocks and the records inside them (whole records per block here)
    raw = open(p, "rb").read()
    voffs, off, first = [], 0, True
    while off < len(raw):
        bsize = struct.unpack_from("<H", raw, off + 16)[0] + 1
        data = zlib.decompress(raw[off + 18:off + bsize - 8], -15)
        u = 0
        if first:   # header block: magic, text, references
            first = False
            u = len(data)
        while u < len(data):
            voffs.append((off << 16) | u)
            u += 4 + struct.unpack_from("<i", data, u)[0]
        off += bsize
    assert len(voffs) == n
    hb = host.synth_host_batch(cfg, 0, n, lib)
    index, n_no_coor = _parse_bai(p + ".bai", 2)
    c = hb.cols
    placed = 0
    first_in_window = {}
    for i in range(n):
        r, pos = int(c["ref_id"][i]), int(c["pos"][i])
        if r < 0 or pos < 0:
            continue
        placed += 1
        end = pos + max(bamio._ref_span(hb, i), 1)
        chunks = index[r][0].get(bamio.reg2bin(pos, end))
        assert chunks and any(c0 <= voffs[i] < c1 for c0, c1 in chunks), f"record {i} is not in its bin"
        for w in range(pos >> 14, ((end - 1) >> 14) + 1):
            first_in_window.setdefault((r, w), voffs[i])
    assert n_no_coor == n - placed
    vset = set(voffs) | {off << 16 for off in range(0)}  # chunk borders are record starts (or the end of the data)
    for r in range(2):
        bins, lin = index[r]
        for b, chunks in bins.items():
            assert all(c0 < c1 and c0 in vset for c0, c1 in chunks) and chunks == sorted(chunks)
        for w, v in enumerate(lin):
            if (r, w) in first_in_window:
                assert v == first_in_window[(r, w)]
    # and the reader's region query finds the first record of chr1 through it
    starts = (C.c_uint64 * 2)()
    bins_n = C.c_uint64()
    assert lib.ngsq_bam_index_ref_starts(p.encode(), 2, starts, C.byref(bins_n)) == 0 and bins_n.value > 10
    assert starts[0] == voffs[int(np.argmax((c["ref_id"] == 0) & (c["pos"] >= 0)))]


def test_record_id_is_what_the_gc_window_is_drawn_from(lib, oracle_mod):
    """include/ngsq.h: record_id == NULL means first_record_index + i; with the column the oracle takes the ids it is
    given (the readers' virtual offsets) -- same ids, same document; other ids, another GC histogram."""
    rng = np.random.default_rng(77)
    hb = random_batch(rng, 3000, [50_000, 7_000], max_len=240, min_len=120, weird=False)
    hb.first_record_index = 1000

    def gc(batch):
        o = oracle_mod.Oracle([50_000, 7_000], facets=ffi.FACET_GC_CONTENT, max_read_len=1024, gc_seed=3)
        o.process_batch(batch)
        o.finalize()
        return o.results(["chr1", "chr2"])["gc_content"]

    base = gc(hb)
    assert gc(bamio.with_ids(hb, np.arange(1000, 1000 + hb.n))) == base
    other = gc(bamio.with_ids(hb, (np.arange(hb.n, dtype=np.uint64) << np.uint64(16)) | np.uint64(5)))
    assert other["histogram"] != base["histogram"] and other["records"] == base["records"]
