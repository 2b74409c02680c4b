"""Device ingest: the HIP BGZF inflate and BAM record parse against
zlib / the host reader on the same bytes.  Bit-exact."""
import ctypes as C
import os
import struct
import zlib

import numpy as np
import pytest

from ngs_amd import ffi, host

pytestmark = pytest.mark.gpu


def bgzf_block(data: bytes, level: int = 6, strategy: int = zlib.Z_DEFAULT_STRATEGY, raw: bytes | None = None) -> bytes:
    """One BGZF block (SAM/BAM spec 4.1) holding `data` (<= 64 KiB)."""
    if raw is None:
        co = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
        raw = co.compress(data) + co.flush()
    bsize = 18 + len(raw) + 8
    assert bsize <= 65536
    return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", bsize - 1) + raw +
            struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))


def bgzf(data: bytes, block: int = 0xFF00, **kw) -> bytes:
    out = [bgzf_block(data[i:i + block], **kw) for i in range(0, len(data), block)]
    return b"".join(out) + bgzf_block(b"")


def device_inflate(lib, ctx, comp: bytes, check_crc: bool = True):
    n = C.c_uint64(0)
    cap = 1 << 16
    for _ in range(2):
        out = np.empty(cap, np.uint8)
        rc = lib.ngsq_bgzf_inflate_device(ctx._ctx, comp, len(comp), out.ctypes.data, cap, C.byref(n), int(check_crc))
        if rc == 0:
            return bytes(out[:n.value])
        if n.value <= cap:
            raise RuntimeError((lib.ngsq_last_error(ctx._ctx) or b'').decode())
        cap = n.value
    raise RuntimeError((lib.ngsq_last_error(ctx._ctx) or b'').decode())


@pytest.fixture(scope="module")
def ctx(gpu_lib):
    c = host.QcContext([1000], [1], lib=gpu_lib)
    yield c
    c.close()


def payloads():
    rng = np.random.default_rng(5)
    text = (b"@HD\tVN:1.6\tSO:coordinate\n" + b"".join(b"read%07d\tACGTACGTTTGACCA\tIIIIIHHHGGG#\n" % i for i in range(9000)))
    yield "text", text
    yield "random", rng.integers(0, 256, 200_000, dtype=np.uint8).tobytes()
    yield "skewed", rng.choice(np.array([2, 11, 25, 37], np.uint8), 300_000, p=[.05, .1, .15, .7]).tobytes()
    yield "zeros", bytes(150_000)
    yield "runs", b"".join(bytes([i & 255]) * (1 + (i * 7) % 300) for i in range(2000))
    yield "one", b"x"
    yield "empty", b""
    # far matches: repeats 20 KiB and 31 KiB back (sources that have left the decoder's LDS ring)
    blk = rng.integers(0, 256, 20_000, dtype=np.uint8).tobytes()
    yield "far20k", blk + blk + blk[:15_000]
    blk = rng.integers(0, 256, 31_000, dtype=np.uint8).tobytes()
    yield "far31k", blk + blk + b"tail" * 100
    yield "far_mixed", b"".join(blk[i:i + 700] + bytes(rng.integers(0, 4, 300, dtype=np.uint8)) for i in range(0, 30_000, 700)) * 2
    # long codes: a very skewed alphabet forces 15-bit Huffman codes
    w = np.array([2.0 ** -(i // 8) for i in range(256)])
    yield "longcodes", rng.choice(np.arange(256, dtype=np.uint8), 400_000, p=w / w.sum()).tobytes()
    # short codes: two byte values (1-bit codes with Z_HUFFMAN_ONLY: 128 symbols in a 128-bit window, more than a
    # batch of 64 takes), sixteen (4-bit codes); and both with a rare third value, which gets a long code
    yield "two_values", rng.integers(0, 2, 150_000, dtype=np.uint8).tobytes()
    yield "sixteen_values", rng.integers(0, 16, 150_000, dtype=np.uint8).tobytes()
    rare = rng.integers(0, 2, 200_000, dtype=np.uint8)
    rare[rng.integers(0, rare.size, 40)] = 200
    yield "two_values_and_a_rare_one", rare.tobytes()
    # every literal/length code length from 1 to 15 in use, and second-level tables of several sizes
    w = np.array([2.0 ** -min(i // 2 + 1, 16) for i in range(256)])
    yield "all_lengths", rng.choice(np.arange(256, dtype=np.uint8), 300_000, p=w / w.sum()).tobytes()


@pytest.mark.parametrize("level,strategy", [(1, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_DEFAULT_STRATEGY),
                                            (9, zlib.Z_DEFAULT_STRATEGY), (0, zlib.Z_DEFAULT_STRATEGY),
                                            (6, zlib.Z_FIXED), (6, zlib.Z_HUFFMAN_ONLY), (6, zlib.Z_RLE)])
def test_inflate_matches_zlib(gpu_lib, ctx, level, strategy):
    for name, data in payloads():
        comp = bgzf(data, level=level, strategy=strategy)
        got = device_inflate(gpu_lib, ctx, comp)
        assert got == data, f"{name}: level {level} strategy {strategy}"


def test_inflate_small_blocks_and_block_boundaries(gpu_lib, ctx):
    rng = np.random.default_rng(9)
    data = rng.choice(np.frombuffer(b"ACGT", np.uint8), 500_000).tobytes()
    for block in (1, 7, 255, 4096, 65280):
        comp = bgzf(data[:block * 300], block=block)
        assert device_inflate(gpu_lib, ctx, comp) == data[:block * 300]
    # the largest block the format allows (ISIZE = 65536: the decoder's symbol offsets are 16 bits wide), twice in a row
    for payload in (data[:131072], bytes(131072), bytes(range(256)) * 512):
        comp = bgzf(payload, block=65536)
        assert device_inflate(gpu_lib, ctx, comp) == payload


def test_inflate_mixed_deflate_blocks_in_one_member(gpu_lib, ctx):
    """Several DEFLATE blocks (stored + fixed + dynamic, sync flushes) inside one gzip member."""
    rng = np.random.default_rng(3)
    a = rng.integers(0, 256, 5000, dtype=np.uint8).tobytes()
    b = b"ACGT" * 3000
    c = rng.choice(np.frombuffer(b"ACGTN", np.uint8), 20000).tobytes()
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    raw = co.compress(a) + co.flush(zlib.Z_FULL_FLUSH) + co.compress(b) + co.flush(zlib.Z_SYNC_FLUSH) + co.compress(c) + co.flush()
    comp = bgzf_block(a + b + c, raw=raw) + bgzf_block(b"")
    assert device_inflate(gpu_lib, ctx, comp) == a + b + c


def test_inflate_random_corruptions(gpu_lib, ctx):
    """A flipped byte anywhere in a block's DEFLATE data: an error, or (the flip did not matter, or the CRC is not looked at)
    some output of the announced size -- never a hang, a crash, or a silent difference with the CRC on."""
    rng = np.random.default_rng(77)
    w = np.array([2.0 ** -(i // 8) for i in range(256)])
    sources = [b"read%07d\tACGTACGTTTGACCA\tIIIIIHHHGGG#\n" * 1500,
               rng.choice(np.array([2, 11, 25, 37], np.uint8), 60_000, p=[.05, .1, .15, .7]).tobytes(),
               rng.choice(np.arange(256, dtype=np.uint8), 60_000, p=w / w.sum()).tobytes(),
               rng.integers(0, 2, 50_000, dtype=np.uint8).tobytes(), bytes(40_000)]
    n_err = n_same = 0
    for data in sources:
        for level, strategy in ((6, zlib.Z_DEFAULT_STRATEGY), (1, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_FIXED), (6, zlib.Z_HUFFMAN_ONLY)):
            good = bgzf(data[:60_000], level=level, strategy=strategy)
            first_len = struct.unpack("<H", good[16:18])[0] + 1
            for _ in range(12):
                bad = bytearray(good)
                off = int(rng.integers(18, first_len - 8))   # inside the first block's compressed data
                bad[off] ^= int(rng.integers(1, 256))
                for crc in (True, False):
                    try:
                        got = device_inflate(gpu_lib, ctx, bytes(bad), check_crc=crc)
                    except RuntimeError:
                        n_err += 1
                        continue
                    assert len(got) == len(data[:60_000])
                    if crc:
                        assert got == data[:60_000]
                        n_same += 1
    assert n_err > 100   # (most flips are noticed by the decoder itself or by the CRC)


def test_inflate_errors(gpu_lib, ctx):
    data = b"The quick brown fox jumps over the lazy dog. " * 500
    good = bgzf(data)
    assert device_inflate(gpu_lib, ctx, good) == data
    # CRC mismatch is reported with the block number
    bad = bytearray(good)
    first_len = struct.unpack("<H", good[16:18])[0] + 1
    bad[first_len - 8] ^= 0x55
    with pytest.raises(RuntimeError, match="block 0: CRC mismatch"):
        device_inflate(gpu_lib, ctx, bytes(bad))
    assert device_inflate(gpu_lib, ctx, bytes(bad), check_crc=False) == data
    # corrupt payload: any error, never a hang or a wrong "ok"
    for off in (20, 25, 40, 100):
        bad = bytearray(good)
        bad[off] ^= 0xFF
        try:
            got = device_inflate(gpu_lib, ctx, bytes(bad))
        except RuntimeError:
            continue
        assert got == data  # a flipped bit that still decodes must fail the CRC, so only identical data passes
    # wrong ISIZE
    bad = bytearray(good)
    bad[first_len - 4:first_len] = struct.pack("<I", 100)
    with pytest.raises(RuntimeError, match="block 0"):
        device_inflate(gpu_lib, ctx, bytes(bad))
    # a match that reaches in front of the block's output, met after 12 KiB of literals (beyond the 2 KiB LDS ring, where a
    # wrapped source offset used to be read from HBM): a raw-deflate stream made with a preset dictionary, inflated without it
    rng = np.random.default_rng(9)
    zdict = rng.integers(0, 256, 3000, dtype=np.uint8).tobytes()
    payload = rng.integers(0, 256, 12_000, dtype=np.uint8).tobytes() + zdict[200:2600] + b"end"
    co = zlib.compressobj(6, zlib.DEFLATED, -15, zdict=zdict)
    raw = co.compress(payload) + co.flush()
    assert zlib.decompressobj(-15, zdict=zdict).decompress(raw) == payload
    blk = (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", 18 + len(raw) + 8 - 1) + raw +
           struct.pack("<II", zlib.crc32(payload) & 0xFFFFFFFF, len(payload)))
    with pytest.raises(RuntimeError, match="block 0"):
        device_inflate(gpu_lib, ctx, blk + bgzf_block(b""))
    # output longer than ISIZE says, in the LAST block with data (nothing behind it but the buffer's end): the decoder must stop
    # at ISIZE instead of flushing its ring past it
    long_data = rng.integers(0, 4, 30_000, dtype=np.uint8).tobytes()
    lying = bytearray(bgzf_block(long_data))
    lying[-4:] = struct.pack("<I", 9000)
    with pytest.raises(RuntimeError, match="block 1"):
        device_inflate(gpu_lib, ctx, bgzf_block(data[:5000]) + bytes(lying) + bgzf_block(b""), check_crc=False)
    assert device_inflate(gpu_lib, ctx, good) == data   # and the context still works
    # framing
    with pytest.raises(RuntimeError, match="not a BGZF block"):
        device_inflate(gpu_lib, ctx, b"\x00" * 64)
    with pytest.raises(RuntimeError, match="truncated"):
        device_inflate(gpu_lib, ctx, good[:-5])


# ---- BAM record parse on the device ------------------------------------------------------------
from tests import bamio  # noqa: E402
from tests.test_bam_ingest import read_all, records_of  # noqa: E402
from tests.util import random_batch  # noqa: E402


def download_batch(lib, ctx, b: ffi.Batch) -> host.HostBatch:
    n = int(b.n_records)

    def arr(ptr, count, dt):
        if not ptr:
            return None
        a = np.empty(count, dtype=dt)
        if count:
            assert lib.ngsq_memcpy_d2h(ctx._ctx, a.ctypes.data, ptr, a.nbytes) == 0
        return a

    assert b.location == ffi.MEM_DEVICE
    cols = {k: arr(getattr(b, k), n, host.COLUMN_DTYPES[k]) for k in host.FIXED_COLUMNS}
    cols["record_id"] = arr(b.record_id, n, np.uint64)
    for data, off, tot in (("seq", "seq_off", b.seq_bytes), ("qual", "qual_off", b.qual_bytes), ("cigar", "cigar_off", b.cigar_ops)):
        cols[off] = arr(getattr(b, off), n + 1, np.uint64) if getattr(b, off) else None
        cols[data] = arr(getattr(b, data), int(tot), host.COLUMN_DTYPES[data])
    return host.HostBatch(n, cols, b.seq_stride, b.qual_stride, b.cigar_stride, int(b.first_record_index))


def read_all_device(lib, ctx, path, max_records):
    h = C.c_void_p()
    assert lib.ngsq_bam_open(path.encode(), 2, C.byref(h)) == 0, lib.ngsq_bam_last_error()
    batches = []
    try:
        while True:
            b = ffi.Batch()
            rc = lib.ngsq_bam_next_batch_device(h, ctx._ctx, max_records, C.byref(b))
            if rc != 0:
                raise RuntimeError(lib.ngsq_bam_last_error().decode())
            if b.n_records == 0:
                break
            batches.append(download_batch(lib, ctx, b))
        n = lib.ngsq_bam_records_read(h)
        read_all_device.last_stats = device_stats(lib, h) if batches else {}
    finally:
        lib.ngsq_bam_close(h)
    return batches, n


def same_batches(a, b):
    assert len(a) == len(b)
    for x, y in zip(a, b):
        assert (x.n, x.seq_stride, x.qual_stride, x.cigar_stride, x.first_record_index) == \
               (y.n, y.seq_stride, y.qual_stride, y.cigar_stride, y.first_record_index)
        for k in x.cols:
            if x.cols[k] is None or y.cols[k] is None:
                assert x.cols[k] is None and y.cols[k] is None, k
            else:
                assert np.array_equal(x.cols[k], y.cols[k]), k


def device_stats(lib, h) -> dict:
    st = ffi.IngestStats()
    assert lib.ngsq_bam_device_stats(h, C.byref(st)) == 0, lib.ngsq_bam_last_error()
    return {k: int(getattr(st, k)) for k, _ in ffi.IngestStats._fields_}


@pytest.mark.parametrize("case", ["ragged", "uniform150", "long", "multiop150", "one_base", "three_bases", "aligner150", "adversarial"])
def test_device_reader_matches_host_reader(gpu_lib, ctx, tmp_path, monkeypatch, case):
    rng = np.random.default_rng(19)
    ref_len = [50_000, 7_000]
    names = aux = None
    if case in ("aligner150", "adversarial"):
        # what an aligner writes around a record (long names, tags), and tag payloads that read as chains of BAM records: the
        # record index must find the file's own chain whatever the bytes in between look like
        from tests.test_bam_ingest import dressed
        hb = random_batch(rng, 6000, ref_len, max_len=150, min_len=150 if case == "aligner150" else 20, weird=case == "adversarial")
        names, aux = dressed(rng, hb, 2, case == "adversarial")
    elif case == "ragged":
        hb = random_batch(rng, 6000, ref_len, max_len=300, weird=True)
    elif case == "uniform150":
        hb = random_batch(rng, 6000, ref_len, max_len=150, min_len=150, weird=False)
    elif case == "multiop150":
        hb = random_batch(rng, 6000, ref_len, max_len=150, min_len=140, weird=True)
    elif case == "one_base":      # fixed-pitch rows narrower than the dword the column kernel writes (found by fuzz_parity --ingest)
        hb = random_batch(rng, 6000, ref_len, max_len=1, min_len=0, weird=False)
    elif case == "three_bases":
        hb = random_batch(rng, 6000, ref_len, max_len=3, min_len=1, weird=True)
    else:
        hb = random_batch(rng, 700, ref_len, max_len=900, min_len=321, weird=False)
    path = str(tmp_path / "t.bam")
    bamio.write_bam(path, hb, ["chr1", "chr2"], ref_len, block_payload=5000, names=names, aux=aux)
    for max_records in (1 << 20, 257):
        _, hbatches, n = read_all(gpu_lib, path, max_records)
        dbatches, dn = read_all_device(gpu_lib, ctx, path, max_records)
        assert dn == n == hb.n
        same_batches(dbatches, hbatches)
    st = read_all_device.last_stats
    if case == "adversarial":     # the fake chains crowd real record starts out of the candidate table: those segments are walked singly
        assert st["walk_one"] > st["chunks"], st
    elif case == "aligner150":    # an aligner's tags do not: only the record cut by a chunk's end takes that path
        assert st["walk_one"] <= st["chunks"], st
    # many small chunks: the cut record at a chunk's end is carried into the next one
    monkeypatch.setenv("NGSQ_INGEST_RAW_MB", "1")
    dbatches, dn = read_all_device(gpu_lib, ctx, path, 1 << 20)
    assert dn == hb.n
    assert [r for b in dbatches for r in records_of(b)] == records_of(hb)


def test_device_reader_synthetic_bam_and_results(gpu_lib, ctx, tmp_path):
    """file -> device ingest -> facet kernels gives the JSON of file -> host ingest -> facet kernels."""
    cfg = host.synth_config(60_000, mode=ffi.SYNTH_MIXED, ref_len=3_000_000)
    p = str(tmp_path / "s.bam")
    assert gpu_lib.ngsq_synth_write_bam(C.byref(cfg), p.encode(), 60_000, 6, 4) == 0
    _, hbatches, n = read_all(gpu_lib, p, 25_000)
    dbatches, dn = read_all_device(gpu_lib, ctx, p, 25_000)
    assert dn == n == 60_000
    same_batches(dbatches, hbatches)

    def run(device: bool):
        q = host.QcContext([3_000_000, 3_000_000], [1, 1], lib=gpu_lib)
        h = C.c_void_p()
        assert gpu_lib.ngsq_bam_open(p.encode(), 2, C.byref(h)) == 0
        while True:
            b = ffi.Batch()
            rc = (gpu_lib.ngsq_bam_next_batch_device(h, q._ctx, 25_000, C.byref(b)) if device
                  else gpu_lib.ngsq_bam_next_batch(h, 25_000, C.byref(b)))
            assert rc == 0
            if b.n_records == 0:
                break
            assert gpu_lib.ngsq_process_batch(q._ctx, C.byref(b), ffi.PASS_BOTH) == 0, gpu_lib.ngsq_last_error(q._ctx)
        gpu_lib.ngsq_bam_close(h)
        q.finalize()
        r = q.results(["chr1", "chr2"])
        q.close()
        return r

    assert run(True) == run(False)


def test_block_cache_serves_the_next_file(gpu_lib, ctx, tmp_path):
    """mem_pool.cpp: the buffers of a finished device ingest are kept for the next file of the process --
    same batches from recycled blocks, ngsq_release_cached_memory() gives them back, and a third scan
    (fresh allocations again) still reads the same."""
    cfg = host.synth_config(40_000, mode=ffi.SYNTH_FIXED, ref_len=3_000_000)
    p = str(tmp_path / "c.bam")
    assert gpu_lib.ngsq_synth_write_bam(C.byref(cfg), p.encode(), 40_000, 6, 4) == 0
    gpu_lib.ngsq_release_cached_memory()
    first, n1 = read_all_device(gpu_lib, ctx, p, 15_000)
    again, n2 = read_all_device(gpu_lib, ctx, p, 15_000)       # from the cache
    released = int(gpu_lib.ngsq_release_cached_memory())
    assert released >= 2 << 20, released                        # at least the raw buffers of the 1 MiB.. chunk were parked
    assert int(gpu_lib.ngsq_release_cached_memory()) == 0       # nothing left
    third, n3 = read_all_device(gpu_lib, ctx, p, 15_000)
    assert n1 == n2 == n3 == 40_000
    same_batches(first, again)
    same_batches(first, third)
    # a different shape next: blocks of the wrong size are not handed out for it
    hb = random_batch(np.random.default_rng(5), 3000, [50_000, 7_000], max_len=300, weird=True)
    q = str(tmp_path / "r.bam")
    bamio.write_bam(q, hb, ["chr1", "chr2"], [50_000, 7_000], block_payload=5000)
    dbatches, dn = read_all_device(gpu_lib, ctx, q, 1 << 20)
    assert dn == hb.n and [r for b in dbatches for r in records_of(b)] == records_of(hb)


def test_device_reader_header_larger_than_segments_and_chunks(gpu_lib, tmp_path, monkeypatch):
    """A header with 20 000 reference sequences (~1.3 MB of BAM header): the first record lies behind twenty 64 KiB
    index segments, and -- with the ingest buffer shrunk to 1 MiB -- behind the whole first chunk."""
    rng = np.random.default_rng(41)
    n_refs = 20_000
    names = [f"contig_{i:05d}_of_a_fragmented_assembly" for i in range(n_refs)]
    ref_len = [3000] * n_refs
    hb = random_batch(rng, 5000, ref_len[:7], max_len=120, weird=False)
    path = str(tmp_path / "h.bam")
    bamio.write_bam(path, hb, names, ref_len, block_payload=30_000)
    with host.QcContext(ref_len, facets=ffi.FACET_GENERAL, lib=gpu_lib) as ctx:
        for raw_mb in ("1024", "1"):
            monkeypatch.setenv("NGSQ_INGEST_RAW_MB", raw_mb)
            dbatches, dn = read_all_device(gpu_lib, ctx, path, 1 << 20)
            assert dn == hb.n
            assert [r for b in dbatches for r in records_of(b)] == records_of(hb)


def test_device_reader_offsets_scan_both_ways(gpu_lib, tmp_path, monkeypatch):
    """The offsets of the variable-width columns are an exclusive scan of the records' lengths (bam_device.hip
    launch_exclusive_scan_u64): pieces of 4096 entries, each block adding up the sums in front of its own piece -- or, beyond
    8192 pieces (33 M records in one chunk), a launch that scans the sums between the two.  Both on 30 000 ragged records (eight
    pieces), and with chunks of 1 MiB (pieces that end inside a chunk)."""
    rng = np.random.default_rng(43)
    ref_len = [50_000, 9000]
    hb = random_batch(rng, 30_000, ref_len, max_len=140, weird=False)
    path = str(tmp_path / "s.bam")
    bamio.write_bam(path, hb, ["chr1", "chr2"], ref_len, block_payload=40_000)
    want = records_of(hb)
    with host.QcContext(ref_len, facets=ffi.FACET_GENERAL, lib=gpu_lib) as ctx:
        for own_carry in ("8192", "3", "0"):
            for raw_mb in ("256", "1"):
                monkeypatch.setenv("NGSQ_SCAN_OWN_CARRY", own_carry)
                monkeypatch.setenv("NGSQ_INGEST_RAW_MB", raw_mb)
                dbatches, dn = read_all_device(gpu_lib, ctx, path, 1 << 20)
                assert dn == hb.n and [r for b in dbatches for r in records_of(b)] == want, (own_carry, raw_mb)


def test_device_reader_errors(gpu_lib, ctx, tmp_path):
    rng = np.random.default_rng(1)
    hb = random_batch(rng, 2000, [9000], max_len=80)
    p = str(tmp_path / "x.bam")
    bamio.write_bam(p, hb, ["chr1"], [9000], block_payload=3000)
    data = open(p, "rb").read()
    # a file cut inside a BGZF block
    open(p, "wb").write(data[:len(data) // 2])
    with pytest.raises(RuntimeError, match="truncated"):
        read_all_device(gpu_lib, ctx, p, 1 << 20)
    # host and device calls do not mix on one handle
    open(p, "wb").write(data)
    h = C.c_void_p()
    assert gpu_lib.ngsq_bam_open(p.encode(), 1, C.byref(h)) == 0
    b = ffi.Batch()
    assert gpu_lib.ngsq_bam_next_batch(h, 10, C.byref(b)) == 0
    assert gpu_lib.ngsq_bam_next_batch_device(h, ctx._ctx, 10, C.byref(b)) == ffi.ERR_STATE
    gpu_lib.ngsq_bam_close(h)


def test_device_reader_malformed_record(gpu_lib, ctx, tmp_path):
    """A record whose fields do not fit its block_size is refused by both readers, wherever it sits."""
    import struct
    rng = np.random.default_rng(2)
    hb = random_batch(rng, 1500, [9000], max_len=120, weird=False)
    text = "@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:chr1\tLN:9000\n"
    head = b"BAM\1" + struct.pack("<i", len(text)) + text.encode() + struct.pack("<i", 1)
    head += struct.pack("<i", 5) + b"chr1\0" + struct.pack("<i", 9000)
    recs = [bamio.record_bytes(hb, i) for i in range(hb.n)]
    for victim in (0, 700, 1499):
        bad = list(recs)
        r = bytearray(bad[victim])
        r[4 + 8] = 0                      # l_read_name = 0 (noodles: the read name is never empty)
        bad[victim] = bytes(r)
        stream = head + b"".join(bad)
        p = str(tmp_path / f"bad{victim}.bam")
        with open(p, "wb") as f:
            for k in range(0, len(stream), 4000):
                f.write(bamio.bgzf_block(stream[k:k + 4000]))
            f.write(bamio.EOF_BLOCK)
        open(p + ".bai", "wb").write(b"BAI\1" + struct.pack("<i", 1) + struct.pack("<ii", 0, 0))
        with pytest.raises(RuntimeError, match=f"malformed record {victim}"):
            read_all_device(gpu_lib, ctx, p, 1 << 20)
        h = C.c_void_p()
        assert gpu_lib.ngsq_bam_open(p.encode(), 1, C.byref(h)) == 0
        b = ffi.Batch()
        assert gpu_lib.ngsq_bam_next_batch(h, 1 << 20, C.byref(b)) != 0
        assert f"malformed record {victim}".encode() in gpu_lib.ngsq_bam_last_error()
        gpu_lib.ngsq_bam_close(h)


def test_device_reader_empty_bam(gpu_lib, ctx, tmp_path):
    p = str(tmp_path / "e.bam")
    bamio.write_bam(p, random_batch(np.random.default_rng(0), 0, [100]), ["chr1"], [100])
    batches, n = read_all_device(gpu_lib, ctx, p, 100)
    assert batches == [] and n == 0


# ---- one BAM file, three ranks (sharded device ingest + owner-computes teardown) -----------------
def _file_shard_worker(rank, world, port, q, bam, writer, transport=None, wrong_guess=False):
    try:
        import os
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from ngs_amd import ffi as F, host as H, shard
        from tests.test_shard_gloo import _make_comm

        comm, done = _make_comm(transport or ("shm" if writer in ("synth", "aligner") else "gloo"), rank, world, port)
        lib = F.load_library()
        ref_len = [3_000_000, 3_000_000] if writer in ("synth", "aligner") else [50_000, 7_000]
        names = ["chr1", "chr2"]
        kw = dict(facets=F.FACETS_DEFAULT, bin_size=50_000, max_read_len=1024, gc_seed=5)

        def whole_file():
            c = H.QcContext(ref_len, device=0, lib=lib, **kw)
            h = C.c_void_p()
            assert lib.ngsq_bam_open(bam.encode(), 2, C.byref(h)) == 0
            n = 0
            while True:
                b = F.Batch()
                assert lib.ngsq_bam_next_batch_device(h, c._ctx, 20_000, C.byref(b)) == 0, lib.ngsq_bam_last_error()
                if b.n_records == 0:
                    break
                n += b.n_records
                assert lib.ngsq_process_batch(c._ctx, C.byref(b), F.PASS_BOTH) == 0
            lib.ngsq_bam_close(h)
            c.finalize()
            r = c.results(names)
            c.close()
            return r, n

        want, n_total = whole_file() if rank == 0 else (None, None)
        ctx = H.QcContext(ref_len, device=0, lib=lib, **kw)
        # "wrong-guess": rank 1 starts from a deliberately wrong first record (one record too late): the shards notice
        # when they compare notes, rank 1 is re-armed from rank 0's end and scans again
        hook = None
        wrong_rank = 1 if world <= 3 else 5
        if wrong_guess and rank == wrong_rank:
            def hook(h):
                probe = C.c_void_p()
                assert lib.ngsq_bam_open(bam.encode(), 1, C.byref(probe)) == 0
                assert lib.ngsq_bam_shard_begin(probe, ctx._ctx, wrong_rank, world, 0) == 0, lib.ngsq_bam_last_error()
                b = F.Batch()
                assert lib.ngsq_bam_next_batch_device(probe, ctx._ctx, 2, C.byref(b)) == 0 and b.n_records == 2
                ids = np.empty(2, dtype=np.uint64)
                assert lib.ngsq_memcpy_d2h(ctx._ctx, ids.ctypes.data, b.record_id, 16) == 0
                lib.ngsq_bam_close(probe)
                assert lib.ngsq_bam_shard_begin(h, ctx._ctx, wrong_rank, world, int(ids[1])) == 0, lib.ngsq_bam_last_error()
        info, rounds, mine = comm.scan_file_shard(ctx, bam, batch_records=7_000, begin_hook=hook)
        assert mine == info.n_records
        if writer == "adversarial":   # the test looks at the sum over the ranks
            open(f"{bam}.rounds{rank}", "w").write(str(rounds))
        else:
            assert rounds == (1 if wrong_guess else 0), rounds
        counts = comm.allgather_ints([mine, int(info.first_record_index)])
        if rank == 0:
            assert sum(c for c, _ in counts) == n_total, (counts, n_total)
            run = 0
            for c, f in counts:   # first_record_index: the records of the shards in front
                assert f == run
                run += c
        comm.exchange(ctx)
        ctx.finalize()
        got = ctx.results(names)
        if rank == 0:
            from tests.util import json_equal as jeq
            jeq(got, want)
        ctx.close()
        done()
        q.put((rank, "ok"))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, "FAIL " + traceback.format_exc()))


@pytest.mark.parametrize("writer,transport,wrong", [("synth", None, False), ("straddling", None, False), ("straddling", "rccl-double", False),
                                                    ("straddling", None, True), ("aligner", None, False), ("adversarial", None, False)])
def test_three_ranks_share_one_bam_file(gpu_lib, tmp_path, writer, transport, wrong):
    """Each rank streams its BGZF block range of the same file through the chunked pipeline; the record boundaries the
    shards assumed are compared afterwards (ngsq_bam_shard_verify); results equal the single-reader run.
    "straddling": a file whose records cross every block boundary, so no shard starts at a record start.
    wrong: a shard that started from a wrong first record is found out and scanned again.
    "aligner": the synthetic file dressed as an aligner's output (long names, NM MD MC AS XS MQ RG SA XA tags, 15 % multi-op CIGARs).
    "adversarial": every record carries a B:C array whose bytes are a chain of well-formed BAM records that runs into the next
    real record, and the blocks are small, so a shard's first block starts inside such a payload: the shard's guess of its
    first record is a fake one, its neighbour's end says so, and the shard is scanned again -- not planted, found."""
    import multiprocessing as mp
    import socket
    bam = str(tmp_path / "f.bam")
    if writer in ("synth", "aligner"):
        cfg = host.synth_config(120_000, mode=ffi.SYNTH_MIXED if writer == "synth" else ffi.SYNTH_FIXED, ref_len=3_000_000,
                                file_style=ffi.SYNTH_FILE_REALISTIC if writer == "aligner" else 0)
        assert gpu_lib.ngsq_synth_write_bam(C.byref(cfg), bam.encode(), 120_000, 6, 4) == 0
    elif writer == "adversarial":
        rng = np.random.default_rng(31)
        hb = random_batch(rng, 9000, [50_000, 7_000], max_len=60, min_len=20, weird=False)
        hb.cols["flag"] &= np.uint16(0xFFFF ^ 0x1)
        aux = [bamio.aligner_aux(rng, int(l)) + bamio.aux_array(b"ZF", b"C", bamio.fake_record_chain(2, rng, int(rng.integers(6, 12))))
               for l in hb.cols["l_seq"]]
        bamio.write_bam(bam, hb, ["chr1", "chr2"], [50_000, 7_000], block_payload=1500, names=[bamio.aligner_name(rng) for _ in range(hb.n)], aux=aux)
    else:
        rng = np.random.default_rng(23)
        hb = random_batch(rng, 9000, [50_000, 7_000], max_len=200, weird=False)
        c = hb.cols
        c["flag"] &= np.uint16(0xFFFF ^ 0x1)
        bamio.write_bam(bam, hb, ["chr1", "chr2"], [50_000, 7_000], block_payload=3000)
    from tests.test_shard_gloo import _run_ranks
    _run_ranks(_file_shard_worker, 3, bam, writer, transport, wrong)   # rccl-double: the RCCL transport over tests/rccl_double
    if writer == "adversarial":
        rounds = [int(open(f"{bam}.rounds{r}").read()) for r in range(3)]
        assert max(rounds) >= 1, rounds      # at least one shard's guess was a fake record: found out and scanned again


@pytest.mark.parametrize("writer,transport,wrong", [("synth", None, False), ("straddling", "rccl-double", True), ("tail-heavy", None, False)])
def test_eight_ranks_share_one_bam_file(gpu_lib, tmp_path, writer, transport, wrong):
    """The same with EIGHT ranks (sharing this box's GPU): seven boundaries found independently and compared in one round, a
    wrong first record planted on rank 5 (found out, re-armed from rank 4's end, scanned again), and -- "tail-heavy" -- a file
    whose last eighth holds no record start at all (one record of 300 kb ends the file: rank 7 is an EMPTY shard and passes
    its predecessor's end on)."""
    bam = str(tmp_path / "f.bam")
    if writer == "synth":
        cfg = host.synth_config(320_000, mode=ffi.SYNTH_MIXED, ref_len=3_000_000)
        assert gpu_lib.ngsq_synth_write_bam(C.byref(cfg), bam.encode(), 320_000, 6, 4) == 0
    else:
        rng = np.random.default_rng(29)
        hb = random_batch(rng, 24_000, [50_000, 7_000], max_len=200, weird=False)
        hb.cols["flag"] &= np.uint16(0xFFFF ^ 0x1)
        if writer == "tail-heavy":
            from tests.util import batch_from_records
            from tests.genome_util import concat
            big = batch_from_records([dict(flag=4, ref_id=-1, pos=-1, cigar="*", seq="".join(rng.choice(list("ACGT"), 300_000)),
                                           qual=[int(x) for x in rng.integers(0, 41, 300_000)])])
            hb = concat(hb.slice(0, 1500), big)   # ~200 kB of records, then ONE of 250 kB (compressed): the last ranks' ranges lie inside it
        bamio.write_bam(bam, hb, ["chr1", "chr2"], [50_000, 7_000], block_payload=3000, sort_order="unsorted" if writer == "tail-heavy" else "coordinate")
    from tests.test_shard_gloo import _run_ranks
    _run_ranks(_file_shard_worker, 8, bam, writer, transport, wrong)


def test_shard_begin_end_api(gpu_lib, ctx, tmp_path, monkeypatch):
    """Neighbouring shards agree on the record at their boundary (each found on its own: the assumed first record of
    shard k+1 is where shard k's chain ends), the shards' records add up to the file, their ids are the records'
    virtual offsets whatever the number of shards; a confirmed begin that differs from the assumed one scans the
    shard from there (pushed to the next shard's first record, the shard keeps no record and only passes the chain
    through).  The ingest buffer is shrunk so that every shard spans several chunks."""
    monkeypatch.setenv("NGSQ_INGEST_RAW_MB", "1")
    rng = np.random.default_rng(29)
    hb = random_batch(rng, 8000, [50_000, 7_000], max_len=180, weird=False)
    bam = str(tmp_path / "p.bam")
    voff = bamio.write_bam(bam, hb, ["chr1", "chr2"], [50_000, 7_000], block_payload=2500)
    size = os.path.getsize(bam)

    def scan(shard, n, begin=0, h=None):
        if h is None:
            h = C.c_void_p()
            assert gpu_lib.ngsq_bam_open(bam.encode(), 1, C.byref(h)) == 0
        assert gpu_lib.ngsq_bam_shard_begin(h, ctx._ctx, shard, n, begin) == 0, gpu_lib.ngsq_bam_last_error()
        info = ffi.ShardInfo()
        assert gpu_lib.ngsq_bam_shard_end(h, C.byref(info)) == ffi.ERR_STATE     # not before the scan has ended
        ids = []
        while True:
            b = ffi.Batch()
            assert gpu_lib.ngsq_bam_next_batch_device(h, ctx._ctx, 700, C.byref(b)) == 0, gpu_lib.ngsq_bam_last_error()
            if b.n_records == 0:
                break
            assert b.first_record_index == sum(len(x) for x in ids)     # numbered within the shard
            ids.append(download_batch(gpu_lib, ctx, b).cols["record_id"])
        assert gpu_lib.ngsq_bam_shard_end(h, C.byref(info)) == 0, gpu_lib.ngsq_bam_last_error()
        return h, info, np.concatenate(ids) if ids else np.zeros(0, np.uint64)

    for n in (1, 2, 5, 64):   # 64: more shards than the file has blocks to give each one something
        parts = [scan(k, n) for k in range(n)]
        assert sum(i.n_records for _, i, _ in parts) == hb.n
        assert np.array_equal(np.concatenate([ids for _, _, ids in parts]), voff)
        for k in range(n - 1):
            assert parts[k][1].end_voffset == parts[k + 1][1].begin_voffset, (n, k)
        assert parts[-1][1].end_voffset == (size << 16)
        for _, i, ids in parts:   # sort keys of the shard's first and last record: refID << 32 | pos + 1, unplaced = ~0
            if i.n_records:
                for key, v in ((i.first_key, ids[0]), (i.last_key, ids[-1])):
                    r = int(np.searchsorted(voff, v))
                    ref, pos = int(hb.cols["ref_id"][r]), int(hb.cols["pos"][r])
                    assert key == ((1 << 64) - 1 if ref < 0 else (ref << 32) | (pos + 1))
        if n == 5:
            # move shard 1's begin to the first record of shard 2: nothing left in shard 1
            h1, i1, ids1 = parts[1]
            assert i1.n_records > 0
            _, moved, ids = scan(1, 5, begin=i1.end_voffset, h=h1)
            assert moved.n_records == 0 and ids.size == 0 and moved.begin_voffset == i1.end_voffset == moved.end_voffset
            # and back: the original begin gives the original shard
            _, again, ids = scan(1, 5, begin=i1.begin_voffset, h=h1)
            assert again.n_records == i1.n_records and again.end_voffset == i1.end_voffset and np.array_equal(ids, ids1)
            # one record too late: the shard loses exactly that record
            _, late, ids = scan(1, 5, begin=int(ids1[1]), h=h1)
            assert late.n_records == i1.n_records - 1 and late.begin_voffset == int(ids1[1]) and np.array_equal(ids, ids1[1:])
            assert gpu_lib.ngsq_bam_shard_begin(h1, ctx._ctx, 1, 5, 5) != 0        # in front of the shard
            assert gpu_lib.ngsq_bam_shard_begin(h1, ctx._ctx, 0, 5, i1.begin_voffset) != 0   # shard 0 starts behind the header
        for h, _, _ in parts:
            gpu_lib.ngsq_bam_close(h)
