"""Device ingest: the HIP BGZF inflate and BAM record parse against
zlib / the host reader on the same bytes.  Bit-exact."""
import ctypes as C
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
    # long codes: a very skewed alphabet forces 15-bit Huffman codes
    w = np.array([2.0 ** -(i // 8) for i in range(256)])
    yield "longcodes", rng.choice(np.arange(256, dtype=np.uint8), 400_000, p=w / w.sum()).tobytes()


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
    # framing
    with pytest.raises(RuntimeError, match="not a BGZF block"):
        device_inflate(gpu_lib, ctx, b"\x00" * 64)
    with pytest.raises(RuntimeError, match="truncated"):
        device_inflate(gpu_lib, ctx, good[:-5])
