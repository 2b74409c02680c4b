"""Test-side BAM / BAI writer (pure Python + zlib): turns a HostBatch into a real BGZF-compressed
BAM file and a minimal, well-formed BAI, so that the C++ ingest (include/ngsq_bam.h) and the
`ngs qc` command line can be exercised end to end.  SAM/BAM specification 4.1, 4.2, 5.2."""
from __future__ import annotations

import struct
import zlib
from typing import List, Sequence

import numpy as np

EOF_BLOCK = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def bgzf_block(data: bytes, level: int = 1) -> bytes:
    assert len(data) <= 65280
    co = zlib.compressobj(level, zlib.DEFLATED, -15)
    comp = co.compress(data) + co.flush()
    bsize = len(comp) + 25
    return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", bsize) + comp +
            struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))


def reg2bin(beg: int, end: int) -> int:
    end -= 1
    for shift, off in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        if beg >> shift == end >> shift:
            return off + (beg >> shift)
    return 0


def record_bytes(hb, i: int, name: bytes = b"r") -> bytes:
    c = hb.cols
    l = int(c["l_seq"][i])
    if c["seq_off"] is not None:
        seq = c["seq"][int(c["seq_off"][i]):int(c["seq_off"][i + 1])].tobytes()
        q = c["qual"][int(c["qual_off"][i]):int(c["qual_off"][i + 1])].tobytes()
    else:
        seq = c["seq"][i * hb.seq_stride:i * hb.seq_stride + (l + 1) // 2].tobytes()
        q = c["qual"][i * hb.qual_stride:i * hb.qual_stride + l].tobytes()
    if len(q) != l:
        q = b"\xff" * l  # absent qualities
    if c["cigar_off"] is not None:
        cig = c["cigar"][int(c["cigar_off"][i]):int(c["cigar_off"][i + 1])]
    else:
        cig = c["cigar"][i * hb.cigar_stride:i * hb.cigar_stride + int(c["n_cigar"][i])]
    nm = name + b"%d" % i + b"\0"
    pos = int(c["pos"][i])
    span = sum(int(x) >> 4 for x in cig if (int(x) & 15) in (0, 2, 3, 7, 8))
    bin_ = reg2bin(max(pos, 0), max(pos, 0) + max(span, 1))
    body = struct.pack("<iiBBHHHIiii", int(c["ref_id"][i]), pos, len(nm), int(c["mapq"][i]), bin_, len(cig),
                       int(c["flag"][i]), l, int(c["mate_ref_id"][i]), -1, int(c["tlen"][i]))
    body += nm + np.asarray(cig, dtype="<u4").tobytes() + seq + q
    return struct.pack("<I", len(body)) + body


def _ref_span(hb, i: int) -> int:
    o0, o1 = int(hb.cols["cigar_off"][i]), int(hb.cols["cigar_off"][i + 1])
    return sum(int(c) >> 4 for c in hb.cols["cigar"][o0:o1] if (int(c) & 15) in (0, 2, 3, 7, 8))


def write_bai(path: str, hb, n_refs: int, rec_voff: Sequence[int], end_voff: int) -> None:
    """A real BAI (SAM spec 5.2) for the records of `hb` written at the virtual offsets `rec_voff`: binning index
    (adjacent records of a bin merged into one chunk) and 16 kb linear index, as samtools index writes them."""
    bins = [dict() for _ in range(n_refs)]
    lin = [dict() for _ in range(n_refs)]
    n_no_coor = 0
    for i in range(hb.n):
        r, pos = int(hb.cols["ref_id"][i]), int(hb.cols["pos"][i])
        if r < 0 or pos < 0:
            n_no_coor += 1
            continue
        end = pos + max(_ref_span(hb, i), 1)
        v0, v1 = rec_voff[i], rec_voff[i + 1] if i + 1 < hb.n else end_voff
        chunks = bins[r].setdefault(reg2bin(pos, end), [])
        if chunks and chunks[-1][1] == v0:
            chunks[-1][1] = v1
        else:
            chunks.append([v0, v1])
        for w in range(pos >> 14, ((end - 1) >> 14) + 1):
            lin[r][w] = min(lin[r].get(w, v0), v0)
    with open(path, "wb") as f:
        f.write(b"BAI\1" + struct.pack("<i", n_refs))
        for r in range(n_refs):
            f.write(struct.pack("<i", len(bins[r])))
            for b in sorted(bins[r]):
                f.write(struct.pack("<Ii", b, len(bins[r][b])))
                for c0, c1 in bins[r][b]:
                    f.write(struct.pack("<QQ", c0, c1))
            n_intv = max(lin[r]) + 1 if lin[r] else 0
            f.write(struct.pack("<i", n_intv))
            last = 0
            for w in range(n_intv):
                last = lin[r].get(w, last)
                f.write(struct.pack("<Q", last))
        f.write(struct.pack("<Q", n_no_coor))


def write_bam(path: str, hb, ref_names: Sequence[str], ref_len: Sequence[int], block_payload: int = 60000,
              with_index: bool = True, sort_order: str = "coordinate", real_index: bool = False) -> np.ndarray:
    """Returns the records' BAM virtual offsets (block file offset << 16 | offset in the block's data): the ids the
    readers of include/ngsq_bam.h give them (ngsq_batch.record_id)."""
    text = f"@HD\tVN:1.6\tSO:{sort_order}\n" + "".join(f"@SQ\tSN:{n}\tLN:{l}\n" for n, l in zip(ref_names, ref_len))
    head = b"BAM\1" + struct.pack("<i", len(text)) + text.encode() + struct.pack("<i", len(ref_names))
    for n, l in zip(ref_names, ref_len):
        head += struct.pack("<i", len(n) + 1) + n.encode() + b"\0" + struct.pack("<i", l)
    out: List[bytes] = []
    while len(head) > block_payload:  # a long header (thousands of @SQ lines) takes blocks of its own
        out.append(bgzf_block(head[:block_payload]))
        head = head[block_payload:]
    cur = bytearray(head)
    rec_at = []  # (block number, offset in the block's data) of every record
    for i in range(hb.n):
        rec = record_bytes(hb, i)
        if len(cur) >= block_payload:  # a record starts in the block that holds its first byte
            out.append(bgzf_block(bytes(cur)))
            cur = bytearray()
        rec_at.append((len(out), len(cur)))
        while len(cur) + len(rec) > block_payload:  # records may straddle blocks
            take = block_payload - len(cur)
            cur += rec[:take]
            rec = rec[take:]
            out.append(bgzf_block(bytes(cur)))
            cur = bytearray()
        cur += rec
    if cur:
        out.append(bgzf_block(bytes(cur)))
    out.append(EOF_BLOCK)
    with open(path, "wb") as f:
        f.write(b"".join(out))
    starts = [0]
    for blk in out:
        starts.append(starts[-1] + len(blk))
    voff = np.array([(starts[k] << 16) | u for k, u in rec_at], dtype=np.uint64)
    if with_index and real_index:
        write_bai(path + ".bai", hb, len(ref_names), [(starts[k] << 16) | u for k, u in rec_at], starts[len(out) - 1] << 16)
    elif with_index:
        # a structurally valid BAI without bins (the hot path scans the file once; the index is
        # only required to exist and parse: utils/formats/bam.rs:86-96)
        with open(path + ".bai", "wb") as f:
            f.write(b"BAI\1" + struct.pack("<i", len(ref_names)))
            for _ in ref_names:
                f.write(struct.pack("<ii", 0, 0))
            f.write(struct.pack("<Q", 0))
    return voff


def with_ids(hb, ids):
    """`hb` with the record_id column set (what the oracle must be given to see the records as a file reader does)."""
    import copy
    out = copy.copy(hb)
    out.cols = dict(hb.cols)
    out.cols["record_id"] = np.ascontiguousarray(ids, dtype=np.uint64)
    assert out.cols["record_id"].size == hb.n
    return out
