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


def record_bytes(hb, i: int, name: bytes = b"r", aux: bytes = b"", full_name: bytes | None = None) -> bytes:
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
    nm = (full_name if full_name is not None else name + b"%d" % i) + b"\0"
    pos = int(c["pos"][i])
    span = sum(int(x) >> 4 for x in cig if (int(x) & 15) in (0, 2, 3, 7, 8))
    if len(cig) > 65535:   # SAM specification 4.2.2: the placeholder <l_seq>S<span>N in the CIGAR field, the operations in a CG:B,I tag
        aux = aux + aux_array(b"CG", b"I", np.asarray(cig, dtype="<u4").tobytes())
        cig = [(l << 4) | 4, (span << 4) | 3]
    bin_ = reg2bin(pos, pos + max(span, 1)) if pos >= 0 else 4680   # (reg2bin(-1, 0), SAM specification 4.2.1)
    body = struct.pack("<iiBBHHHIiii", int(c["ref_id"][i]), pos, len(nm), int(c["mapq"][i]), bin_, len(cig),
                       int(c["flag"][i]), l, int(c["mate_ref_id"][i]), -1, int(c["tlen"][i]))
    body += nm + np.asarray(cig, dtype="<u4").tobytes() + seq + q + aux
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


def aux_z(tag: bytes, text: bytes) -> bytes:
    return tag + b"Z" + text + b"\0"


def aux_int(tag: bytes, x: int) -> bytes:
    """An integer tag in its smallest type, as htslib writes it."""
    if 0 <= x < 256:
        return tag + b"C" + bytes([x])
    if 0 <= x < 65536:
        return tag + b"S" + struct.pack("<H", x)
    if x >= 0:
        return tag + b"I" + struct.pack("<I", x)
    return tag + b"i" + struct.pack("<i", x)


def aux_array(tag: bytes, sub: bytes, payload: bytes) -> bytes:
    w = {b"c": 1, b"C": 1, b"s": 2, b"S": 2, b"i": 4, b"I": 4, b"f": 4}[sub]
    assert len(payload) % w == 0
    return tag + b"B" + sub + struct.pack("<I", len(payload) // w) + payload


def aligner_name(rng) -> bytes:
    """An Illumina read name, 37-39 characters (instrument:run:flowcell:lane:tile:x:y)."""
    return b"A00741:215:HG7WKDSXX:%d:%d%d%02d:%d:%d" % (rng.integers(1, 5), rng.integers(1, 3), rng.integers(1, 7), rng.integers(1, 79),
                                                        rng.integers(1000, 32624), rng.integers(1000, 37000))


def aligner_aux(rng, l_seq: int, mapped: bool = True) -> bytes:
    """What bwa-mem + samtools fixmate leave on a record: NM MD MC AS XS MQ RG, SA / XA / a B array on a few per cent."""
    out = b""
    if mapped:
        nm = int(rng.choice([0, 0, 0, 1, 1, 2, 5]))
        md = b"%d" % l_seq if nm == 0 or l_seq < 4 else b"%dA%d" % (l_seq // 3, l_seq - l_seq // 3 - 1)
        out += aux_int(b"NM", nm) + aux_z(b"MD", md)
    out += aux_z(b"MC", b"%dM" % max(l_seq, 1))
    if mapped:
        out += aux_int(b"AS", max(0, l_seq - 5 * int(rng.integers(0, 3)))) + aux_int(b"XS", int(rng.choice([0, 0, 19, 77, 300])))
    out += aux_int(b"MQ", int(rng.choice([60, 60, 0, 17]))) + aux_z(b"RG", b"HG7WKDSXX.L00%d.SJNORM0415" % rng.integers(1, 5))
    r = rng.random()
    if r < 0.03:
        out += aux_z(b"SA", b"chr%d,%d,+,%dS%dM,60,1;" % (rng.integers(1, 23), rng.integers(1, 2 * 10 ** 8), l_seq // 2, l_seq - l_seq // 2))
    elif r < 0.06:
        out += aux_z(b"XA", b";".join(b"chr%d,-%d,%dM,%d" % (rng.integers(1, 23), rng.integers(1, 2 * 10 ** 8), l_seq, k) for k in range(int(rng.integers(1, 4)))) + b";")
    elif r < 0.08:
        out += aux_array(b"ZB", b"S", rng.integers(0, 65536, int(rng.integers(0, 41))).astype("<u2").tobytes())
    elif r < 0.09:
        out += aux_int(b"ms", -int(rng.integers(1, 70000))) + b"XTA" + bytes(rng.choice(list(b"URNM"), 1).tolist()) + b"ZfB" + b"f" + struct.pack("<I", 2) + struct.pack("<ff", 0.5, 1e9)
    return out


def fake_record_chain(n_ref: int, rng, k: int, overshoot: int = 0) -> bytes:
    """ADVERSARIAL aux payload: the bytes of `k` well-formed minimal BAM records back to back (block_size, a reference id inside
    the header's table, positions >= -1, a one-byte name, no CIGAR, no bases), the last one's block_size claiming `overshoot`
    bytes more than it has -- so that a parser that starts anywhere inside finds a plausible record CHAIN that runs into
    whatever follows the payload.  When the payload is a record's last tag that is the next real record: the chain is then
    indistinguishable from the file's own but for where it starts.  (A guess of a shard's first record must survive this:
    DESIGN.md section 8 / 9 "Record boundaries".)"""
    out = b""
    for j in range(k):
        extra = int(rng.integers(0, 3))
        ref, pos = int(rng.integers(-1, n_ref)), int(rng.integers(-1, 1000))
        body = struct.pack("<iiBBHHHIiii", ref, pos, 1, int(rng.integers(0, 61)), 4680 if pos < 0 else 4681, 0, int(rng.integers(0, 4096)) | 4, 0,
                           int(rng.integers(-1, n_ref)), int(rng.integers(-1, 1000)), 0) + b"\0" + bytes(extra)
        out += struct.pack("<I", len(body) + (overshoot if j == k - 1 else 0)) + body
    return out


def adversarial_aux(rng, n_ref: int) -> bytes:
    """A B:C array (any bytes) or a Z string (no NUL: only the one-hop kind) that reads as record heads."""
    kind = rng.random()
    if kind < 0.6:     # a chain of fake records that lands exactly on the next real record (the payload is the last tag)
        return aux_array(b"ZF", b"C", fake_record_chain(n_ref, rng, int(rng.integers(1, 12))))
    if kind < 0.8:     # ... that lands 1..40 bytes INTO the next record (a chain that dies there)
        return aux_array(b"ZF", b"C", fake_record_chain(n_ref, rng, int(rng.integers(1, 6)), overshoot=int(rng.integers(1, 41))))
    # one fake head whose block_size jumps far ahead (no byte of it is zero, so it also fits a Z string): block_size and
    # l_seq of tens of MB with the other fields 0x01.. / -1 -- plausible whenever that much data follows in the chunk
    head = struct.pack("<IiiBBHHHIii", 0x02010101, -1, -1, 1, 1, 4680, 0x0101, 0x0505, 0x01010101, -1, -1)
    return aux_z(b"ZJ", head + b"AAAA") if kind < 0.9 else aux_array(b"ZJ", b"C", head + b"\1\1\1\1")


def write_bam(path: str, hb, ref_names: Sequence[str], ref_len: Sequence[int], block_payload: int = 60000,
              with_index: bool = True, sort_order: str = "coordinate", real_index: bool = False, names=None, aux=None) -> np.ndarray:
    """names / aux: per record, the read name (without its NUL) and the auxiliary bytes behind the qualities (None: "r<i>",
    nothing -- the tag-less records of rounds 1-3).
    Returns the records' BAM virtual offsets (block file offset << 16 | offset in the block's data): the ids the
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
        rec = record_bytes(hb, i, aux=aux[i] if aux is not None else b"", full_name=names[i] if names is not None else None)
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
