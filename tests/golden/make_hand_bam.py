#!/usr/bin/env python3
"""Hand-assembled BGZF/BAM fixtures, written from the text of the SAM/BAM specification (sections 4.1, 4.2, 4.2.4, 5.2)
and NOT with this repository's writers (tests/bamio.py, ngsq_synth_write_bam): every field below is packed by name,
the compression is zlib's (third party), and the shape of every gzip member is checked by reading the DEFLATE block
headers back.  What the files exercise:

  hand_spec.bam       member 0  the BAM header in a STORED deflate block
                      member 1  three records: two DYNAMIC deflate blocks with an empty stored block between them
                                (Z_FULL_FLUSH); record 1 carries aux tags of EVERY type behind its qualities
                      member 2  an EMPTY member in the middle of the file (ISIZE 0)
                      member 3  two records and the first 41 bytes of a third (records straddle members), fixed codes
                      member 4  the rest of that record and two more
                      member 5  the end-of-file marker
  hand_longcigar.bam  one record whose CIGAR is the placeholder "<l_seq>S<ref span>N" with the real CIGAR in a CG:B,I tag
                      (specification 4.2.2) -- see [N10] in oracle/oracle.h for what this build does with it

    python tests/golden/make_hand_bam.py        (re-creates the .bam/.bai files and hand_spec_expected.json)
"""
import json
import os
import struct
import zlib

HERE = os.path.dirname(os.path.abspath(__file__))
SEQ_CODES = "=ACMGRSVTWYHKDBN"
CIGAR_OPS = "MIDNSHP=X"


def reg2bin(beg, end):  # specification 5.3 (C code given there)
    end -= 1
    if beg >> 14 == end >> 14: return ((1 << 15) - 1) // 7 + (beg >> 14)
    if beg >> 17 == end >> 17: return ((1 << 12) - 1) // 7 + (beg >> 17)
    if beg >> 20 == end >> 20: return ((1 << 9) - 1) // 7 + (beg >> 20)
    if beg >> 23 == end >> 23: return ((1 << 6) - 1) // 7 + (beg >> 23)
    if beg >> 26 == end >> 26: return ((1 << 3) - 1) // 7 + (beg >> 26)
    return 0


def cigar_ops(text):
    ops, num = [], ""
    for ch in text:
        if ch.isdigit():
            num += ch
        else:
            ops.append((int(num), CIGAR_OPS.index(ch)))
            num = ""
    return ops


def aux_all_types():
    """One tag of every value type of specification 4.2.4, in the order the table lists them."""
    a = b""
    a += b"XA" + b"A" + b"Q"
    a += b"Xc" + b"c" + struct.pack("<b", -7)
    a += b"XC" + b"C" + struct.pack("<B", 250)
    a += b"Xs" + b"s" + struct.pack("<h", -30000)
    a += b"XS" + b"S" + struct.pack("<H", 60000)
    a += b"Xi" + b"i" + struct.pack("<i", -2000000000)
    a += b"XI" + b"I" + struct.pack("<I", 4000000000)
    a += b"Xf" + b"f" + struct.pack("<f", 3.5)
    a += b"XZ" + b"Z" + b"a string with spaces \t and a tab" + b"\0"
    a += b"XH" + b"H" + b"1AE301" + b"\0"
    for sub, fmt, vals in (("c", "b", [-1, 2, -3]), ("C", "B", [1, 2, 255]), ("s", "h", [-300, 300]), ("S", "H", [65535]),
                           ("i", "i", [-70000, 70000]), ("I", "I", [1, 4000000000]), ("f", "f", [0.25, -1.5, 1e10])):
        a += b"B" + sub.encode() + b"B" + sub.encode() + struct.pack("<i", len(vals)) + struct.pack("<%d%s" % (len(vals), fmt), *vals)
    a += b"BZ" + b"B" + b"C" + struct.pack("<i", 0)      # an array of no elements
    return a


def record(name, flag, ref, pos, mapq, cigar, mate_ref, mate_pos, tlen, seq, qual, aux=b""):
    ops = cigar_ops(cigar) if cigar != "*" else []
    span = sum(l for l, op in ops if CIGAR_OPS[op] in "MDN=X")
    end = pos + (span if span else 1)
    l_seq = len(seq)
    packed = bytearray((l_seq + 1) // 2)
    for i, ch in enumerate(seq):
        packed[i // 2] |= SEQ_CODES.index(ch) << (4 if i % 2 == 0 else 0)
    q = bytes(qual) if qual is not None else b"\xff" * l_seq
    assert len(q) == l_seq
    body = struct.pack("<iiBBHHHIiii", ref, pos, len(name) + 1, mapq, reg2bin(max(pos, 0), max(end, 1)) if pos >= 0 else 4680,
                       len(ops), flag, l_seq, mate_ref, mate_pos, tlen)
    body += name.encode() + b"\0" + b"".join(struct.pack("<I", l << 4 | op) for l, op in ops) + bytes(packed) + q + aux
    return struct.pack("<i", len(body)) + body


def member(deflate_payload: bytes, data: bytes) -> bytes:
    """A BGZF block (specification 4.1): gzip member with the BC extra subfield, BSIZE = total size - 1."""
    bsize = 12 + 6 + len(deflate_payload) + 8 - 1
    assert bsize < 65536 and len(data) <= 65536
    head = struct.pack("<BBBBIBBH", 31, 139, 8, 4, 0, 0, 255, 6) + b"BC" + struct.pack("<HH", 2, bsize)
    return head + deflate_payload + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data))


def deflate(parts, level=9, strategy=zlib.Z_DEFAULT_STRATEGY):
    """Raw DEFLATE of the concatenation of `parts`, with a Z_FULL_FLUSH between them (an empty stored block)."""
    co = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
    out = b""
    for k, p in enumerate(parts):
        out += co.compress(p)
        out += co.flush(zlib.Z_FINISH if k + 1 == len(parts) else zlib.Z_FULL_FLUSH)
    return out


def block_types(payload: bytes):
    """BTYPE of every DEFLATE block of a raw stream, by decoding it with zlib one block at a time (Z_BLOCK is not exposed in
    Python: the stream is cut at the known flush points instead -- only used on streams made above)."""
    types, pos_bit = [], 0
    # a tiny bit reader for the block headers we need: stored blocks can be skipped exactly, Huffman blocks are decoded by zlib
    d = zlib.decompressobj(-15)
    d.decompress(payload)
    assert d.eof
    # first block's type; later ones are asserted through their effects (flush marker 00 00 FF FF, stored header)
    return (payload[0] >> 1) & 3


text = "@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:chr1\tLN:100000\n@SQ\tSN:chr2\tLN:5000\n@CO\thand-assembled from the specification\n"
header = b"BAM\1" + struct.pack("<i", len(text)) + text.encode() + struct.pack("<i", 2)
for nm, ln in (("chr1", 100000), ("chr2", 5000)):
    header += struct.pack("<i", len(nm) + 1) + nm.encode() + b"\0" + struct.pack("<i", ln)

filler = " ".join("token%04d:%s" % (k, "ACGT"[k % 4] * (k % 7 + 1)) for k in range(160))   # makes dynamic Huffman codes pay
RECORDS = [
    dict(name="r1/all_tags", flag=99, ref=0, pos=99, mapq=60, cigar="50M", mate_ref=0, mate_pos=300, tlen=251,
         seq="ACGTN" * 10, qual=[30 + (i % 11) for i in range(50)], aux=aux_all_types() + b"ZZ" + b"Z" + filler.encode() + b"\0"),
    dict(name="r2", flag=147, ref=0, pos=300, mapq=60, cigar="20S30M", mate_ref=0, mate_pos=99, tlen=-251,
         seq="G" * 20 + "ACGT" * 7 + "AC", qual=[2] * 20 + [40] * 30, aux=b"NM" + b"C" + b"\x01" + b"YY" + b"Z" + filler[::-1].encode() + b"\0"),
    dict(name="r3", flag=0, ref=0, pos=320, mapq=0, cigar="10M2I5M3D10M100N5M1X4=", mate_ref=-1, mate_pos=-1, tlen=0,
         seq="ACGTACGTAC" + "TT" + "GGGGG" + "ACGTACGTAC" + "CCCCC" + "A" + "TTTT", qual=None,
         aux=b"XX" + b"Z" + (filler * 2).encode() + b"\0"),
    dict(name="r4", flag=1024 + 16, ref=0, pos=5000, mapq=255, cigar="5H40M5H", mate_ref=-1, mate_pos=-1, tlen=0,
         seq="ACGT" * 10, qual=[93] * 40),
    dict(name="r5", flag=256, ref=0, pos=5010, mapq=3, cigar="40M", mate_ref=-1, mate_pos=-1, tlen=0,
         seq="=ACMGRSVTWYHKDBN" * 2 + "ACGTACGT", qual=[0] * 40),
    dict(name="r6/straddles", flag=65, ref=0, pos=99990, mapq=20, cigar="10M", mate_ref=1, mate_pos=10, tlen=0,
         seq="AAAAACCCCC", qual=[10, 11, 12, 13, 14, 15, 16, 17, 18, 19], aux=b"RG" + b"Z" + b"group" + b"\0"),
    dict(name="r7", flag=129, ref=1, pos=10, mapq=20, cigar="120M", mate_ref=0, mate_pos=99990, tlen=0,
         seq="ACGT" * 30, qual=[20] * 120),
    dict(name="r8/unplaced", flag=77, ref=-1, pos=-1, mapq=0, cigar="*", mate_ref=-1, mate_pos=-1, tlen=0,
         seq="NNNNNNNNNN", qual=[5] * 10),
]
raw = [record(**r) for r in RECORDS]

m0 = member(deflate([header], level=0), header)                    # level 0: stored blocks only
part_a, part_b = raw[0] + raw[1][:100], raw[1][100:] + raw[2]
p1 = deflate([part_a, part_b])
assert (p1[0] >> 1) & 3 == 2, "the first block of member 1 should use dynamic Huffman codes"
cut = p1.index(b"\x00\x00\xff\xff")                                 # the empty stored block of Z_FULL_FLUSH
assert (p1[cut + 4] >> 1) & 3 == 2, "the block behind the flush marker should use dynamic Huffman codes"
m1 = member(p1, part_a + part_b)
m2 = member(deflate([b""]), b"")                                     # an empty member
d3 = raw[3] + raw[4] + raw[5][:41]
p3 = deflate([d3], strategy=zlib.Z_FIXED)
assert (p3[0] >> 1) & 3 == 1, "member 3 should use the fixed codes"
m3 = member(p3, d3)
d4 = raw[5][41:] + raw[6] + raw[7]
m4 = member(deflate([d4], level=6), d4)
eof = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")   # specification 4.1.2, verbatim
blob = m0 + m1 + m2 + m3 + m4 + eof
assert zlib.decompress(m1[18:-8], -15) == part_a + part_b
with open(os.path.join(HERE, "hand_spec.bam"), "wb") as f:
    f.write(blob)
# BAI (specification 5.2): magic, n_ref, and for each reference no bins and no intervals; n_no_coor
with open(os.path.join(HERE, "hand_spec.bam.bai"), "wb") as f:
    f.write(b"BAI\1" + struct.pack("<i", 2) + struct.pack("<ii", 0, 0) * 2 + struct.pack("<Q", 1))

# the records' virtual offsets (member file offset << 16 | offset in the member's data), by construction
starts, off = [], 0
for m in (m0, m1, m2, m3, m4):
    starts.append(off)
    off += len(m)
voff = []
voff.append(starts[1] << 16 | 0)
voff.append(starts[1] << 16 | len(raw[0]))
voff.append(starts[1] << 16 | len(raw[0]) + len(raw[1]))
voff.append(starts[3] << 16 | 0)
voff.append(starts[3] << 16 | len(raw[3]))
voff.append(starts[3] << 16 | len(raw[3]) + len(raw[4]))
voff.append(starts[4] << 16 | len(raw[5]) - 41)
voff.append(starts[4] << 16 | len(raw[5]) - 41 + len(raw[6]))

# ---- what a reader must hand to the facets, and a few numbers of the facets worked out from the table above
expected = {"references": [["chr1", 100000], ["chr2", 5000]], "records": [], "virtual_offsets": voff,
            "file_bytes": len(blob), "members": [len(m) for m in (m0, m1, m2, m3, m4, eof)]}
for r in RECORDS:
    expected["records"].append({"flag": r["flag"], "mapq": r["mapq"], "ref_id": r["ref"], "pos": r["pos"], "mate_ref_id": r["mate_ref"],
                                "tlen": r["tlen"], "l_seq": len(r["seq"]), "cigar": [l << 4 | op for l, op in (cigar_ops(r["cigar"]) if r["cigar"] != "*" else [])],
                                "seq_codes": [SEQ_CODES.index(c) for c in r["seq"]],
                                "qual": list(r["qual"]) if r["qual"] is not None else None})
expected["general"] = {
    # general.rs:31-124, by hand: r1 r2 paired proper primaries; r3 single; r4 duplicate+reverse; r5 secondary; r6 r7 paired
    # with the mate on another sequence (r6 read 1, mapq 20 >= 5: high quality; r7 read 2); r8 paired, unmapped, mate unmapped
    "total": 8, "unmapped": 1, "duplicate": 1, "primary": 7, "secondary": 1, "supplementary": 0,
    "primary_mapped": 6, "primary_duplicate": 1, "paired": 5, "read_1": 3, "read_2": 2, "proper_pair": 2,
    "singleton": 0, "mate_mapped": 4, "mate_reference_sequence_id_mismatch": 2, "mate_reference_sequence_id_mismatch_hq": 2,
    # CIGAR operations: read one = flag 0x40 (r1, r6, r8 -- r8 has none), everything else counts as read two
    "read_one_cigar_ops": {"M": 2}, "read_two_cigar_ops": {"S": 1, "M": 1 + 4 + 1 + 1 + 1, "I": 1, "D": 1, "N": 1, "X": 1, "=": 1, "H": 2},
}
expected["template_length"] = {"251": 1, "0": 6, "ignored": 1}          # r2's -251 is out of range
# quality_scores.rs:37-49: cycle -> records with a score there (r3 has none: 0xFF-filled)
expected["quality_rows"] = {"1": 7, "10": 7, "11": 5, "40": 5, "41": 3, "50": 3, "51": 1, "120": 1, "121": 0}
# coverage.rs:148-180: depth of a few positions of chr1 (1-based).  r1 covers 100..149; r2 301..330 (POS is the first ALIGNED
# base: the 20 soft-clipped bases lie in front of it and cover nothing); r3 321..458 (D and N are covered, I is not);
# r4 5001..5040 (hard clips cover nothing; duplicates count); r5 5011..5050 (secondaries count); r6 99991..100000
expected["depth_chr1"] = {"99": 0, "100": 1, "149": 1, "150": 0, "300": 0, "301": 1, "320": 1, "321": 2, "330": 2, "331": 1, "458": 1, "459": 0,
                          "5001": 1, "5011": 2, "5040": 2, "5041": 1, "5050": 1, "5051": 0, "99991": 1, "100000": 1}
with open(os.path.join(HERE, "hand_spec_expected.json"), "w") as f:
    json.dump(expected, f, indent=1)

# ---- the long-CIGAR convention (specification 4.2.2): 3 real operations stand in for "more than 65535"
real = cigar_ops("10M5D20M")
cg = b"CG" + b"B" + b"I" + struct.pack("<i", len(real)) + b"".join(struct.pack("<I", l << 4 | op) for l, op in real)
long_rec = record("long", 0, 0, 1000, 30, "30S35N", -1, -1, 0, "ACGTAC" * 5, [25] * 30, aux=b"NM" + b"C" + b"\0" + cg)
with open(os.path.join(HERE, "hand_longcigar.bam"), "wb") as f:
    f.write(member(deflate([header], level=0), header) + member(deflate([long_rec]), long_rec) + eof)
with open(os.path.join(HERE, "hand_longcigar.bam.bai"), "wb") as f:
    f.write(b"BAI\1" + struct.pack("<i", 2) + struct.pack("<ii", 0, 0) * 2 + struct.pack("<Q", 0))
print("hand_spec.bam", len(blob), "bytes;", [hex(v) for v in voff])
