"""The reference FASTA of the Edits facet (include/ngsq_reference.h) against the reference's handling of it:
EditsFacet::setup (src/qc/sequence_based/edits.rs:177-215: the record whose name matches, first one wins, "sequence {} not
found in reference FASTA."), the slice + Base::try_from of edits.rs:257-261 ([N9] in oracle/oracle.h: case is folded; a byte
that is no base letter fails the reads over it and no others; a FASTA sequence shorter than @SQ LN fails the reads that
run past its end).

CPU: the definition-line index (scan and .fai), the byte -> code table against the oracle's restatement, the hand golden
through the oracle.  GPU (-m gpu): file -> pinned -> HIP conversion kernels -> Edits kernels against the oracle fed the same
file parsed in Python."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from ngs_amd import ffi, host
from tests.util import batch_from_records, compare_contexts, json_equal, make_edit_friendly, random_batch, take_records

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hand_softmasked.json")
LETTERS = "=ACMGRSVTWYHKDBN"


def parse_fasta(data: bytes):
    """[(name, sequence bytes)] as noodles-fasta reads them: a record starts at a line that begins with '>', its name ends at
    the first blank, its sequence is the following lines without their terminators ("\\n" or "\\r\\n") -- written independently
    of the library's index and kernels."""
    out, name, seq = [], None, []
    for line in data.split(b"\n"):
        if line.startswith(b">"):
            if name is not None:
                out.append((name, b"".join(seq)))
            name = line[1:].split(b" ")[0].split(b"\t")[0].rstrip(b"\r").decode()
            seq = []
        elif name is not None:
            seq.append(line[:-1] if line.endswith(b"\r") else line)
    if name is not None:
        out.append((name, b"".join(seq)))
    return out


def fasta_index(lib, path, threads=0):
    h = C.c_void_p()
    assert lib.ngsq_fasta_open(path.encode(), threads, C.byref(h)) == 0, lib.ngsq_fasta_last_error()
    n = lib.ngsq_fasta_n_records(h)
    recs = [(lib.ngsq_fasta_record_name(h, i).decode(), int(lib.ngsq_fasta_record_text_bytes(h, i))) for i in range(max(n, 0))]
    from_fai = lib.ngsq_fasta_index_from_fai(h)
    lib.ngsq_fasta_close(h)
    return n, recs, from_fai


def test_base_code_table_is_the_oracles(lib, oracle_mod):
    """the product's byte -> code conversion (host function and device kernel share it) against orc_fasta_base_code"""
    o = oracle_mod.load()
    for b in range(256):
        assert lib.ngsq_fasta_base_code(b) == o.orc_fasta_base_code(b), b
    for k, ch in enumerate(LETTERS):
        assert lib.ngsq_fasta_base_code(ord(ch)) == k
        assert lib.ngsq_fasta_base_code(ord(ch.lower())) == k          # [N9]: case is folded
    for ch in "*-.UXuxj\n\r >0":
        assert lib.ngsq_fasta_base_code(ord(ch)) == -1


def test_definition_line_index(lib, tmp_path):
    rng = np.random.default_rng(3)
    recs = [("chr1", 5000, 60), ("chrUn_x", 0, 60), ("with", 777, 50), ("chr1", 90, 60), ("last", 1234, 70)]
    text = b"\n"                                                         # a blank line in front is tolerated
    expect = []
    for name, L, w in recs:
        s = bytes(rng.choice(np.frombuffer(b"ACGTacgtN", np.uint8), L))
        body = b"".join(s[k:k + w] + b"\n" for k in range(0, L, w))
        text += f">{name} description > with a bracket\n".encode() + body
        expect.append((name, len(body)))
    text = text[:-1]                                                     # no newline at the end of the file
    expect[-1] = (expect[-1][0], expect[-1][1] - 1)
    p = str(tmp_path / "a.fa")
    open(p, "wb").write(text)
    for threads in (1, 3, 7):
        n, got, from_fai = fasta_index(lib, p, threads)
        assert n == len(recs) and got == expect and not from_fai
    assert [x[0] for x in parse_fasta(text)] == [x[0] for x in expect]
    # a .fai that agrees is used; one that does not (an offset off by one, a record appended later) is ignored
    lines, off = [], 1
    for name, L, w in recs:
        off += len(f">{name} description > with a bracket\n")
        lines.append(f"{name}\t{L}\t{off}\t{w}\t{w + 1}\n")
        off += L + (L + w - 1) // w
    open(p + ".fai", "w").write("".join(lines))
    n, got, from_fai = fasta_index(lib, p)
    assert from_fai and n == len(recs) and [g[0] for g in got] == [e[0] for e in expect]
    open(p + ".fai", "w").write("".join(lines[:-1]))                     # the index misses the last record
    n, got, from_fai = fasta_index(lib, p)
    assert not from_fai and got == expect
    bad = lines[:]
    bad[2] = bad[2].replace(f"\t{777}\t", f"\t{778}\t")
    f2 = bad[3].split("\t")
    f2[2] = str(int(f2[2]) + 1)
    bad[3] = "\t".join(f2)
    open(p + ".fai", "w").write("".join(bad))
    n, got, from_fai = fasta_index(lib, p)
    assert not from_fai and got == expect
    # sequence data before the first definition line; an empty file; a file that is not there
    open(p, "wb").write(b"ACGT\n>x\nAC\n")
    os.remove(p + ".fai")
    h = C.c_void_p()
    assert lib.ngsq_fasta_open(p.encode(), 0, C.byref(h)) == 0
    assert lib.ngsq_fasta_n_records(h) == ffi.ERR_INVALID_ARGUMENT and b"before the first definition line" in lib.ngsq_fasta_last_error()
    lib.ngsq_fasta_close(h)
    open(p, "wb").write(b"")
    assert fasta_index(lib, p)[0] == 0
    assert lib.ngsq_fasta_open(str(tmp_path / "nope.fa").encode(), 0, C.byref(h)) == ffi.ERR_INVALID_ARGUMENT
    assert b"No such file" in lib.ngsq_fasta_last_error()


def golden_expectations(g, r1, r2, vaf, refs=None, alts=None):
    e = g["expected"]

    def dense(d, n):
        a = np.zeros(n, dtype=np.uint64)
        for k, v in d.items():
            a[int(k)] = v
        return a
    np.testing.assert_array_equal(r1, dense(e["read_one_edits"], ffi.EDITS_BINS))
    np.testing.assert_array_equal(r2, dense(e["read_two_edits"], ffi.EDITS_BINS))
    np.testing.assert_array_equal(vaf, dense(e["vaf_histogram"], ffi.VAF_BINS))
    if refs is not None:
        np.testing.assert_array_equal(refs, dense(e["refs_per_position"], 21))
        np.testing.assert_array_equal(alts, dense(e["alts_per_position"], 21))


def test_hand_golden_softmasked_through_the_oracle(oracle_mod):
    """tests/golden/hand_softmasked.json: lower-case reference bases under reads, worked by hand"""
    g = json.load(open(GOLD))
    (name, seq), = parse_fasta(g["fasta"].encode())
    assert name == "chrA" and len(seq) == 20
    codes = oracle_mod.fasta_codes(seq)
    assert codes.max() <= 15
    o = oracle_mod.Oracle(g["ref_len"], facets=ffi.FACET_EDITS, ref_bases=[codes])
    o.process_batch(batch_from_records(g["records"]))
    assert o.finalize() == 0
    golden_expectations(g, *o.edits())
    doc = o.results(g["ref_names"])["edits"]
    assert doc["summary"] == {"mean_edits_read_one": g["expected"]["mean_edits_read_one"], "mean_edits_read_two": g["expected"]["mean_edits_read_two"]}


# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_hand_golden_softmasked_from_the_file(gpu_lib, tmp_path):
    g = json.load(open(GOLD))
    p = str(tmp_path / "g.fa")
    open(p, "w").write(g["fasta"])
    with host.QcContext(g["ref_len"], facets=ffi.FACET_EDITS, ref_fasta=p, ref_names=g["ref_names"], lib=gpu_lib) as gpu:
        gpu.process_batch(batch_from_records(g["records"]))
        assert gpu.finalize() == 0
        golden_expectations(g, *gpu.edits(), *gpu.edits_positions(0))
        st = gpu.reference_wait()
        assert st["sequences"] == 1 and st["bases"] == 20 and st["invalid_bytes"] == 0 and st["text_bytes"] == 23


def awkward_fasta(rng, names, lens, fasta_len=None, crlf=False, poison=None):
    """FASTA text for `names` with everything a reader has to cope with: its own sequence order, records the BAM does not have,
    a second record of a name that is already there (the first wins), line widths per record, soft-masked stretches, blank
    lines, no newline at the end.  fasta_len[r]: the FASTA's own length of sequence r.  poison: {r: [(0-based position, byte)]}.
    Returns (text, sequences by name as the FASTA holds them: first record of each name)."""
    eol = b"\r\n" if crlf else b"\n"
    order = list(rng.permutation(len(names)))
    out, seqs = [], {}
    out.append(b">decoy_1 not in the bam" + eol + b"ACGTNNNNNN" * 30 + eol)
    for r in order:
        L = int(fasta_len[r]) if fasta_len is not None else int(lens[r])
        s = rng.choice(np.frombuffer(b"ACGTN", np.uint8), L, p=[.25, .25, .25, .24, .01]).copy()
        mask = np.zeros(L, dtype=bool)
        p = 0
        while p < L:
            run = int(rng.integers(5, 400))
            if rng.random() < 0.5:
                mask[p:p + run] = True
            p += run
        s = np.where(mask, s | 0x20, s).astype(np.uint8)
        for pos, byte in (poison or {}).get(r, []):
            s[pos] = byte
        s = s.tobytes()
        seqs[names[r]] = s
        w = int(rng.choice([1, 7, 60, 61, 70, 4096, 100_000]))
        out.append(f">{names[r]}\tdescribed, tab in front".encode() + eol)
        for k in range(0, L, w):
            out.append(s[k:k + w] + eol)
            if rng.random() < 0.01:
                out.append(eol)                                           # a blank line inside a sequence
        if r == order[len(order) // 2]:
            out.append(f">{names[r]} a second record of the same name".encode() + eol + b"TTTTTTTTTT" * 10 + eol)
    text = b"".join(out)
    return text[:-len(eol)], seqs


@pytest.mark.gpu
@pytest.mark.parametrize("crlf", [False, True])
def test_file_reference_matches_the_oracle(gpu_lib, oracle_mod, tmp_path, crlf):
    rng = np.random.default_rng(41 + crlf)
    lens = [30_000, 8_191, 4_096, 4_097, 1, 12_289, 700]
    names = [f"s{i}" for i in range(len(lens))]
    text, seqs = awkward_fasta(rng, names, lens, crlf=crlf)
    p = str(tmp_path / "r.fa")
    open(p, "wb").write(text)
    parsed = dict(reversed(parse_fasta(text)))                            # first record of a name wins
    assert all(parsed[n] == seqs[n] for n in names)
    bases = [oracle_mod.fasta_codes(parsed[n]) for n in names]
    hb = make_edit_friendly(random_batch(rng, 20_000, lens, weird=False, min_len=1, max_len=250), rng, bases, lens)
    kw = dict(facets=ffi.FACET_EDITS | ffi.FACET_GC_CONTENT, gc_seed=9)
    orc = oracle_mod.Oracle(lens, ref_bases=bases, **kw)
    orc.process_batch(hb)
    rc = orc.finalize(allow_malformed=True)          # (reads that do not fit the 1-base and 700-base sequences: the same aborts on both sides)
    for threads in (1, 4):
        with host.QcContext(lens, ref_fasta=p, ref_names=names, fasta_threads=threads, lib=gpu_lib, **kw) as gpu:
            gpu.process_batch(hb.slice(0, 7000))
            gpu.process_batch(gpu.upload(hb.slice(7000, hb.n)))
            assert gpu.finalize(allow_malformed=True) == rc
            compare_contexts(gpu, orc, len(lens), kw["facets"], 50_000, lens)
            for r in range(len(lens)):
                for a, b in zip(gpu.edits_positions(r), orc_positions(orc, hb, lens, bases, r)):
                    np.testing.assert_array_equal(a, b)
            st = gpu.reference_wait()
            assert st["sequences"] == len(lens) and st["bases"] == sum(lens) and st["invalid_bytes"] == 0 and st["shorter"] == st["longer"] == 0


def orc_positions(orc, hb, lens, bases, r):
    from tests.test_parity_gpu import brute_force_refs_alts
    if not hasattr(orc, "_pos_cache"):
        orc._pos_cache = brute_force_refs_alts(hb, lens, bases)
    refs, alts = orc._pos_cache
    return refs[r], alts[r]


@pytest.mark.gpu
def test_fasta_lengths_and_bytes_that_are_no_bases(gpu_lib, oracle_mod, tmp_path):
    """A FASTA sequence shorter than @SQ LN fails the reads that run past ITS end, a longer one changes nothing for reads inside
    LN (beyond it: test_reads_beyond_ln_inside_a_longer_fasta_sequence); a byte
    Base::try_from refuses fails the reads whose slice start..start+span holds it -- under a deletion or a skip too -- and no
    other (edits.rs:257-261).  The device keeps such positions in a list; the oracle sees them as codes above 15."""
    rng = np.random.default_rng(77)
    lens = [20_000, 9_000, 5_000, 3_000]
    names = ["a", "b", "c", "d"]
    fasta_len = [20_000, 7_500, 6_000, 3_000]                             # b: shorter in the FASTA; c: longer
    poison = {0: [(100, ord("*")), (5_000, ord("-")), (5_001, ord("u")), (19_999, ord("."))], 3: [(1_500, ord("j"))]}
    text, seqs = awkward_fasta(rng, names, lens, fasta_len=fasta_len, poison=poison)
    p = str(tmp_path / "r.fa")
    open(p, "wb").write(text)
    parsed = dict(reversed(parse_fasta(text)))
    assert [len(parsed[n]) for n in names] == fasta_len
    bases = [oracle_mod.fasta_codes(parsed[n]) for n in names]           # (the FASTA's own lengths: ref_bases_len below)
    assert int((bases[0] > 15).sum()) == 4 and int((bases[3] > 15).sum()) == 1
    clean = [np.where(b > 15, 15, b).astype(np.uint8) for b in bases]
    padded = [np.concatenate([b, np.full(max(0, L - len(b)), 15, np.uint8)]) for b, L in zip(clean, lens)]
    hb = make_edit_friendly(random_batch(rng, 30_000, lens, weird=False, min_len=1, max_len=200), rng, padded, lens)
    kw = dict(facets=ffi.FACET_EDITS)
    orc = oracle_mod.Oracle(lens, ref_bases=bases, ref_bases_len=fasta_len, **kw)
    orc.process_batch(hb)
    rc = orc.finalize(allow_malformed=True)
    errs = orc.error_counts()
    assert rc == ffi.ERR_MALFORMED_RECORD and errs["edits_bad_reference"] > 100     # reads past b's end, reads over the five bytes
    with host.QcContext(lens, ref_fasta=p, ref_names=names, lib=gpu_lib, **kw) as gpu:
        gpu.process_batch(hb)
        assert gpu.finalize(allow_malformed=True) == rc
        compare_contexts(gpu, orc, 4, ffi.FACET_EDITS, 50_000, lens)
        st = gpu.reference_wait()
        assert (st["invalid_bytes"], st["shorter"], st["longer"]) == (5, 1, 1)
    # the batch API with ngsq_config.ref_bases_len: the same lengths rule (the codes must be 4-bit there)
    orc2 = oracle_mod.Oracle(lens, ref_bases=clean, ref_bases_len=[len(b) for b in clean], **kw)
    orc2.process_batch(hb)
    rc2 = orc2.finalize(allow_malformed=True)
    with host.QcContext(lens, ref_bases=clean, ref_bases_len=[len(b) for b in clean], lib=gpu_lib, **kw) as gpu:
        gpu.process_batch(hb)
        assert gpu.finalize(allow_malformed=True) == rc2
        compare_contexts(gpu, orc2, 4, ffi.FACET_EDITS, 50_000, lens)
    assert 0 < orc2.error_counts()["edits_bad_reference"] < errs["edits_bad_reference"]


@pytest.mark.gpu
def test_reads_beyond_ln_inside_a_longer_fasta_sequence(gpu_lib, oracle_mod, tmp_path):
    """A FASTA sequence LONGER than @SQ LN: edits.rs:257-261 slices the FASTA's own sequence, so a read may end beyond LN inside
    it; what stops the run then is refs/alts_per_position.increment().unwrap() (:283-291, LN + 1 bins) -- for an `M` base only.
    Bases beyond LN under D, N, = or X go through; an M there, a slice beyond the FASTA's end, a refused byte under the tail do
    not.  (Round 6: until tests/literal_model.py read the source a second time every read ending beyond LN was counted as an
    abort.)  Three judges: the oracle, the literal model, the numbers worked out here."""
    from tests import literal_model as lm
    rng = np.random.default_rng(5)
    lens, names, fasta_len = [5_000, 900], ["c", "d"], [6_000, 900]
    text, seqs = awkward_fasta(rng, names, lens, fasta_len=fasta_len, poison={0: [(5_499, ord("!"))]})   # position 5500 of c
    p = str(tmp_path / "r.fa")
    open(p, "wb").write(text)
    parsed = dict(reversed(parse_fasta(text)))
    codes = oracle_mod.fasta_codes(parsed["c"])
    letters = "=ACMGRSVTWYHKDBN"

    def read_at(pos0, n):   # the reference's own bases (N where the FASTA has the refused byte)
        return "".join(letters[c] if c <= 15 else "N" for c in codes[pos0:pos0 + n])
    recs = [
        dict(flag=0x40, ref_id=0, pos=4899, cigar="101M50D", seq=read_at(4899, 101), qual=[30] * 101),            # M 4900..5000, D 5001..5050: goes through
        dict(flag=0x40, ref_id=0, pos=4899, cigar="102M", seq=read_at(4899, 102), qual=[30] * 102),               # an M base at 5001: stops
        dict(flag=0x40, ref_id=0, pos=4899, cigar="101M2000N", seq=read_at(4899, 101), qual=[30] * 101),          # the slice ends at 7000 > 6000: stops
        dict(flag=0x80, ref_id=0, pos=4949, cigar="51M10N5X", seq=read_at(4949, 51) + "ACGTA", qual=[30] * 56),   # N and X beyond LN: goes through
        dict(flag=0x80, ref_id=0, pos=4949, cigar="40M20D5M", seq=read_at(4949, 40) + "ACGTA", qual=[30] * 45),   # the second M at 5010..: stops
        dict(flag=0x80, ref_id=0, pos=4949, cigar="51M600N5=", seq=read_at(4949, 51) + "ACGTA", qual=[30] * 56),  # the slice holds position 5500: stops
        dict(flag=0x40, ref_id=0, pos=4989, cigar="11M989D", seq="T" * 11, qual=[30] * 11),                       # ends exactly at the FASTA's 6000... over 5500: stops
        dict(flag=0x40, ref_id=0, pos=5099, cigar="10M", seq="A" * 10, qual=[30] * 10),                           # starts beyond LN: not yielded by query()
        dict(flag=0x40, ref_id=1, pos=0, cigar="900M", seq=seqs["d"].decode().upper(), qual=[30] * 900),          # d, whole: no edit
    ]
    hb = batch_from_records(recs)
    bases = [oracle_mod.fasta_codes(parsed[n]) for n in names]
    kw = dict(facets=ffi.FACET_EDITS)
    orc = oracle_mod.Oracle(lens, ref_bases=bases, ref_bases_len=fasta_len, **kw)
    orc.process_batch(hb)
    assert orc.finalize(allow_malformed=True) == ffi.ERR_MALFORMED_RECORD
    errs = orc.error_counts()
    assert errs["edits_bad_reference"] == 5 and sum(errs.values()) == 5, errs
    r1, r2, _vaf = orc.edits()
    assert int(r1.sum()) == 2 and int(r2.sum()) == 1 and int(r1[0]) == 2      # the 101M50D read and d; 51M10N5X
    # the literal model: the same five records stop it, and without them it writes the oracle's document
    model = lm.records_of(hb)
    stopped = []
    for i, rec in enumerate(model):
        e = lm.Edits(dict(zip(names, [parsed[n] for n in names])))
        e.setup(names[rec.ref_id], lens[rec.ref_id])
        try:
            if any(True for _ in lm.query([rec], rec.ref_id, lens[rec.ref_id])):
                e.process(names[rec.ref_id], lens[rec.ref_id], rec)
        except lm.Abort:
            stopped.append(i)
    assert stopped == [1, 2, 4, 5, 6]
    with host.QcContext(lens, ref_fasta=p, ref_names=names, lib=gpu_lib, **kw) as gpu:
        gpu.process_batch(hb)
        assert gpu.finalize(allow_malformed=True) == ffi.ERR_MALFORMED_RECORD
        compare_contexts(gpu, orc, 2, ffi.FACET_EDITS, 50_000, lens)
        st = gpu.reference_wait()
        assert (st["invalid_bytes"], st["shorter"], st["longer"]) == (1, 0, 1)
    # the batch API (ngsq_config.ref_bases_len: 4-bit codes, so without the refused byte -- the 600N read then goes through)
    clean = [np.where(b > 15, 15, b).astype(np.uint8) for b in bases]
    orc2 = oracle_mod.Oracle(lens, ref_bases=clean, ref_bases_len=fasta_len, **kw)
    orc2.process_batch(hb)
    assert orc2.finalize(allow_malformed=True) == ffi.ERR_MALFORMED_RECORD and orc2.error_counts()["edits_bad_reference"] == 3
    for layout in ("offsets", "uploaded"):
        with host.QcContext(lens, ref_bases=clean, ref_bases_len=fasta_len, lib=gpu_lib, **kw) as gpu:
            gpu.process_batch(hb if layout == "offsets" else gpu.upload(hb))
            assert gpu.finalize(allow_malformed=True) == ffi.ERR_MALFORMED_RECORD
            compare_contexts(gpu, orc2, 2, ffi.FACET_EDITS, 50_000, lens)
    keep = take_records(hb, np.array([0, 3, 7, 8]))
    orc3 = oracle_mod.Oracle(lens, ref_bases=bases, ref_bases_len=fasta_len, **kw)
    orc3.process_batch(keep)
    assert orc3.finalize() == 0
    want = lm.run(lm.records_of(keep), names, lens, [1, 1], general=False, template_length=False, gc_content=False, quality_scores=False,
                  coverage=False, fasta=dict(zip(names, [parsed[n] for n in names])))
    json_equal(orc3.results(names)["edits"], want["edits"])


@pytest.mark.gpu
def test_carriage_returns(gpu_lib, oracle_mod, tmp_path):
    """noodles-fasta strips "\\n" and "\\r\\n"; a '\\r' anywhere else stays a byte of the sequence -- one Base::try_from refuses"""
    text = b">q\nAC\rGT\nAA\r\n>r\r\nAC\r\nGT\r"                        # q = AC\rGTAA (7 bytes), r = ACGT (the last \r ends the file's last line)
    assert parse_fasta(text) == [("q", b"AC\rGTAA"), ("r", b"ACGT")]
    p = str(tmp_path / "c.fa")
    open(p, "wb").write(text)
    recs = [dict(flag=0, ref_id=0, pos=0, cigar="2M", seq="AC", qual=[9, 9]),        # positions 1-2: fine
            dict(flag=0, ref_id=0, pos=1, cigar="2M", seq="CA", qual=[9, 9]),        # 2-3: holds the \r
            dict(flag=0, ref_id=0, pos=3, cigar="4M", seq="GTAA", qual=[9] * 4),     # 4-7: fine
            dict(flag=0, ref_id=1, pos=0, cigar="4M", seq="ACGA", qual=[9] * 4)]     # r: one edit
    hb = batch_from_records(recs)
    bases = [oracle_mod.fasta_codes(b"AC\rGTAA"), oracle_mod.fasta_codes(b"ACGT")]
    orc = oracle_mod.Oracle([7, 4], facets=ffi.FACET_EDITS, ref_bases=bases)
    orc.process_batch(hb)
    assert orc.finalize(allow_malformed=True) == ffi.ERR_MALFORMED_RECORD and orc.error_counts()["edits_bad_reference"] == 1
    with host.QcContext([7, 4], facets=ffi.FACET_EDITS, ref_fasta=p, ref_names=["q", "r"], lib=gpu_lib) as gpu:
        gpu.process_batch(hb)
        assert gpu.finalize(allow_malformed=True) == ffi.ERR_MALFORMED_RECORD
        compare_contexts(gpu, orc, 2, ffi.FACET_EDITS, 50_000, [7, 4])
        assert gpu.edits()[1][1] == 1 and gpu.reference_wait()["invalid_bytes"] == 1


@pytest.mark.gpu
def test_missing_and_unwanted_sequences(gpu_lib, oracle_mod, tmp_path):
    rng = np.random.default_rng(5)
    lens = [6_000, 5_000, 4_000]
    names = ["x", "y", "z"]
    text, seqs = awkward_fasta(rng, names[:2], lens[:2])
    p = str(tmp_path / "r.fa")
    open(p, "wb").write(text)
    # EditsFacet::setup bails on the sequence the FASTA does not have (edits.rs:207-209)
    with host.QcContext(lens, facets=ffi.FACET_EDITS, ref_fasta=p, ref_names=names, lib=gpu_lib) as gpu:
        with pytest.raises(host.NgsqError) as e:
            gpu.reference_wait()
        assert "sequence z not found in reference FASTA." in str(e.value)
        hb = random_batch(rng, 100, lens, weird=False)
        with pytest.raises(host.NgsqError) as e:
            gpu.process_batch(hb)
        assert "sequence z not found in reference FASTA." in str(e.value)
    # a worker of a sharded scan loads what its byte range can reach: records on a sequence it left out are errors, never
    # compared with nothing
    bases = [oracle_mod.fasta_codes(seqs[n]) for n in names[:2]]
    hb = make_edit_friendly(random_batch(rng, 5000, lens[:2], weird=False, min_len=1), rng, bases, lens[:2])
    orc = oracle_mod.Oracle(lens[:2], facets=ffi.FACET_EDITS, ref_bases=[bases[0], None])
    orc.process_batch(hb)
    rc = orc.finalize(allow_malformed=True)
    with host.QcContext(lens[:2], facets=ffi.FACET_EDITS, ref_fasta=p, ref_names=names[:2], ref_wanted=[1, 0], lib=gpu_lib) as gpu:
        gpu.process_batch(hb)
        assert gpu.finalize(allow_malformed=True) == rc == ffi.ERR_MALFORMED_RECORD
        compare_contexts(gpu, orc, 2, ffi.FACET_EDITS, 50_000, lens[:2])
        assert gpu.reference_wait()["sequences"] == 1
    # lifecycle: a deferred context without a load; a load on a context that was not created for one
    cfg_ctx = host.QcContext(lens, facets=ffi.FACET_EDITS, lib=gpu_lib)
    h = C.c_void_p()
    assert gpu_lib.ngsq_fasta_open(p.encode(), 0, C.byref(h)) == 0
    arr = (C.c_char_p * 3)(*[n.encode() for n in names])
    assert gpu_lib.ngsq_reference_load(cfg_ctx._ctx, h, arr, None) == ffi.ERR_STATE
    gpu_lib.ngsq_fasta_close(h)
    cfg_ctx.close()
