"""The `ngs qc` command line (ngs_amd/csrc/cli/ngs_main.cpp) against the reference's CLI contract
(src/qc/command.rs:36-218, :226-421): flags, error texts, output file, and -- on a GPU -- the
whole path BAM file -> ingest -> kernels -> <prefix>.results.json compared with the oracle."""
import json
import os
import subprocess

import numpy as np
import pytest

from ngs_amd import build, ffi, host
from tests import bamio
from tests.util import json_equal, random_batch

GENOME = "GRCh38_no_alt_AnalysisSet"
NAMES = ["chr1", "chr2", "chrM", "chrUn_KI270302v1"]      # chrM is not in the primary assembly
LENS = [60_000, 20_000, 16_569, 2_274]
PRIMARY = [1, 1, 0, 1]


@pytest.fixture(scope="module")
def ngs(lib):
    return build.build_cli(verbose=False)


def run(ngs, *args, cwd=None, env=None):
    return subprocess.run([ngs, *args], capture_output=True, text=True, cwd=cwd, env=env)


def sorted_batch(seed, n, max_len=150, min_len=150):
    rng = np.random.default_rng(seed)
    hb = random_batch(rng, n, LENS, max_len=max_len, min_len=min_len, weird=False)
    # placed, coordinate-sorted, well-formed (no records the reference would abort on)
    c = hb.cols
    c["flag"] &= np.uint16(0xFFFF ^ 0x1)          # unpaired: no mate reference needed
    order = np.lexsort((c["pos"], c["ref_id"]))
    recs = [hb.slice(int(i), int(i) + 1) for i in order]
    cols = {}
    for k in host.FIXED_COLUMNS:
        cols[k] = np.concatenate([r.cols[k] for r in recs])
    for data, off in (("seq", "seq_off"), ("qual", "qual_off"), ("cigar", "cigar_off")):
        cols[data] = np.concatenate([r.cols[data] for r in recs])
        lens = [len(r.cols[data]) for r in recs]
        cols[off] = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    return host.HostBatch(hb.n, cols, 0, 0, 0, 0)


def test_error_texts(ngs, tmp_path):
    hb = sorted_batch(1, 50)
    bam = str(tmp_path / "a.bam")
    bamio.write_bam(bam, hb, NAMES, LENS, with_index=False)
    r = run(ngs, "qc", bam, "hg19")
    assert r.returncode == 1 and "reference genome is not supported: hg19." in r.stderr
    r = run(ngs, "qc", bam, GENOME)
    assert r.returncode == 1 and "reading BAM index" in r.stderr           # <bam>.bai is required
    bamio.write_bam(bam, hb, NAMES, LENS)
    r = run(ngs, "qc", bam, GENOME.lower(), "--only", "Nope")                # genome names match case-insensitively
    assert r.returncode == 1 and "No facets matched the specified `--only` flag: Nope" in r.stderr
    bad = str(tmp_path / "b.bam")
    bamio.write_bam(bad, hb, ["chr1", "chr2", "chrM", "contig_7"], LENS)
    r = run(ngs, "qc", bad, GENOME)
    assert r.returncode == 1 and 'Sequence "contig_7" not found in specified reference genome.' in r.stderr
    r = run(ngs, "qc", str(tmp_path / "a.sam"), GENOME)
    assert r.returncode == 1
    r = run(ngs, "qc", bam)
    assert r.returncode == 1 and "required arguments" in r.stderr
    r = run(ngs, "view", bam)
    assert r.returncode == 1


def test_return_when_done_hands_on_a_failure(ngs, lib, tmp_path):
    """NGSQ_RETURN_WHEN_DONE=1 (one process): the scan runs in a forked child and the command returns when the document is on disk;
    a child that ends BEFORE it has reported hands its exit status on -- here, on a box without a GPU, the missing device."""
    if lib.ngsq_device_count() > 0:
        pytest.skip("needs a box without a HIP device (the GPU suite checks the successful return)")
    hb = sorted_batch(1, 50)
    bam = str(tmp_path / "a.bam")
    bamio.write_bam(bam, hb, NAMES, LENS)
    r = run(ngs, "qc", bam, GENOME, "-o", str(tmp_path), env=dict(os.environ, NGSQ_RETURN_WHEN_DONE="1"))
    assert r.returncode == 1 and "no HIP device" in r.stderr
    r = run(ngs, "qc", bam, "hg19", env=dict(os.environ, NGSQ_RETURN_WHEN_DONE="1"))     # (fails before the fork)
    assert r.returncode == 1 and "reference genome is not supported" in r.stderr


def test_cli_surface_is_the_reference_s(ngs, tmp_path):
    """tests/golden/qc_cli_surface.json is derived from the reference's clap definition (src/qc/command.rs:36-102):
    every option there -- long name, short letter, value name, default -- is in this build's `ngs qc --help`, and the
    parser accepts it (with everything given but the BAM, what is missing is a positional, not an unknown option)."""
    surface = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "qc_cli_surface.json")))
    r = run(ngs, "qc", "--help")
    assert r.returncode == 0
    usage = r.stderr + r.stdout
    assert [p["value_name"] for p in surface["positionals"]] == ["BAM", "REFERENCE_GENOME"]
    assert "<BAM> <REFERENCE_GENOME>" in usage
    args = []
    for o in surface["options"]:
        flag = f"--{o['long']} <{o['value_name']}>"
        line = next((ln for ln in usage.splitlines() if flag in ln), None)
        assert line is not None, f"{flag} missing from the usage text"
        if o["short"]:
            assert f"-{o['short']}, --{o['long']}" in line
        if o["default"]:
            assert f"[default: {o['default']}]" in line
        args += [f"--{o['long']}", "7" if o["value_name"] == "USIZE" else str(tmp_path / "x")]
        if o["short"]:
            args += [f"-{o['short']}", "7" if o["value_name"] == "USIZE" else str(tmp_path / "x")]
    r = run(ngs, "qc", *args)
    assert r.returncode == 1 and "required arguments" in r.stderr and "unexpected argument" not in r.stderr
    r = run(ngs, "qc", "--no-such-option", "x")
    assert r.returncode == 1 and "unexpected argument '--no-such-option' found" in r.stderr


def test_format_sniffing_known_answers(ngs):
    """tests/golden/format_sniffing.json holds the reference's own tests of BioinformaticsFileFormat::try_detect
    (utils/formats.rs:192-322) and its Display names.  `ngs qc` opens its BAM through the same sniffing
    (formats/bam.rs:32-56): a BAM name passes (and the missing file is then reported), every other known format is
    'incompatible formats: required BAM, found <name>', an unknown extension is reported as such."""
    fx = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "format_sniffing.json")))
    assert len(fx["cases"]) >= 20
    for case in fx["cases"]:
        r = run(ngs, "qc", "/nonexistent/" + case["file"], GENOME)
        assert r.returncode == 1
        if case["format"] == "BAM":
            assert "opening BAM file" in r.stderr, r.stderr
        else:
            assert f"incompatible formats: required BAM, found {fx['display'][case['format']]}" in r.stderr, (case, r.stderr)
    r = run(ngs, "qc", "/nonexistent/sample.SAM", GENOME)             # extensions match case-insensitively ...
    assert "required BAM, found SAM" in r.stderr
    r = run(ngs, "qc", "/nonexistent/sample.FA.GZ", GENOME)           # ... the names in front of .gz do not
    assert "Not able to determine filetype for extension: GZ" in r.stderr
    r = run(ngs, "qc", "/nonexistent/sample.txt", GENOME)
    assert "Not able to determine filetype for extension: txt" in r.stderr


# messages of the reference this build does not print, and why (everything else in the fixture must be carried)
NOT_CARRIED = {
    "Too many facets matched": "cannot happen: facet names are distinct",
    "constructing BAM index filepath": "string concatenation cannot fail",
    "parsing BAM header": "the header is read and parsed in one step ('reading BAM header')",
    "writing VAF file header": "written with the file's creation ('creating VAF file')",
    "writing VAF file": "stdio errors surface at close",
    "Could not parse read name": "noodles decode step; read names are not decoded on this path",
    "could not lookup reference sequence for read": "counted on the device (ngsq_error_counts.edits_bad_ref), reported by ngsq_finalize",
    "Could not parse reference sequence id for read": "counted on the device (features_missing_reference_id)",
    "Could not map reference sequence id to header for read": "ids outside the header are refused when the file is read",
    "Could not parse record's start position.": "counted on the device (features_missing_position)",
}


def test_error_texts_are_the_reference_s():
    """tests/golden/qc_error_texts.json lists every bail!/anyhow!/with_context literal on the reference's `ngs qc` path
    (tests/golden/make_error_texts.py).  Each is either in this build's sources word for word (formatted values
    aside) or named above with the reason it cannot occur here."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = ""
    for rel in ("ngs_amd/csrc/cli/ngs_main.cpp", "ngs_amd/csrc/cli/gff_loader.h", "ngs_amd/csrc/reference.cpp", "ngs_amd/csrc/bam_reader.cpp",
                "oracle/oracle.c", "oracle/histogram.c"):
        src = open(os.path.join(root, rel)).read()
        src = re.sub(r'"\s*\n\s*"', "", src)          # adjacent C string literals are one string
        text += src.replace('\\"', '"')
    msgs = json.load(open(os.path.join(root, "tests", "golden", "qc_error_texts.json")))["messages"]
    assert len(msgs) >= 25
    carried = 0
    for m in msgs:
        skip = next((why for key, why in NOT_CARRIED.items() if m["text"].startswith(key) or key in m["text"]), None)
        parts = [p for p in m["text"].split("{}") if p.strip(" .")]
        have = all(p in text or p.rstrip(".") in text for p in parts)
        if skip is None:
            assert have, f"{m['file']}:{m['line']}: {m['text']!r} is not in this build's sources"
            carried += 1
    assert carried >= 15


def oracle_json(oracle_mod, hb, facets=ffi.FACETS_DEFAULT, ref_bases=None, pass1=None, pass2=None):
    o = oracle_mod.Oracle(LENS, PRIMARY, facets=facets, max_read_len=1024, gc_seed=0x4E4753, ref_bases=ref_bases)
    if pass1 is None:
        o.process_batch(hb)
    else:
        o.process_batch(pass1, ffi.PASS_RECORD)
        for b in pass2:
            o.process_batch(b, ffi.PASS_SEQUENCE)
    o.finalize()
    return o.results(NAMES)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", ["uniform150", "ragged"])
def test_end_to_end(ngs, gpu_lib, oracle_mod, tmp_path, shape):
    hb = sorted_batch(3, 6000) if shape == "uniform150" else sorted_batch(4, 6000, max_len=260, min_len=30)
    bam = str(tmp_path / "sample.bam")
    hb = bamio.with_ids(hb, bamio.write_bam(bam, hb, NAMES, LENS, block_payload=20_000))
    out = tmp_path / "out"
    r = run(ngs, "-v", "qc", bam, GENOME, "-o", str(out), "--batch-records", "1700")
    assert r.returncode == 0, r.stderr
    assert "Processed 6,000 records in the first pass." in r.stderr
    got = json.load(open(out / "sample.bam.results.json"))     # default prefix = BAM file name
    json_equal(got, oracle_json(oracle_mod, hb))
    from tests import literal_model as lm                       # (the second reading of the source judges the command too)
    json_equal(got, lm.run(lm.records_of(hb), NAMES, LENS, PRIMARY, gc_seed=0x4E4753))
    assert "chrM" not in got["coverage"]["mean_coverage"]      # not part of the primary assembly
    # NGSQ_RETURN_WHEN_DONE=1: the command returns when the document is on disk (the scan in a forked child); same document
    out2 = tmp_path / "out2"
    r = run(ngs, "-q", "qc", bam, GENOME, "-o", str(out2), "--batch-records", "1700", env=dict(os.environ, NGSQ_RETURN_WHEN_DONE="1"))
    assert r.returncode == 0, r.stderr
    json_equal(json.load(open(out2 / "sample.bam.results.json")), got)
    # --only + -p, written to the current directory
    r = run(ngs, "-q", "qc", bam, GENOME, "--only", "gc content", "-p", "x", cwd=str(tmp_path))
    assert r.returncode == 0 and r.stderr == ""
    g2 = json.load(open(tmp_path / "x.results.json"))
    assert g2["gc_content"] == got["gc_content"] and all(g2[k] is None for k in g2 if k != "gc_content")


@pytest.mark.gpu
@pytest.mark.parametrize("gpus", [1, 3])
@pytest.mark.parametrize("index", ["bare", "real"])
@pytest.mark.parametrize("n", [0, 1, 7, 300, 100_000])
def test_num_records_rules(ngs, gpu_lib, oracle_mod, tmp_path, n, index, gpus):
    """-n: pass 1 stops after n records; pass 2 shares ONE counter over all sequences, so every
    sequence after the n-th record still processes one record (command.rs:354,384-388).  With a real BAI the
    sequence pass runs as region queries through the index (seek to the sequence's first chunk); with an index
    that holds no bins it scans the file: same document."""
    hb = sorted_batch(5, 2500)
    bam = str(tmp_path / "s.bam")
    hb = bamio.with_ids(hb, bamio.write_bam(bam, hb, NAMES, LENS, block_payload=3000 if index == "real" else 60000, real_index=index == "real"))
    # --gpus 3: worker 0 applies both rules, the others bring an empty state to the exchange -- same document
    r = run(ngs, "-v", "qc", bam, GENOME, "-n", str(n), "-o", str(tmp_path), "--batch-records", "999",
            *(["--gpus", str(gpus), "--same-device"] if gpus > 1 else []))
    assert r.returncode == 0, r.stderr
    assert ("region queries through the index" in r.stderr) == (index == "real")
    got = json.load(open(tmp_path / "s.bam.results.json"))
    # emulate the reference driver on the record list
    c = hb.cols
    keep1 = min(max(n, 1), hb.n)
    picks, counter = [], 0
    for ref in range(len(NAMES)):
        for i in range(hb.n):
            if c["ref_id"][i] != ref or c["pos"][i] < 0:
                continue
            ops = c["cigar"][int(c["cigar_off"][i]):int(c["cigar_off"][i + 1])]
            span = sum(int(x) >> 4 for x in ops if (int(x) & 15) in (0, 2, 3, 7, 8))
            s = int(c["pos"][i]) + 1
            if s + span - 1 == 0 or s > LENS[ref]:
                continue                      # not yielded by query()
            picks.append(i)
            counter += 1
            if counter >= n:
                break
    want = oracle_json(oracle_mod, hb, pass1=hb.slice(0, keep1), pass2=[hb.slice(i, i + 1) for i in picks])
    json_equal(got, want)
    # the second reading of the source (tests/literal_model.py) with its own `-n`: the command's document, directly
    from tests import literal_model as lm
    json_equal(got, lm.run(lm.records_of(hb), NAMES, LENS, PRIMARY, gc_seed=0x4E4753, num_records=n))


@pytest.mark.gpu
def test_edits_with_reference_fasta(ngs, gpu_lib, oracle_mod, tmp_path):
    from tests.util import batch_from_records
    rng = np.random.default_rng(6)
    letters = "=ACMGRSVTWYHKDBN"
    bases = [rng.choice(np.array([1, 2, 4, 8, 15], dtype=np.uint8), L, p=[.25, .25, .25, .24, .01]) for L in LENS]
    recs = []
    for _ in range(3000):
        ref = int(rng.integers(0, len(LENS)))
        pos = int(rng.integers(0, LENS[ref] - 120))
        codes = bases[ref][pos:pos + 100].copy()
        codes[rng.integers(0, 100, 2)] = 1   # up to two edits (some coincide with an A already there)
        recs.append(dict(flag=int(rng.choice([0, 0x40, 0x80, 0x400, 0x4])), mapq=30, ref_id=ref, pos=pos,
                         mate_ref_id=-1, tlen=0, cigar=str(rng.choice(["100M", "40M5D60M", "30M10S60M", "10S90M", "50M3I47M"])),
                         seq="".join(letters[x] for x in codes), qual=[int(x) for x in rng.integers(0, 60, 100)]))
    recs.sort(key=lambda r: (r["ref_id"], r["pos"]))
    hb = batch_from_records(recs)
    bam = str(tmp_path / "e.bam")
    hb = bamio.with_ids(hb, bamio.write_bam(bam, hb, NAMES, LENS))
    fa = tmp_path / "ref.fa"
    with open(fa, "w") as f:
        for name, b in zip(NAMES, bases):
            f.write(f">{name} test\n")
            s_ = "".join(letters[x] for x in b)
            for k in range(0, len(s_), 60):
                f.write(s_[k:k + 60] + "\n")
    r = run(ngs, "-q", "qc", bam, GENOME, "-r", str(fa), "-o", str(tmp_path))
    assert r.returncode == 0, r.stderr
    got = json.load(open(tmp_path / "e.bam.results.json"))
    want = oracle_json(oracle_mod, hb, facets=ffi.FACETS_DEFAULT | ffi.FACET_EDITS, ref_bases=bases)
    json_equal(got, want)
    assert sum(got["edits"]["read_one_edits"]["values"][1:]) > 0 and sum(got["edits"]["vaf_histogram"]["values"]) > 0
    from tests import literal_model as lm   # the second reading of the source, fed the FASTA's letters
    json_equal(got, lm.run(lm.records_of(hb), NAMES, LENS, PRIMARY, gc_seed=0x4E4753,
                           fasta={name: "".join(letters[x] for x in b).encode() for name, b in zip(NAMES, bases)}))
    # --vaf-file (edits.rs:134-151, :320-341): one line per covered position, in header order; binning the
    # written f32 values the way the facet does reproduces the VAF histogram; Rust's f32 Display
    vaf = tmp_path / "vafs.tsv"
    r = run(ngs, "-q", "qc", bam, GENOME, "-r", str(fa), "-o", str(tmp_path), "--vaf-file", str(vaf))
    assert r.returncode == 0, r.stderr
    lines = open(vaf).read().splitlines()
    assert lines[0] == "Sequence\tPosition\tVAF"
    hist = [0] * 101
    last = (-1, -1)
    for ln in lines[1:]:
        name, pos, val = ln.split("\t")
        key = (NAMES.index(name), int(pos))
        assert key > last
        last = key
        v = np.float32(val)
        assert val == np.format_float_positional(v, unique=True, trim="-") and "e" not in val
        hist[int(v * np.float32(100.0))] += 1
    assert hist == got["edits"]["vaf_histogram"]["values"]
    r = run(ngs, "-q", "qc", bam, GENOME, "-r", str(fa), "-o", str(tmp_path), "--vaf-file", str(vaf))
    assert r.returncode == 1 and "refusing to overwrite existing VAF file" in r.stderr
    # a sequence missing from the FASTA aborts like EditsFacet::setup (edits.rs:207-209)
    with open(fa, "w") as f:
        f.write(">chr1\n" + "".join(letters[x] for x in bases[0]) + "\n")
    r = run(ngs, "-q", "qc", bam, GENOME, "-r", str(fa), "-o", str(tmp_path))
    assert r.returncode == 1 and "sequence chr2 not found in reference FASTA." in r.stderr


@pytest.mark.gpu
def test_genomic_features_with_gff(ngs, gpu_lib, oracle_mod, tmp_path):
    """-f GFF: file -> gene model -> Genomic Features facet, against the oracle fed the same intervals;
    plus the facet's own error paths (formats/gff.rs:19-47, features.rs:291,310-312)."""
    import gzip
    hb = sorted_batch(5, 3000)
    c = hb.cols   # a mapped record needs a sequence and a position here (features.rs:132-140,171-174 bail otherwise)
    c["flag"][(c["pos"] < 0) | (c["ref_id"] < 0)] |= np.uint16(0x4)
    bam = str(tmp_path / "g.bam")
    hb = bamio.with_ids(hb, bamio.write_bam(bam, hb, NAMES, LENS))
    rng = np.random.default_rng(8)
    types = ["five_prime_UTR", "three_prime_UTR", "CDS", "exon", "gene", "transcript", "start_codon"]
    rows, model, every_row = ["##gff-version 3", "#comment"], [], []
    for _ in range(400):
        seq = int(rng.integers(0, 4))
        s = int(rng.integers(1, LENS[seq]))
        e = min(LENS[seq], s + int(rng.integers(0, 3000)))
        t = types[int(rng.integers(0, len(types)))]
        rows.append(f"{NAMES[seq]}\tHAVANA\t{t}\t{s}\t{e}\t.\t{'+-'[int(rng.integers(0, 2))]}\t.\tID=x")
        every_row.append((seq, t, s, e))
        if t in types[:5] and PRIMARY[seq]:
            model.append((seq, types.index(t), s, e))
    rows += ["chr9\tHAVANA\tgene\t5\t900\t.\t+\t.\tID=primary_but_not_in_this_bam", "##FASTA", ">junk", "ACGT"]
    gff = str(tmp_path / "m.gff3.gz")
    with gzip.open(gff, "wt") as f:
        f.write("\n".join(rows) + "\n")
    r = run(ngs, "-q", "qc", bam, GENOME, "-f", gff, "-o", str(tmp_path))
    assert r.returncode == 0, r.stderr
    got = json.load(open(str(tmp_path / "g.bam.results.json")))
    o = oracle_mod.Oracle(LENS, PRIMARY, facets=ffi.FACETS_DEFAULT | ffi.FACET_FEATURES, max_read_len=1024, gc_seed=0x4E4753)
    o.set_features(*[np.array(c) for c in zip(*model)])
    o.process_batch(hb)
    o.finalize()
    json_equal(got, o.results(NAMES))
    assert got["features"]["records"]["processed"] > 1000 and got["features"]["gene_regions"]["exonic_count"] > 0
    from tests import literal_model as lm   # the second reading sorts the GFF's rows into its stores itself (features.rs:288-341)
    json_equal(got, lm.run(lm.records_of(hb), NAMES, LENS, PRIMARY, gc_seed=0x4E4753, intervals=every_row, role_names=tuple(types[:5])))
    # --only selects it by its display name, case-insensitively (qc.rs:101-123)
    r = run(ngs, "-q", "qc", bam, GENOME, "-f", gff, "-o", str(tmp_path), "--only", "genomic features")
    assert r.returncode == 0, r.stderr
    only = json.load(open(str(tmp_path / "g.bam.results.json")))
    assert only["features"] == got["features"] and only["general"] is None and only["coverage"] is None
    # custom feature names: exons called "gene" make every genic read ... still genic, never exonic
    r = run(ngs, "-q", "qc", bam, GENOME, "-f", gff, "-o", str(tmp_path), "--only", "Genomic Features",
            "--exon-feature-name", "gene")
    assert r.returncode == 0, r.stderr
    same = json.load(open(str(tmp_path / "g.bam.results.json")))["features"]["gene_regions"]
    assert same["exonic_count"] == 0 and same["intronic_count"] == got["features"]["gene_regions"]["exonic_count"] + \
        got["features"]["gene_regions"]["intronic_count"]
    # error paths
    plain = str(tmp_path / "bad.gff")
    open(plain, "w").write("chr1\tx\tgene\t10\t20\t.\t.\t.\tID=1\n")
    r = run(ngs, "-q", "qc", bam, GENOME, "-f", plain, "-o", str(tmp_path))
    assert r.returncode == 1 and "attempted to parse strand from value: ." in r.stderr
    open(plain, "w").write("chr1\tx\tgene\t10\n")
    r = run(ngs, "-q", "qc", bam, GENOME, "-f", plain, "-o", str(tmp_path))
    assert r.returncode == 1 and "invalid GFF record on line 1" in r.stderr
    r = run(ngs, "-q", "qc", bam, GENOME, "-f", str(tmp_path / "m.bed"), "-o", str(tmp_path))
    assert r.returncode == 1 and "opening GFF file" in r.stderr


@pytest.mark.gpu
def test_coverage_modes_and_unsorted_fallback(ngs, gpu_lib, oracle_mod, tmp_path):
    """--coverage stream / array / auto give the same document on a sorted file; a file whose header claims
    SO:coordinate but whose records are not in order is re-run on the depth arrays (auto) or refused (stream)."""
    hb = sorted_batch(9, 20_000)
    bam = str(tmp_path / "s.bam")
    hb = bamio.with_ids(hb, bamio.write_bam(bam, hb, NAMES, LENS, block_payload=30_000))
    docs = {}
    for mode in ("auto", "stream", "array"):
        out = tmp_path / mode
        r = run(ngs, "-q", "qc", bam, GENOME, "-o", str(out), "--coverage", mode, "--batch-records", "7001")
        assert r.returncode == 0, r.stderr
        docs[mode] = json.load(open(out / "s.bam.results.json"))
    json_equal(docs["stream"], oracle_json(oracle_mod, hb))
    json_equal(docs["array"], docs["stream"])
    json_equal(docs["auto"], docs["stream"])
    # same records, two of them swapped: not sorted any more
    from tests.util import take_records
    order = np.arange(hb.n)
    order[[5000, 15000]] = [15000, 5000]
    bad = take_records(hb, order)
    ubam = str(tmp_path / "u.bam")
    bamio.write_bam(ubam, bad, NAMES, LENS)
    r = run(ngs, "qc", ubam, GENOME, "-o", str(tmp_path / "u1"), "--coverage", "stream")
    assert r.returncode == 1 and "coordinate order" in r.stderr
    r = run(ngs, "qc", ubam, GENOME, "-o", str(tmp_path / "u2"))
    assert r.returncode == 0 and "scanning again with --coverage array" in r.stderr, r.stderr
    got = json.load(open(tmp_path / "u2" / "u.bam.results.json"))
    json_equal(got["coverage"], docs["array"]["coverage"])          # coverage does not depend on the record order
    json_equal(got["general"], docs["array"]["general"])


@pytest.mark.gpu
def test_gpus_flag_workers_share_one_file(ngs, gpu_lib, oracle_mod, tmp_path):
    """`ngs qc --gpus 3`: three worker processes (here on one device, exchange through shared memory; with one
    device each it is RCCL), each ingesting its BGZF block range; the document equals the single-process run.
    A file that breaks the promised order is scanned again on the depth arrays by every worker alike."""
    hb = sorted_batch(13, 30_000, max_len=200, min_len=40)
    bam = str(tmp_path / "g.bam")
    hb = bamio.with_ids(hb, bamio.write_bam(bam, hb, NAMES, LENS, block_payload=9_000))
    want = oracle_json(oracle_mod, hb)
    for mode in ("auto", "array"):
        out = tmp_path / ("w_" + mode)
        r = run(ngs, "qc", bam, GENOME, "-o", str(out), "--gpus", "3", "--same-device", "--coverage", mode,
                "--batch-records", "4001")
        assert r.returncode == 0, r.stderr
        assert "Worker 0 of 3 on device 0, exchange over shm." in r.stderr
        json_equal(json.load(open(out / "g.bam.results.json")), want)
    # the RCCL transport (what a node with a device per worker uses), librccl replaced by tests/rccl_double so that the
    # three workers can share this box's GPU
    from tests.test_shard_gloo import rccl_double_path
    out = tmp_path / "w_rccl"
    r = run(ngs, "qc", bam, GENOME, "-o", str(out), "--gpus", "3", "--same-device", "--transport", "rccl", "--batch-records", "4001",
            env=dict(os.environ, NGSQ_RCCL_LIB=rccl_double_path()))
    assert r.returncode == 0, r.stderr
    assert "Worker 0 of 3 on device 0, exchange over rccl." in r.stderr
    json_equal(json.load(open(out / "g.bam.results.json")), want)
    # RCCL that does not come up (the real library refuses three ranks on one device; so does a rank whose library is
    # missing): named, the run fails on every worker; not named (auto: what a device per worker gets without the option),
    # the workers agree on the shared-memory transport they met in and the document is the same
    r = run(ngs, "qc", bam, GENOME, "-o", str(tmp_path / "w_fail"), "--gpus", "3", "--same-device", "--transport", "rccl")
    assert r.returncode != 0 and ("ncclCommInitRank" in r.stderr or "RCCL" in r.stderr), r.stderr
    for lib_env in ({}, {"NGSQ_RCCL_LIB": "/nonexistent/librccl.so"}):
        out = tmp_path / ("w_auto%d" % len(lib_env))
        r = run(ngs, "qc", bam, GENOME, "-o", str(out), "--gpus", "3", "--same-device", "--transport", "auto", "--batch-records", "4001",
                env=dict(os.environ, **lib_env))
        assert r.returncode == 0, r.stderr
        assert "the exchange runs over shared memory instead" in r.stderr
        json_equal(json.load(open(out / "g.bam.results.json")), want)
    # more workers than devices without --same-device
    if gpu_lib.ngsq_device_count() < 3:
        r = run(ngs, "qc", bam, GENOME, "-o", str(tmp_path), "--gpus", "3")
        assert r.returncode == 1 and "device(s) visible" in r.stderr
    from tests.util import take_records
    order = np.arange(hb.n)
    order[[7000, 23000]] = [23000, 7000]
    ubam = str(tmp_path / "gu.bam")
    bamio.write_bam(ubam, take_records(hb, order), NAMES, LENS, block_payload=9_000)
    r = run(ngs, "qc", ubam, GENOME, "-o", str(tmp_path / "wu"), "--gpus", "3", "--same-device")
    assert r.returncode == 0 and "scanning again with --coverage array" in r.stderr, r.stderr
    got = json.load(open(tmp_path / "wu" / "gu.bam.results.json"))
    json_equal(got["coverage"], want["coverage"])
    json_equal(got["general"], want["general"])


@pytest.mark.gpu
def test_gpus_8_workers_on_one_device(ngs, gpu_lib, oracle_mod, tmp_path):
    """`ngs qc --gpus 8 --same-device`: the launcher, eight workers, the eight-way reader split of the CPU quota
    (bam_device_reader.cpp: max(2, (quota - 2 n) / n) threads each), seven boundaries, one exchange -- over shared memory and
    over the RCCL transport bound to tests/rccl_double; the document equals the oracle's."""
    hb = sorted_batch(17, 64_000, max_len=200, min_len=40)
    bam = str(tmp_path / "g8.bam")
    hb = bamio.with_ids(hb, bamio.write_bam(bam, hb, NAMES, LENS, block_payload=9_000))
    want = oracle_json(oracle_mod, hb)
    from tests.test_shard_gloo import rccl_double_path
    for k, (extra, env) in enumerate(((["--coverage", "auto"], None), (["--coverage", "array"], None),
                                      (["--transport", "rccl"], dict(os.environ, NGSQ_RCCL_LIB=rccl_double_path())))):
        out = tmp_path / f"w{k}"
        r = run(ngs, "qc", bam, GENOME, "-o", str(out), "--gpus", "8", "--same-device", "--batch-records", "3001", *extra, env=env)
        assert r.returncode == 0, r.stderr
        assert "Worker 0 of 8 on device 0, exchange over %s." % ("rccl" if env else "shm") in r.stderr
        json_equal(json.load(open(out / "g8.bam.results.json")), want)


@pytest.mark.gpu
def test_baseline_config_0_general_only(ngs, gpu_lib, oracle_mod, tmp_path):
    """BASELINE.json configs[0] verbatim: `ngs qc` general-metrics-only (--only General) on a 10k-record 150 bp
    single-reference BAM (one @SQ chr1 LN:248956422, the synthetic workload's records written as a real BGZF BAM +
    BAI by ngsq_synth_write_bam), host ingest and device ingest, against the oracle on the same records."""
    import ctypes as C
    n, L = 10_000, 248_956_422
    cfg = host.synth_config(n, ref_len=L, n_refs=1)
    bam = str(tmp_path / "cfg0.bam")
    assert gpu_lib.ngsq_synth_write_bam(C.byref(cfg), bam.encode(), n, 6, 2) == 0
    hb = host.synth_host_batch(cfg, 0, n, gpu_lib)
    o = oracle_mod.Oracle([L], [1], facets=ffi.FACET_GENERAL, max_read_len=1024, gc_seed=0x4E4753)
    o.process_batch(hb)
    o.finalize()
    want = o.results(["chr1"])
    assert want["general"]["records"]["total"] == n and all(want[k] is None for k in want if k != "general")
    for ingest in ("host", "device"):
        out = tmp_path / ingest
        r = run(ngs, "qc", bam, GENOME, "--only", "General", "-o", str(out), "--ingest", ingest)
        assert r.returncode == 0, r.stderr
        assert "Processed 10,000 records in the first pass." in r.stderr
        assert "No facets specified that require second pass. Skipping..." in r.stderr
        json_equal(json.load(open(out / "cfg0.bam.results.json")), want)
