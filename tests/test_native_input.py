"""Row N1: the native reader (csrc/pf_input.cpp) and the Python restatement of
/root/reference/panfeed/input.py:274-468 (oracle/input_restatement.py), (1) both against tests/golden/n1.json.gz --
what the reference's OWN parse_gff / prep_data_n_fasta / set_input_output / iter_gene_clusters yielded in the build
container with a declared double of pyfaidx.Fasta (tools/gen_golden_n1.py) -- and (2) against each other on synthetic
on-disk pangenomes the goldens do not cover.  PARITY UNPINNED only for the double's five operations (contig lookup,
slice, reverse complement, reversal, str): see DESIGN.md."""
import gzip
import json
import os

import numpy as np
import pytest

from numpy_packer import build_batch

from oracle import input_restatement as ir
from panfeed_amd import native_input as ni
from panfeed_amd import packing, synth


@pytest.fixture(scope="module")
def pangenome(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("pangenome"))
    cl = synth.generate(30, 12, flank=0, mean_len=300, min_len=40, max_len=900, n_rate=0.05, paralog_rate=0.1,
                        shuffle_columns=3)
    names = cl[0].names
    csvp, gffs, fas = synth.write_pangenome(d, cl, drop_gff_for=(names[2],), separate_fasta_for=(names[4], names[5]))
    gn = sorted(gffs)
    return dict(dir=d, csv=csvp, names=names, genomes=gn, gff=[gffs[n] for n in gn], fasta=[fas[n] for n in gn])


def _expected(p, up, down, dsc, genes=None, log=None):
    strains, table = ir.load_table(p["csv"])
    data = ir.load_genomes(p["genomes"], p["gff"], p["fasta"])
    return strains, list(ir.iter_gene_clusters(strains, table, data, up, down, dsc, gene_list=genes, log=log))


def _open(p, up=0, down=0, dsc=False, **kw):
    return ni.Pangenome(p["csv"], None, None, up, down, dsc, genome_names=p["genomes"], gff_paths=p["gff"],
                        fasta_paths=p["fasta"], **kw)


@pytest.mark.parametrize("up,down,dsc", [(0, 0, False), (50, 30, False), (200, 200, False), (20, 40, True),
                                         (0, 10, True)])
def test_records_equal_restatement(pangenome, up, down, dsc):
    log = []
    strains, exp = _expected(pangenome, up, down, dsc, log=log)
    with _open(pangenome, up, down, dsc) as pg:
        assert pg.strains == strains and pg.sorted_strains == sorted(strains)
        assert pg.n_clusters == len(exp)
        got = list(pg.records(7))
        nlog = pg.take_log()
    assert len(got) == len(exp)
    nseq = 0
    for (g, gi, gp), (e, ei, ep) in zip(got, exp):
        assert gi == ei
        assert gp.dtype == ep.dtype and (gp == ep).all()
        assert list(g.keys()) == list(e.keys())        # dict order: present strains in table order, absent sorted
        for s in g:
            assert g[s] == e[s], (gi, s)
            nseq += len(g[s])
    assert nseq > 200
    assert nlog.strip().split("\n") == log and len(log) >= 2      # the missing-GFF warning + the dropped CDS lines


def test_gene_list_filter(pangenome):
    strains, table = ir.load_table(pangenome["csv"])
    genes = [table[3][0], table[17][0], "not_a_cluster"]
    _s, exp = _expected(pangenome, 10, 10, False, genes=set(genes))
    with _open(pangenome, 10, 10, genes=genes) as pg:
        got = list(pg.records(4))
    assert [g[1] for g in got] == [e[1] for e in exp] == genes[:2]
    assert all(g[0] == e[0] for g, e in zip(got, exp))


@pytest.mark.parametrize("k,canon,ntg", [(31, True, 0), (15, False, 2), (40, True, 1)])
def test_batches_equal_packer_over_restated_records(pangenome, k, canon, ntg):
    """reader -> pf_pack_records by pointer == the numpy packer over the restated records"""
    tg = tuple(pangenome["names"][i] for i in (0, 7)[:ntg])
    _s, exp = _expected(pangenome, 40, 25, False)
    W = 1
    with _open(pangenome, 40, 25, targets=tg) as pg:
        hbs = list(pg.batches(k, canon, W, max_clusters=11, first_ordinal=5, with_names=True))
    pos = 0
    for hb in hbs:
        n = len(hb.idx)
        ref = build_batch(exp[pos:pos + n], k, canon, W, stroi=tg, first_ordinal=5 + pos)
        for f in ("packed", "seg_word_off", "seg_len", "seg_sample", "seg_ord_base", "seg_strand_off",
                  "cluster_seg_off", "extra_cluster", "extra_ord", "extra_bits", "cluster_nstrains",
                  "cluster_npresab", "cluster_presab", "cluster_ordinal"):
            assert np.array_equal(np.asarray(getattr(hb, f)), np.asarray(getattr(ref, f))), f
        assert hb.extra_keys == ref.extra_keys and hb.idx == ref.idx and hb.sorted_strains == ref.sorted_strains
        assert hb.n_instances == ref.n_instances and hb.n_strand_words == ref.n_strand_words
        assert all((x == y).all() for x, y in zip(hb.presab, ref.presab))
        assert [(x.cluster, x.strain, x.seq, x.segs, x.ambig) for x in hb.targets] == \
               [(y.cluster, y.strain, y.seq, y.segs, y.ambig) for y in ref.targets]
        pos += n
    assert pos == len(exp)


def test_discovery_and_directory_open(pangenome, tmp_path):
    """what_are_my_inputfiles (input.py:16-64): directory and file-of-files listings, .gff only, fasta needs a gff"""
    gffdir = os.path.join(pangenome["dir"], "gffs")
    names, with_fa, gffs, fastas = ni.what_are_my_inputfiles(gffdir, gffdir)
    assert names == pangenome["genomes"]
    assert with_fa == sorted(n for n, f in zip(pangenome["genomes"], pangenome["fasta"]) if f)
    fof = tmp_path / "gffs.txt"
    fof.write_text("\n".join(pangenome["gff"] + ["/x/readme.txt"]) + "\n")
    names2, _fa, gffs2, _ = ni.what_are_my_inputfiles(str(fof))
    assert names2 == names and gffs2 == gffs
    with pytest.raises(FileNotFoundError):
        ni.what_are_my_inputfiles(str(tmp_path))
    _s, exp = _expected(pangenome, 5, 5, False)
    with ni.Pangenome(pangenome["csv"], gffdir, gffdir, 5, 5) as pg:
        got = list(pg.records(100))
    assert all(g[0] == e[0] and g[1] == e[1] for g, e in zip(got, exp)) and len(got) == len(exp)


def test_reader_errors(pangenome, tmp_path):
    from panfeed_amd._lib import PanfeedHipError
    with pytest.raises(PanfeedHipError):
        ni.Pangenome(str(tmp_path / "nope.csv"), None, genome_names=["a"], gff_paths=[pangenome["gff"][0]])
    with pytest.raises(PanfeedHipError):
        ni.Pangenome(pangenome["csv"], None, genome_names=["a"], gff_paths=[str(tmp_path / "nope.gff")])
    bad = tmp_path / "nofasta.gff"
    bad.write_text("##gff-version 3\nc1\tx\tCDS\t1\t9\t.\t+\t0\tID=g1\n")
    with pytest.raises(PanfeedHipError):        # the reference's .split("##FASTA")[1] IndexError (input.py:104-107)
        ni.Pangenome(pangenome["csv"], None, genome_names=["a"], gff_paths=[str(bad)])


def test_oracle_refuses_mask_of_other_length():
    """panfeed.py:19 `v[clusterpresab == 0] = np.nan` with len(clusterpresab) != len(cluster): numpy IndexError"""
    from oracle import oracle as po
    from panfeed_amd.classes import Seqinfo
    gs = {"a": [Seqinfo("ACGTACGTAA", "TGCATGCATT", "g", "c", 1, 10, 1, 0)], "b": []}
    with pytest.raises(IndexError, match="boolean index did not match"):
        po.OracleRun(klength=5, consider_missing=True).feed([(gs, "x", np.array([1, 0, 1]))])
    po.OracleRun(klength=5, consider_missing=False).feed([(gs, "x", np.array([1, 0, 1]))])


def test_by_reference_batches_describe_the_same_input(pangenome):
    """resident-genome mode without a GPU: the reader hands out (contig, start, length, strand) instead of text for
    pure-ACGT sequences of non-target strains; cutting those ranges out of the contigs (what gather_segments_kernel
    does on the device) must rebuild, word for word, the packed buffer of the text mode; all other arrays equal"""
    import ctypes as C
    from panfeed_amd import _lib
    L = _lib.load()
    k, W = 21, 1
    tg = (pangenome["names"][0],)
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    with _open(pangenome, 40, 25, targets=tg) as pg:
        text = list(pg.batches(k, True, W, max_clusters=11))
    with _open(pangenome, 40, 25, targets=tg) as pg:
        n, ptrs, lens = C.c_uint32(), C.POINTER(C.c_char_p)(), C.POINTER(C.c_uint64)()
        _lib.check(L.pf_pangenome_contigs(pg.h, C.byref(n), C.byref(ptrs), C.byref(lens)))
        off = (C.c_uint64 * n.value)()
        contig_at, tot = {}, 0
        for i in range(n.value):                       # the layout pf_genomes_upload assigns
            off[i] = tot
            contig_at[tot] = C.string_at(ptrs[i], lens[i])
            tot += 2 * ((lens[i] + 63) // 64) + 2
        _lib.check(L.pf_pangenome_set_store(pg.h, off, n.value))
        pg.resident = True
        refd = list(pg.batches(k, True, W, max_clusters=11))
    assert len(text) == len(refd)
    n_ref = n_rev = n_lit = 0
    for a, b in zip(text, refd):
        for f in ("seg_word_off", "seg_len", "seg_sample", "seg_ord_base", "seg_strand_off", "cluster_seg_off",
                  "extra_cluster", "extra_ord", "extra_bits", "cluster_nstrains", "cluster_npresab", "cluster_presab"):
            assert np.array_equal(getattr(a, f), getattr(b, f)), f
        assert a.extra_keys == b.extra_keys and a.n_instances == b.n_instances and a.n_strand_words == b.n_strand_words
        assert [(x.cluster, x.strain, x.seq, x.segs, x.ambig) for x in a.targets] == \
               [(y.cluster, y.strain, y.seq, y.segs, y.ambig) for y in b.targets]
        assert b.n_words_dev == len(a.packed)
        rebuilt = np.zeros(b.n_words_dev, dtype=np.uint64)
        for s in range(len(b.seg_len)):
            ln, wo = int(b.seg_len[s]), int(b.seg_word_off[s])
            nw = 2 * ((ln + 63) // 64)
            fl, so, st = int(b.gather_src_flags[s]), int(b.gather_src_off[s]), int(b.gather_src_start[s])
            if fl & 1:
                rebuilt[wo:wo + nw] = b.packed[so:so + nw]
                n_lit += 1
                continue
            seq = contig_at[so][st:st + ln]
            assert len(seq) == ln
            if fl & 2:
                seq = seq[::-1].translate(comp)
                n_rev += 1
            buf = (C.c_uint64 * max(nw, 1))()
            assert L.pf_pack_acgt(seq, ln, buf) == nw
            rebuilt[wo:wo + nw] = np.ctypeslib.as_array(buf)[:nw]
            n_ref += 1
        assert np.array_equal(rebuilt, a.packed)
    assert n_ref > 150 and n_rev > 50 and n_lit > 10          # text stays for the target strain and the N-carrying ones


def test_parallel_writer_makes_a_valid_pangenome(tmp_path):
    """synth.write_pangenome(workers=N) (bench.py's long end-to-end leg): every sample from its own random stream, written
    by forked processes; the reader and the restatement agree on it and every cluster's sequences are the generator's"""
    cl = synth.generate(9, 14, first=77, flank=10, mean_len=150, min_len=30, max_len=400, n_rate=0.05, paralog_rate=0.1)
    csvp, gffs, fas = synth.write_pangenome(str(tmp_path), cl, seed=5, workers=3, missing_gene_rate=0.0, lower_rate=0.0)
    gn = sorted(gffs)
    p = dict(csv=csvp, genomes=gn, gff=[gffs[n] for n in gn], fasta=[fas[n] for n in gn])
    _s, exp = _expected(p, 10, 10, False)
    with _open(p, 10, 10) as pg:
        got = list(pg.records(4))
    assert len(got) == len(exp) == len(cl)
    for (g, gi, gp), (e, ei, ep), c in zip(got, exp, cl):
        assert gi == ei == c.idx and (gp == ep).all() and g == e
        want = sorted(c.seq_string(q) for q in range(c.n_seqs))
        assert sorted(s.sequence for seqs in g.values() for s in seqs) == want


@pytest.mark.parametrize("seed", range(12))
def test_reader_fuzz_against_restatement(tmp_path, seed):
    """random small pangenomes and options: native reader == Python restatement, record for record"""
    rng = np.random.default_rng(1000 + seed)
    S = int(rng.integers(3, 20))
    cl = synth.generate(int(rng.integers(1, 12)), S, first=int(rng.integers(0, 10**5)), flank=0,
                        mean_len=int(rng.choice([30, 120, 600])), min_len=int(rng.choice([3, 20])), max_len=900,
                        n_rate=float(rng.choice([0.0, 0.1, 0.5])), paralog_rate=float(rng.choice([0.0, 0.2])),
                        shuffle_columns=int(rng.integers(0, 50)) if rng.random() < 0.5 else None)
    names = cl[0].names
    drop = tuple(rng.choice(names, size=int(rng.integers(0, 2)), replace=False).tolist())
    sep = tuple(n for n in names if rng.random() < 0.3 and n not in drop)
    csvp, gffs, fas = synth.write_pangenome(str(tmp_path), cl, seed=seed, wrap=int(rng.choice([10, 60, 1000])), drop_gff_for=drop,
                                            missing_gene_rate=float(rng.choice([0.0, 0.05])),
                                            lower_rate=float(rng.choice([0.0, 0.3])), separate_fasta_for=sep)
    gn = sorted(gffs)
    p = dict(csv=csvp, genomes=gn, gff=[gffs[n] for n in gn], fasta=[fas[n] for n in gn])
    up, down, dsc = int(rng.choice([0, 7, 150, 5000])), int(rng.choice([0, 3, 90, 5000])), bool(rng.random() < 0.3)
    genes = None
    if rng.random() < 0.3:
        _s, table = ir.load_table(csvp)
        genes = [t[0] for t in table if rng.random() < 0.5] + ["absent_cluster"]
    log = []
    strains, exp = _expected(p, up, down, dsc, genes=set(genes) if genes is not None else None, log=log)
    with _open(p, up, down, dsc, genes=genes) as pg:
        got = list(pg.records(int(rng.integers(1, 6))))
        nlog = pg.take_log()
    assert len(got) == len(exp)
    for (g, gi, gp), (e, ei, ep) in zip(got, exp):
        assert gi == ei and (gp == ep).all() and list(g.keys()) == list(e.keys())
        for s in g:
            assert g[s] == e[s], (gi, s)
    assert [x for x in nlog.strip().split("\n") if x] == log


# ---------------------------------------------------------------------------------------------------------------------
# the reference's own reader code, run in the build container (tools/gen_golden_n1.py)
_N1 = json.load(gzip.open(os.path.join(os.path.dirname(__file__), "golden", "n1.json.gz"), "rt"))
_N1_CASES = {c["name"]: c for c in _N1["cases"]}


def _n1_setup(case, tmp_path):
    root = str(tmp_path)
    for rel, text in _N1["pangenomes"][case["pangenome"]].items():
        p = os.path.join(root, rel)
        os.makedirs(os.path.dirname(p), exist_ok=True)
        with open(p, "w", newline="") as fh:
            fh.write(text)
    e = case["expect"]
    # (set_input_output's "No target strains provided" belongs to the CLI surface, not to the reader)
    warnings = [w.replace("{DIR}", root) for w in e["warnings"] if w != "No target strains provided"]
    gffdir = os.path.join(root, "gffs")
    return root, gffdir, (gffdir if case["opts"]["fasta_dir"] else None), warnings


def _as_json(records):
    return [{"idx": idx, "presab": [int(x) for x in presab],
             "strains": [[nm, [[s.sequence, s.compsequence, s.id, s.chromosome, int(s.start), int(s.end), int(s.strand),
                                int(s.offset)] for s in seqs]] for nm, seqs in gs.items()]}
            for gs, idx, presab in records]


@pytest.mark.parametrize("name", sorted(_N1_CASES))
def test_restatement_equals_reference_goldens(name, tmp_path):
    case = _N1_CASES[name]
    o, e = case["opts"], case["expect"]
    root, gffdir, fastadir, warnings = _n1_setup(case, tmp_path)
    names, with_fa, gffs, fastas = ni.what_are_my_inputfiles(gffdir, fastadir)
    assert names == e["filelist"] and with_fa == e["fastalist"]
    strains, table = ir.load_table(os.path.join(root, "gene_presence_absence.csv"))
    assert strains == e["strains"]
    log = []
    data = ir.load_genomes(names, [gffs[n] for n in names], [fastas.get(n) for n in names], log=log)
    feats = {g: {k: [f.id, f.chromosome, f.start, f.end, f.strand] for k, f in data[g][1].items()} for g in names}
    assert feats == e["features"]                       # parse_gff, input.py:274-332 (the last field keeps its newline)
    got, raised = [], None
    try:
        for rec in ir.iter_gene_clusters(strains, table, data, o["up"], o["down"], o["dsc"],
                                         gene_list=set(o["genes"]) if o["genes"] is not None else None, log=log,
                                         raise_missing=o["raise_missing"]):
            got.append(rec)
    except KeyError as ex:
        raised = ex.args[0]
    assert raised == e["raises"]
    assert _as_json(got) == e["records"]
    assert log == warnings


@pytest.mark.parametrize("name", sorted(_N1_CASES))
def test_native_reader_equals_reference_goldens(name, tmp_path):
    from panfeed_amd._lib import PanfeedHipError
    case = _N1_CASES[name]
    o, e = case["opts"], case["expect"]
    root, gffdir, fastadir, warnings = _n1_setup(case, tmp_path)
    got, raised, nlog = [], None, ""
    try:
        with ni.Pangenome(os.path.join(root, "gene_presence_absence.csv"), gffdir, fastadir, o["up"], o["down"], o["dsc"],
                          targets=o["targets"] or (), genes=o["genes"], raise_missing=o["raise_missing"]) as pg:
            assert pg.strains == e["strains"] and pg.sorted_strains == sorted(e["strains"])
            try:
                for rec in pg.records(1):               # one row per call: rows before a raising one are yielded, as the generator does
                    got.append(rec)
            finally:
                nlog = pg.take_log()
    except PanfeedHipError as ex:
        raised = str(ex)
    if e["raises"] is None:
        assert raised is None
        assert [x for x in nlog.split("\n") if x] == [x for w in warnings for x in w.split("\n") if x]
    else:
        assert raised is not None and e["raises"] in raised
    assert _as_json(got) == e["records"]


def _mangle_fasta(path, rng, style):
    """re-wrap the FASTA of a GFF (or FASTA) file without changing a letter's meaning: what parse_fasta -- and pyfaidx --
    make of the file stays the same"""
    raw = open(path, "rb").read()
    cut = raw.find(b"##FASTA")
    head, fa = (raw[:cut + 8], raw[cut + 8:]) if cut >= 0 else (b"", raw)
    recs, name, chunks = [], None, []
    for line in fa.split(b"\n"):
        if line.startswith(b">"):
            if name is not None:
                recs.append((name, b"".join(chunks)))
            name, chunks = line, []
        elif name is not None:
            chunks.append(line.rstrip(b"\r"))
    if name is not None:
        recs.append((name, b"".join(chunks)))
    out = [head]
    eol = b"\r\n" if style == "crlf" else b"\n"
    for i, (hdr, seq) in enumerate(recs):
        out.append(hdr + (b" description with spaces" if style == "ragged" else b"") + eol)
        if style == "ragged":
            at = 0
            while at < len(seq):
                n = int(rng.integers(1, 97))
                out.append(seq[at:at + n] + eol)
                at += n
                if rng.random() < 0.1:
                    out.append(eol)                                   # a blank line inside a record
        elif style == "lower":
            s2 = bytearray(seq)
            for _ in range(3):
                a = int(rng.integers(0, max(len(s2), 1)))
                s2[a:a + 40] = bytes(s2[a:a + 40]).lower()
            for at in range(0, len(s2), 70):
                out.append(bytes(s2[at:at + 70]) + eol)
        elif style == "oneline":
            out.append(seq + eol)
        else:
            w = 80 if style == "crlf" else 60
            for at in range(0, len(seq), w):
                out.append(seq[at:at + w] + eol)
            if style == "trailing":
                out.append(eol + eol)
        if style == "ragged" and i == 0:
            out.append(b">emptyrecord" + eol)                           # a record without a letter
    if recs and style in ("ragged", "trailing"):
        out.append(recs[0][0] + eol + b"ACGTACGTNNACGT" + eol)          # a repeated name: the first record stays
    blob = b"".join(out)
    if style == "oneline":
        blob = blob.rstrip(b"\n")                                       # no newline at the end of the file
    open(path, "wb").write(blob)


def _decode_store(words, word_off, start, n):
    """n letters from base `start` of the contig whose words begin at word_off (2 bits per base, first base in bits 63:62)"""
    j = np.arange(start, start + n, dtype=np.uint64)
    w = words[(np.uint64(word_off) + (j >> np.uint64(5))).astype(np.int64)]
    code = (w >> (np.uint64(62) - np.uint64(2) * (j & np.uint64(31)))) & np.uint64(3)
    return np.frombuffer(b"ACGT", dtype=np.uint8)[code.astype(np.int64)].tobytes()


_STYLES = ["plain", "crlf", "ragged", "lower", "oneline", "trailing"]


def check_one_pass(root, style_of, mangle_seed=3, synth_kw=None, up=30, down=20, k=21, max_clusters=5, strict=True):
    """the body of test_one_pass_reader_describes_the_same_input; tests/fuzz_reader.py draws its arguments at random
    (style_of: genome name -> one of _STYLES)"""
    import ctypes as C
    from panfeed_amd import _lib
    L = _lib.load()
    rng = np.random.default_rng(mangle_seed)
    kw = dict(first=500, flank=0, mean_len=260, min_len=40, max_len=700, n_rate=0.08, paralog_rate=0.1)
    kw.update(synth_kw or {})
    n_clusters, n_samples = kw.pop("n_clusters", 14), kw.pop("n_samples", 9)
    cl = synth.generate(n_clusters, n_samples, **kw)
    names = cl[0].names
    csvp, gffs, fas = synth.write_pangenome(str(root), cl, separate_fasta_for=(names[min(3, len(names) - 1)],), missing_gene_rate=0.0,
                                            lower_rate=0.0)
    gn = sorted(gffs)
    p = dict(csv=csvp, names=names, genomes=gn, gff=[gffs[n] for n in gn], fasta=[fas[n] for n in gn])
    W = (len(names) + 31) // 32
    tg = (names[1],)
    with _open(p, up, down, targets=tg) as pg:
        text = list(pg.batches(k, True, W, max_clusters=max_clusters))
    if any(style_of(n) != "plain" for n in gn):
        for n in gn:
            if style_of(n) != "plain":
                _mangle_fasta(fas[n] or gffs[n], rng, style_of(n))
        with _open(p, up, down, targets=tg) as pg:                      # (parse_fasta on the mangled files: the same input)
            again = list(pg.batches(k, True, W, max_clusters=max_clusters))
        assert all(np.array_equal(a.packed, b.packed) and np.array_equal(a.seg_len, b.seg_len) for a, b in zip(text, again))
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    with _open(p, up, down, targets=tg, debug_hostsink=True) as pg:
        assert pg.resident and pg.one_pass
        store = pg.store_words
        refd = list(pg.batches(k, True, W, max_clusters=max_clusters))
        if strict:                                                      # (a fuzz case may keep every contig as text)
            for _ in range(2):                                          # its contigs are not text any more -- ask twice
                with pytest.raises(_lib.PanfeedHipError):
                    n, ptrs, lens = C.c_uint32(), C.POINTER(C.c_char_p)(), C.POINTER(C.c_uint64)()
                    _lib.check(L.pf_pangenome_contigs(pg.h, C.byref(n), C.byref(ptrs), C.byref(lens)))
    assert len(text) == len(refd)
    n_ref = n_rev = n_lit = 0
    for a, b in zip(text, refd):
        for f in ("seg_word_off", "seg_len", "seg_sample", "seg_ord_base", "seg_strand_off", "cluster_seg_off",
                  "extra_cluster", "extra_ord", "extra_bits", "cluster_nstrains", "cluster_npresab", "cluster_presab"):
            assert np.array_equal(getattr(a, f), getattr(b, f)), f
        assert a.extra_keys == b.extra_keys and a.n_instances == b.n_instances
        assert [(x.cluster, x.strain, x.seq, x.segs, x.ambig) for x in a.targets] == \
               [(y.cluster, y.strain, y.seq, y.segs, y.ambig) for y in b.targets]
        assert b.n_words_dev == len(a.packed)
        rebuilt = np.zeros(b.n_words_dev, dtype=np.uint64)
        for s in range(len(b.seg_len)):
            ln, wo = int(b.seg_len[s]), int(b.seg_word_off[s])
            nw = 2 * ((ln + 63) // 64)
            fl, so, st = int(b.gather_src_flags[s]), int(b.gather_src_off[s]), int(b.gather_src_start[s])
            if fl & 1:
                rebuilt[wo:wo + nw] = b.packed[so:so + nw]
                n_lit += 1
                continue
            seq = _decode_store(store, so, st, ln)
            if fl & 2:
                seq = seq[::-1].translate(comp)
                n_rev += 1
            buf = (C.c_uint64 * max(nw, 1))()
            assert L.pf_pack_acgt(seq, ln, buf) == nw
            rebuilt[wo:wo + nw] = np.ctypeslib.as_array(buf)[:nw]
            n_ref += 1
        assert np.array_equal(rebuilt, a.packed)
    if strict:
        assert n_ref > 60 and n_rev > 15 and n_lit > 8
    return n_ref, n_rev, n_lit


@pytest.mark.parametrize("style", _STYLES)
def test_one_pass_reader_describes_the_same_input(tmp_path, style):
    """pf_pangenome_open_device's reader side (scan_fasta through the host sink): contigs measured where their letters lie
    -- evenly wrapped, CRLF, lines of any length with blank ones between (joined in place), lower case, one line without a
    newline at the end of the file, blank lines at the end, a repeated contig name, a record without letters -- and packed
    by the kernel's addressing.  The by-reference batches must rebuild, word for word, the packed buffer the text-mode
    reader makes of the UNMANGLED files; everything else equal; text kept exactly for 'N'-carrying contigs and the target
    strain."""
    check_one_pass(tmp_path, lambda n: style)
