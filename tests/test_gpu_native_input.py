"""Row N1 end to end on the GPU: files on disk -> native reader -> packer -> HIP path -> the three TSV texts,
against the oracle run over the Python restatement's records (oracle/input_restatement.py; parity unpinned at
the pyfaidx boundary)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("resident", [False, True], ids=["text", "resident"])
@pytest.mark.parametrize("k,canon,up,down,dsc,cm", [(31, True, 0, 0, False, False), (21, False, 60, 40, False, True),
                                                  (25, True, 30, 50, True, False), (40, True, 300, 300, False, False)])
def test_files_to_tsv(tmp_path, k, canon, up, down, dsc, cm, resident):
    """resident: contigs live on the GPU as 2 bits per base (pf_genomes_upload) and the segments are cut out of them
    -- or reverse-complemented -- by gather_segments_kernel; the host sends coordinates, not bases"""
    from oracle import input_restatement as ir
    from oracle import oracle as po
    from panfeed_amd import native_input as ni
    from panfeed_amd import synth
    from panfeed_amd.engine import Engine
    cl = synth.generate(40, 70, first=77, flank=0, mean_len=350, min_len=50, max_len=1200, n_rate=0.03,
                        paralog_rate=0.05, shuffle_columns=9)
    names = cl[0].names
    # a table strain without a GFF leaves len(clusterpresab) != len(cluster): fine unless --consider-missing-cluster,
    # where the reference stops with numpy's IndexError (panfeed.py:19) -- checked below
    csvp, gffs, fas = synth.write_pangenome(str(tmp_path), cl, drop_gff_for=() if cm else (names[11],),
                                            separate_fasta_for=(names[1],))
    gn = sorted(gffs)
    tg = (names[0], names[33])
    strains, table = ir.load_table(csvp)
    data = ir.load_genomes(gn, [gffs[n] for n in gn], [fas[n] for n in gn])
    recs = list(ir.iter_gene_clusters(strains, table, data, up, down, dsc))
    run = po.OracleRun(klength=k, canon=canon, stroi=set(tg), consider_missing=cm)
    run.feed(recs)
    ek, ekh, ehp = run.texts()
    eng = Engine(klength=k, canon=canon, max_strains=96, stroi=set(tg), consider_missing=cm)
    with ni.Pangenome(csvp, None, None, up, down, dsc, targets=tg, genome_names=gn, gff_paths=[gffs[n] for n in gn],
                      fasta_paths=[fas[n] for n in gn]) as pg:
        if resident:
            pg.make_resident(eng)
            assert pg.n_contigs >= 69 and pg.genome_bases > 100000
        outs = list(eng.run_pangenome(pg, batch_clusters=9))
    assert len(outs) == 5
    assert "".join(o.kmers_to_hashes for o in outs) == ekh
    assert "".join(o.hashes_to_patterns for o in outs) == ehp
    assert "".join(o.kmers_tsv for o in outs) == ek
    eng.close()


def test_missing_gff_under_consider_missing_raises(tmp_path):
    from oracle import input_restatement as ir
    from oracle import oracle as po
    from panfeed_amd import native_input as ni
    from panfeed_amd import synth
    from panfeed_amd.engine import Engine
    cl = synth.generate(3, 20, first=5, flank=0, mean_len=120, min_len=50, max_len=300)
    names = cl[0].names
    csvp, gffs, fas = synth.write_pangenome(str(tmp_path), cl, drop_gff_for=(names[4],))
    gn = sorted(gffs)
    strains, table = ir.load_table(csvp)
    recs = list(ir.iter_gene_clusters(strains, table, ir.load_genomes(gn, [gffs[n] for n in gn]), 0, 0, False))
    with pytest.raises(IndexError, match="boolean index did not match"):
        po.OracleRun(klength=11, consider_missing=True).feed(recs)
    eng = Engine(klength=11, max_strains=32, consider_missing=True)
    with ni.Pangenome(csvp, None, genome_names=gn, gff_paths=[gffs[n] for n in gn]) as pg:
        with pytest.raises(IndexError, match="boolean index did not match"):
            list(eng.run_pangenome(pg))
    eng.close()


def test_gather_argument_checks(tmp_path):
    """pf_submit_gather refuses ranges outside the resident genomes / literal buffer instead of reading them"""
    from panfeed_amd import native_input as ni
    from panfeed_amd import synth
    from panfeed_amd._lib import PanfeedHipError
    from panfeed_amd.engine import Engine
    cl = synth.generate(4, 12, first=1, flank=0, mean_len=150, min_len=60, max_len=300, n_rate=0.0)
    csvp, gffs, fas = synth.write_pangenome(str(tmp_path), cl, missing_gene_rate=0.0, lower_rate=0.0)
    gn = sorted(gffs)
    eng = Engine(klength=15, max_strains=32)
    with ni.Pangenome(csvp, None, genome_names=gn, gff_paths=[gffs[n] for n in gn]) as pg:
        pg.make_resident(eng)
        hb = next(pg.batches(15, True, eng.W))
    assert hb.gather_src_off is not None and (hb.gather_src_flags & 1).sum() == 0      # everything by reference
    eng.submit_host_batch(hb)                                                            # fine
    bad = hb.gather_src_off.copy()
    hb.gather_src_off = bad + np.uint64(1 << 40)
    with pytest.raises(PanfeedHipError, match="outside the resident genomes"):
        eng.submit_host_batch(hb)
    hb.gather_src_off = bad
    fl = hb.gather_src_flags.copy()
    hb.gather_src_flags = fl | np.uint32(1)                                              # claims a literal source
    with pytest.raises(PanfeedHipError, match="literal source"):
        eng.submit_host_batch(hb)
    hb.gather_src_flags = fl
    assert eng.L.pf_genomes_clear(eng.ctx) == 0
    with pytest.raises(PanfeedHipError, match="outside the resident genomes"):
        eng.submit_host_batch(hb)
    eng.close()


@pytest.mark.parametrize("resident,device_text", [(True, True), (False, True), (True, False)],
                         ids=["resident_devtext", "text_devtext", "resident_hosttext"])
@pytest.mark.parametrize("compress", [False, True], ids=["plain", "gzip"])
def test_run_files_pipeline(tmp_path, compress, resident, device_text):
    """row N3: one call from the files on disk to the three output files (writer thread, parallel gzip), headers
    included, against the oracle over the restated records"""
    import gzip
    import os
    from oracle import input_restatement as ir
    from oracle import oracle as po
    from panfeed_amd import synth
    from panfeed_amd.engine import KMERS_TO_HASHES_HEADER, KMERS_TSV_HEADER, hashes_to_patterns_header
    from panfeed_amd.pipeline import run_files
    cl = synth.generate(30, 50, first=300, flank=0, mean_len=300, min_len=50, max_len=900, n_rate=0.03, paralog_rate=0.05)
    names = cl[0].names
    src = tmp_path / "in"
    csvp, gffs, fas = synth.write_pangenome(str(src), cl)
    gn = sorted(gffs)
    tg = (names[3],)
    out = str(tmp_path / "out")
    tg = tg if compress else ()         # with and without kmers.tsv rows (batches with them use the host renderers)
    st = run_files(csvp, str(src / "gffs"), out, klength=23, upstream=40, downstream=20, targets=tg, compress=compress,
                   batch_clusters=7, resident=resident, device_text=device_text)
    strains, table = ir.load_table(csvp)
    recs = list(ir.iter_gene_clusters(strains, table, ir.load_genomes(gn, [gffs[n] for n in gn]), 40, 20, False))
    run = po.OracleRun(klength=23, stroi=set(tg))
    run.feed(recs)
    ek, ekh, ehp = run.texts()
    assert st["clusters"] == 30 and st["instances"] == run.stats()["instances"]

    def rd(name):
        if compress:
            with gzip.open(os.path.join(out, name + ".gz"), "rt") as fh:
                return fh.read()
        with open(os.path.join(out, name)) as fh:
            return fh.read()
    assert rd("kmers_to_hashes.tsv") == KMERS_TO_HASHES_HEADER + ekh
    assert rd("hashes_to_patterns.tsv") == hashes_to_patterns_header(strains) + ehp
    assert rd("kmers.tsv") == KMERS_TSV_HEADER + ek


def test_run_files_start_ups_write_the_same_files(tmp_path):
    """run_files' three start-ups -- one pass (the genomes go to the GPU as their files are read,
    pf_pangenome_open_device), two steps overlapped (context made while the reader opens, genomes uploaded while the packer
    starts: the store's layout comes from the contig lengths alone), two steps in series -- write the same three files"""
    import os
    from panfeed_amd import synth
    from panfeed_amd.pipeline import run_files
    cl = synth.generate(40, 30, first=900, flank=10, mean_len=250, min_len=50, max_len=700, n_rate=0.03, paralog_rate=0.05)
    src = tmp_path / "in"
    csvp, _gffs, _fas = synth.write_pangenome(str(src), cl, workers=2)
    texts = {}
    for one_pass, overlap in ((True, True), (False, True), (False, False)):
        out = str(tmp_path / f"out_{one_pass}_{overlap}")
        st = run_files(csvp, str(src / "gffs"), out, klength=21, upstream=10, downstream=10, batch_clusters=9, overlap=overlap,
                       one_pass=one_pass)
        assert st["clusters"] == 40 and st["stages"]["one_pass_ingest"] == one_pass
        if overlap and not one_pass:
            assert "first_submit_wait_s" in st["stages"] and st["stages"]["genome_upload_s"] > 0
        texts[(one_pass, overlap)] = {f: open(os.path.join(out, f)).read() for f in sorted(os.listdir(out))}
    assert texts[(True, True)] == texts[(False, True)] == texts[(False, False)] and len(texts[(True, True)]) == 3


@pytest.mark.parametrize("style", ["crlf", "ragged", "lower", "oneline", "trailing"])
def test_one_pass_ingest_of_awkward_fasta(tmp_path, style):
    """genome_pack_text_kernel on FASTA that is not wrapped the usual way (CRLF, lines of any length with blank ones
    between, lower case, one line without a final newline, blank lines and a repeated record at the end), a target strain
    (its text stays on the host), 'N's, flanks over contig ends: files -> three files against the oracle over the restated
    records of the same files, and against the two-step start-up"""
    import os
    from oracle import input_restatement as ir
    from oracle import oracle as po
    from panfeed_amd import synth
    from panfeed_amd.engine import KMERS_TO_HASHES_HEADER, KMERS_TSV_HEADER, hashes_to_patterns_header
    from panfeed_amd.pipeline import run_files
    from test_native_input import _mangle_fasta
    rng = np.random.default_rng(9)
    cl = synth.generate(24, 20, first=40, flank=0, mean_len=280, min_len=50, max_len=800, n_rate=0.05, paralog_rate=0.08)
    names = cl[0].names
    src = tmp_path / "in"
    csvp, gffs, fas = synth.write_pangenome(str(src), cl, separate_fasta_for=(names[2],), lower_rate=0.0)
    gn = sorted(gffs)
    for n in gn:
        _mangle_fasta(fas[n] or gffs[n], rng, style)
    tg = (names[4],)
    outs = {}
    for one_pass in (True, False):
        out = str(tmp_path / f"out_{one_pass}")
        st = run_files(csvp, str(src / "gffs"), out, fastadir=str(src / "gffs"), klength=25, upstream=300, downstream=120,
                       targets=tg, batch_clusters=5, one_pass=one_pass)
        assert st["stages"]["one_pass_ingest"] == one_pass
        outs[one_pass] = {f: open(os.path.join(out, f)).read() for f in sorted(os.listdir(out))}
    assert outs[True] == outs[False]
    strains, table = ir.load_table(csvp)
    data = ir.load_genomes(gn, [gffs[n] for n in gn], [fas[n] for n in gn])
    run = po.OracleRun(klength=25, stroi=set(tg))
    run.feed(list(ir.iter_gene_clusters(strains, table, data, 300, 120, False)))
    ek, ekh, ehp = run.texts()
    assert outs[True]["kmers_to_hashes.tsv"] == KMERS_TO_HASHES_HEADER + ekh
    assert outs[True]["hashes_to_patterns.tsv"] == hashes_to_patterns_header(strains) + ehp
    assert outs[True]["kmers.tsv"] == KMERS_TSV_HEADER + ek


def test_run_files_multiple_files(tmp_path):
    """--multiple-files through the pipeline: one directory per cluster, patterns start empty in each"""
    import os
    from oracle import input_restatement as ir
    from oracle import oracle as po
    from panfeed_amd import synth
    from panfeed_amd.engine import KMERS_TO_HASHES_HEADER, KMERS_TSV_HEADER, hashes_to_patterns_header
    from panfeed_amd.pipeline import run_files
    cl = synth.generate(9, 30, first=900, flank=0, mean_len=200, min_len=50, max_len=500, n_rate=0.03, paralog_rate=0.05)
    names = cl[0].names
    src = tmp_path / "in"
    csvp, gffs, fas = synth.write_pangenome(str(src), cl)
    gn = sorted(gffs)
    tg = (names[5],)
    out = str(tmp_path / "out")
    run_files(csvp, str(src / "gffs"), out, klength=19, targets=tg, multiple_files=True, batch_clusters=4)
    strains, table = ir.load_table(csvp)
    recs = list(ir.iter_gene_clusters(strains, table, ir.load_genomes(gn, [gffs[n] for n in gn]), 0, 0, False))
    assert sorted(os.listdir(out)) == sorted(r[1] for r in recs)
    for rec in recs:
        run = po.OracleRun(klength=19, stroi=set(tg), multiple_files=True)
        run.feed([rec])
        ek, ekh, ehp = run.texts()
        d = os.path.join(out, rec[1])
        assert open(os.path.join(d, "kmers_to_hashes.tsv")).read() == KMERS_TO_HASHES_HEADER + ekh
        assert open(os.path.join(d, "hashes_to_patterns.tsv")).read() == hashes_to_patterns_header(strains) + ehp
        assert open(os.path.join(d, "kmers.tsv")).read() == KMERS_TSV_HEADER + ek


def test_run_files_no_cluster_selected(tmp_path):
    """--genes naming no cluster of the table: the three files exist and hold their headers only"""
    import os
    from panfeed_amd import synth
    from panfeed_amd.engine import KMERS_TO_HASHES_HEADER, KMERS_TSV_HEADER, hashes_to_patterns_header
    from panfeed_amd.pipeline import run_files
    cl = synth.generate(3, 8, first=1, flank=0, mean_len=100, min_len=40, max_len=200)
    src = tmp_path / "in"
    csvp, gffs, fas = synth.write_pangenome(str(src), cl)
    out = str(tmp_path / "out")
    st = run_files(csvp, str(src / "gffs"), out, klength=11, genes=["not_in_the_table"])
    assert st["clusters"] == 0 and st["bytes"] == 0
    assert open(os.path.join(out, "kmers.tsv")).read() == KMERS_TSV_HEADER
    assert open(os.path.join(out, "kmers_to_hashes.tsv")).read() == KMERS_TO_HASHES_HEADER
    assert open(os.path.join(out, "hashes_to_patterns.tsv")).read() == hashes_to_patterns_header(cl[0].names)


def test_run_files_refuses_an_existing_output_directory(tmp_path):
    """input.py:213-216: the reference stops when the output directory exists"""
    from panfeed_amd import synth
    from panfeed_amd.pipeline import run_files
    cl = synth.generate(2, 6, first=1, flank=0, mean_len=100, min_len=40, max_len=200)
    csvp, gffs, fas = synth.write_pangenome(str(tmp_path / "in"), cl)
    out = tmp_path / "out"
    out.mkdir()
    with pytest.raises(FileExistsError):
        run_files(csvp, str(tmp_path / "in" / "gffs"), str(out), klength=11)


@pytest.mark.parametrize("world", [2, 3])
def test_run_files_sharded_equals_single_process(tmp_path, world):
    """files on disk -> files on disk over `world` ranks (each opens the pangenome, takes its weight-balanced range,
    keeps its genomes resident, writes its parts): the same bytes as the one-process pipeline"""
    import os
    from panfeed_amd import synth
    from panfeed_amd.pipeline import run_files
    from test_sharded_cpu import FILES, read_out, run_world
    cl = synth.generate(23, 40, first=500, flank=0, mean_len=300, min_len=50, max_len=900, n_rate=0.03, paralog_rate=0.05)
    names = cl[0].names
    src = tmp_path / "in"
    csvp, gffs, fas = synth.write_pangenome(str(src), cl)
    single = str(tmp_path / "single")
    run_files(csvp, str(src / "gffs"), single, klength=23, upstream=30, downstream=10, targets=(names[3], names[20]),
              batch_clusters=5)
    out = str(tmp_path / "sharded")
    stats = run_world("gpu", world, out, f"files:{csvp}:{src / 'gffs'}:23:30:10:{names[3]},{names[20]}")
    for f in FILES:
        assert read_out(out, f, False) == read_out(single, f, False), f
    assert sum(s["clusters"] for s in stats.values()) == 23
    assert sorted(os.listdir(out)) == sorted(FILES)
