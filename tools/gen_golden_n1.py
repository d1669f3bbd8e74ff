#!/usr/bin/env python3
"""Generate tests/golden/n1.json.gz: the REFERENCE's own reader code (row N1 of SURVEY 8f) run in this container.

What runs for real, imported from /root/reference/panfeed/input.py: `what_are_my_inputfiles` (:16-64), `set_input_output`
(:180-232: the `pd.read_csv(...).drop(...)` table load, targets / genes files), `prep_data_n_fasta` (:67-138: the
`##FASTA` split, `create_faidx`, `parse_gff`), `parse_gff` (:274-332) and `iter_gene_clusters` (:335-468: table walk,
`sortstrain`, `clusterpresab`, paralog split, the four offset-clipping branches :413-446, `seq_start` / `seq_end`,
absent strains, `gene_list`, `raise_missing`, the warnings).

What does NOT: pyfaidx (third party, unpinned in pyproject.toml:26-32, neither under /root/reference nor installed --
SURVEY 8c).  `input.Fasta` is bound to the DECLARED DOUBLE below.  The reference touches a `Fasta` through exactly
these operations -- `Fasta(path, sequence_always_upper=True, rebuild=False)`, `fa[contig]` (KeyError when absent),
`record[a:b]`, unary `-` on the slice, `[::-1]` on that, `str()`, `.close()` -- and the double states, in ~40 lines,
what each is ASSUMED to do (pyfaidx's documented behaviour):
    * a record's key is its header line up to the first whitespace; its sequence is the lines joined, upper-cased;
    * `record[a:b]` is Python-slice semantics on that string (clipped at the contig's ends);
    * `-seq` is the reverse complement under ACTGNactgnYRWSKMDVHBXyrwskmdvhbx -> TGACNtgacnRYWSMKHBDVXryswmkhbdvx,
      other characters unchanged; `seq[::-1]` the reversal; `str(seq)` the letters.
That is the whole unpinned surface of row N1: five operations, not the 195 lines around them.  (A repeated contig name,
ragged line lengths and an empty header are errors in pyfaidx and are not generated.)

The fixture holds inputs we made (texts of the table / GFF3 / FASTA files, options) and what the reference yielded
(records, parsed features, warnings, the exception a `raise_missing` run ends with) -- data, no reference text.

Usage: python tools/gen_golden_n1.py            (rewrites tests/golden/n1.json.gz; build container only)
"""
import gzip
import json
import logging
import os
import shutil
import sys
import tempfile
import types

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

_COMP = str.maketrans("ACTGNactgnYRWSKMDVHBXyrwskmdvhbx", "TGACNtgacnRYWSMKHBDVXryswmkhbdvx")


# ------------------------------------------------------------------ the declared pyfaidx double (the one assumption)
class _Seq:
    def __init__(self, letters):
        self.seq = letters

    def __getitem__(self, sl):                     # only `[::-1]` is used on a slice result (input.py:452)
        return _Seq(self.seq[sl])

    def __neg__(self):                             # input.py:431, 443, 449
        return _Seq(self.seq[::-1].translate(_COMP))

    def __str__(self):                             # input.py:455
        return self.seq


class _Record:
    def __init__(self, letters):
        self.letters = letters

    def __getitem__(self, sl):                     # input.py:427, 431, 439, 443
        return _Seq(self.letters[sl])


class FastaDouble:
    def __init__(self, file_name, sequence_always_upper=False, rebuild=True):
        self.records, name = {}, None
        with open(file_name) as fh:
            for line in fh:
                line = line.rstrip("\n").rstrip("\r")
                if line.startswith(">"):
                    name = line[1:].split()[0]
                    assert name not in self.records, "repeated contig name: an error in pyfaidx, not generated"
                    self.records[name] = []
                elif name is not None:
                    self.records[name].append(line)
        up = (lambda s: s.upper()) if sequence_always_upper else (lambda s: s)
        self.records = {k: _Record(up("".join(v))) for k, v in self.records.items()}

    def __getitem__(self, name):                   # input.py:405 (KeyError -> "Could not find chromosome")
        return self.records[name]

    def close(self):                               # input.py:265, 462
        pass


_stub = types.ModuleType("pyfaidx")
_stub.Fasta = FastaDouble
sys.modules["pyfaidx"] = _stub
sys.path.insert(0, "/root/reference")
from panfeed import input as ref_input  # noqa: E402  (reference)

assert ref_input.Fasta is FastaDouble

from panfeed_amd import synth  # noqa: E402


class _Capture(logging.Handler):
    def __init__(self):
        super().__init__(level=logging.WARNING)
        self.lines = []

    def emit(self, record):
        self.lines.append(record.getMessage())


def materialise(files, root):
    for rel, text in files.items():
        p = os.path.join(root, rel)
        os.makedirs(os.path.dirname(p), exist_ok=True)
        with open(p, "w", newline="") as fh:
            fh.write(text)


def run_reference(files, o):
    """the reference's own call sequence, __main__.py:245-275"""
    root = tempfile.mkdtemp(prefix="golden_n1_")
    cap = _Capture()
    lg = logging.getLogger("panfeed.input")
    lg.addHandler(cap)
    lg.setLevel(logging.WARNING)
    try:
        materialise(files, root)
        gffdir = os.path.join(root, "gffs")
        fastadir = gffdir if o["fasta_dir"] else None
        filelist, fastalist = ref_input.what_are_my_inputfiles(gffdir, fastadir)
        targets_in = genes_in = None
        if o["targets"] is not None:
            targets_in = os.path.join(root, "targets.txt")
            with open(targets_in, "w") as fh:
                fh.write("".join(t + "\n" for t in o["targets"]))
        if o["genes"] is not None:
            genes_in = os.path.join(root, "genes.txt")
            with open(genes_in, "w") as fh:
                fh.write("".join(g + "\n" for g in o["genes"]))
        out = os.path.join(root, "out")
        stroi, genes, _ks, _hp, _kh, genepres = ref_input.set_input_output(
            targets_in, genes_in, os.path.join(root, "gene_presence_absence.csv"), out, single_file=False)
        data = ref_input.prep_data_n_fasta(filelist, fastalist, gffdir, fastadir, out)
        exp = {"filelist": list(filelist), "fastalist": list(fastalist), "strains": [str(c) for c in genepres.columns],
               "stroi": sorted(stroi) if stroi != "" else None, "genes": sorted(genes) if genes is not None else None,
               "features": {g: {k: [f.id, f.chromosome, f.start, f.end, f.strand] for k, f in data[g][1].items()}
                            for g in filelist},
               "records": [], "raises": None}
        try:
            for gs, idx, presab in ref_input.iter_gene_clusters(genepres, data, o["up"], o["down"], o["dsc"], True,
                                                                gene_list=genes, raise_missing=o["raise_missing"]):
                assert presab.dtype == np.dtype(int)
                exp["records"].append({"idx": idx, "presab": [int(x) for x in presab],
                                       "strains": [[nm, [[s.sequence, s.compsequence, s.id, s.chromosome, int(s.start),
                                                          int(s.end), int(s.strand), int(s.offset)] for s in seqs]]
                                                   for nm, seqs in gs.items()]})
        except KeyError as e:                       # input.py:345, 399, 409
            exp["raises"] = e.args[0]
        exp["warnings"] = [w.replace(root, "{DIR}") for w in cap.lines]
        return exp
    finally:
        lg.removeHandler(cap)
        shutil.rmtree(root, ignore_errors=True)


def read_tree(root):
    files = {}
    for d, _sub, names in os.walk(root):
        for n in sorted(names):
            p = os.path.join(d, n)
            with open(p, newline="") as fh:
                files[os.path.relpath(p, root)] = fh.read()
    return files


def synth_pangenome(seed, n_clusters, n_samples, **kw):
    wp = {k: kw.pop(k) for k in ("wrap", "drop", "sep", "missing_gene_rate", "lower_rate") if k in kw}
    cl = synth.generate(n_clusters, n_samples, first=seed * 100, **kw)
    names = cl[0].names
    root = tempfile.mkdtemp(prefix="golden_n1_src_")
    try:
        synth.write_pangenome(root, cl, seed=seed, wrap=wp.get("wrap", 60),
                              drop_gff_for=tuple(names[i] for i in wp.get("drop", ())),
                              separate_fasta_for=tuple(names[i] for i in wp.get("sep", ())),
                              missing_gene_rate=wp.get("missing_gene_rate", 0.05), lower_rate=wp.get("lower_rate", 0.05))
        return read_tree(root), names
    finally:
        shutil.rmtree(root, ignore_errors=True)


def handmade_pangenome():
    """what synth does not draw: every branch of parse_gff (:286-329) and of the offset clipping (:413-446) on purpose"""
    c1 = "ACGTTGCAAGGCTTAACCGGATCGATTACGGCTAGCTAGGATCCGATCGTTAGCAAGCTTGGCCAATGCATGCAAGT"      # 76 bases
    c2 = "ttgacagctagctcagtcctaggtataatgctagcNNacgtRYKMacgtacgtaggctagctaacgcgatatcgcg"       # lower case, N, IUPAC
    a = "\n".join([
        "##gff-version 3",
        "  # a comment after blanks",
        "#!processor x",
        "",                                                        # split -> ['\n']: IndexError -> warning
        "a_c1\tsrc\tCDS\t1\t30\t.\t+\t0\tID=a_g1;product=flush at the contig's first base",
        "a_c1\tsrc\tCDS\t47\t76\t.\t-\t0\tID=a_g2;Name=flush at the contig's last base",
        "a_c1\tsrc\tCDS\t5\t40\t.\t-\t0\tID=a_g3;x=1",
        "a_c1\tsrc\tCDS\t6\t41\t.\t-\t0\tID=a_g11",                # the line's last field keeps its newline: ID 'a_g11\n'
        "a_c1\tsrc\tgene\t5\t40\t.\t-\t0\tID=a_gene_not_cds",
        "a_c1\tsrc\tCDS\t10\t20\t.\t.\t0\tID=a_g4;note=strand '.' reads as -1",
        "a_c1\tsrc\tCDS\tabc\t20\t.\t+\t0\tID=a_bad_int",
        "a_c1\tsrc\tCDS\t 12 \t+25\t.\t+\t0\tID=a_g5;note=int() takes blanks and a sign",
        "a_c1\tsrc\tCDS\t1_0\t2_5\t.\t+\t0\tID=a_g6;note=int() takes underscores",
        "a_c1\tsrc\tCDS\t3\t9",
        # short rows: the FIRST failing expression names the warning (int(entries[3]), int(entries[4]) come before
        # entries[6] and entries[8]); a row's last field keeps its newline, in the repr too
        "a_c1\tsrc\tCDS\txyz\t9",
        "a_c1\tsrc\tCDS\t3\tqq",
        "a_c1\tsrc\tCDS\t3",
        "a_c1\tsrc\tCDS\t3\t9\t.\t+",
        "a_c1\tsrc\tCDS\t3\t9\t.\t+\t0",
        "a_c1\tsrc\tCDS\t3\t9\t.\t+\t0\tName=no id at all",
        "a_c1\tsrc\tCDS\t3\t19\t.\t+\t0\tParent=x;ID=a_g7=tail;IDx=a_g7b;y=2",
        "a_c1\tsrc\tCDS\t2\t18\t.\t+\t0\tIDENTITY;ID=a_g8;",
        "a_c9\tsrc\tCDS\t2\t18\t.\t+\t0\tID=a_g9;note=contig absent from the FASTA",
        "a_c2\tsrc\tCDS\t20\t50\t.\t+\t0\tID=a_g1;note=a repeated ID: the later line wins",
        "a_c2\tsrc\tCDS\t30\t60\t.\t-\t0\tID=a_g10\textra\tcolumns",
        " \t ##FASTA is only a marker after lstrip",
        "a_c1\tsrc\tCDS\t1\t5\t.\t+\t0\tID=a_after_marker",
        ">a_c1 first contig",
    ] + [c1[i:i + 20] for i in range(0, len(c1), 20)] + [">a_c2\tsecond"] + [c2[i:i + 20] for i in range(0, len(c2), 20)]) + "\n"
    b = "\n".join([
        "##gff-version 3",
        "b_c1\tsrc\tCDS\t4\t33\t.\t+\t0\tID=b_g1;x=1",
        "b_c1\tsrc\tCDS\t40\t70\t.\t-\t0\tID=b_g2;x=1",
        "b_c1\tsrc\tCDS\t76\t76\t.\t+\t0\tID=b_g3;note=one base, the contig's last",
        "b_c1\tsrc\tCDS\t70\t90\t.\t+\t0\tID=b_g4;note=runs past the contig's end",
        "##FASTA",
        ">b_c1",
        c1[::-1],
    ]) + "\n"
    cgff = "##gff-version 3\nc_c1\tsrc\tCDS\t2\t31\t.\t-\t0\tID=c_g1;x=1\nc_c1\tsrc\tCDS\t35\t64\t.\t+\t0\tID=c_g2;x=1\n"
    cfna = ">c_c1 separate nucleotide file\n" + c2.upper().replace("N", "A") + "\n"
    table = "\n".join([
        "Gene,Non-unique Gene name,Annotation,b,a,d,c",
        'grp_edges,,"x, y",b_g1;b_g2,a_g1;a_g2;a_g3;a_g11,,c_g1',
        "grp_parse,,,b_g3;b_g4,a_g4;a_g5;a_g6;a_g7;a_g8,d_g1,c_g2",
        "grp_missing,,,NA,a_g9;a_bad_int;a_g10;a_after_marker;a_gene_not_cds,,",
        'grp_quoted,,,"b_g1","a_g2",NaN,"c_g1;c_g2"',
        "grp_none,,,,,,",
    ]) + "\n"
    return {"gene_presence_absence.csv": table, "gffs/a.gff": a, "gffs/b.gff": b, "gffs/c.gff": cgff,
            "gffs/c.fna": cfna, "gffs/readme.txt": "not a gff\n"}


def main():
    pangenomes, cases = {}, []

    def opts(up=0, down=0, dsc=False, genes=None, targets=None, raise_missing=False, fasta_dir=True):
        return dict(up=up, down=down, dsc=dsc, genes=genes, targets=targets, raise_missing=raise_missing, fasta_dir=fasta_dir)

    def add(name, pg, **o):
        case = {"name": name, "pangenome": pg, "opts": opts(**o)}
        case["expect"] = run_reference(pangenomes[pg], case["opts"])
        cases.append(case)

    pangenomes["hand"] = handmade_pangenome()
    for nm, o in [("hand_0_0", {}), ("hand_3_4", dict(up=3, down=4)), ("hand_25_10", dict(up=25, down=10)),
                  ("hand_10_25_dsc", dict(up=10, down=25, dsc=True)), ("hand_1000_1000", dict(up=1000, down=1000)),
                  ("hand_1000_1000_dsc", dict(up=1000, down=1000, dsc=True)), ("hand_0_7_dsc", dict(down=7, dsc=True)),
                  ("hand_genes", dict(up=2, down=2, genes=["grp_parse", "grp_none", "absent_cluster"], targets=["a", "zz"])),
                  ("hand_raise", dict(raise_missing=True))]:
        add(nm, "hand", **o)
    pangenomes["hand_all_gffs"] = {k: v for k, v in handmade_pangenome().items()}
    pangenomes["hand_all_gffs"]["gffs/d.gff"] = "##gff-version 3\nd_c1\tsrc\tCDS\t3\t12\t.\t+\t0\tID=d_g1\n##FASTA\n>d_c1\nACGTACGTACGTACGTAC\n"
    add("hand_all_gffs_raise_gene", "hand_all_gffs", raise_missing=True)
    add("hand_all_gffs_5_5", "hand_all_gffs", up=5, down=5)

    files, names = synth_pangenome(3, 14, 10, flank=0, mean_len=160, min_len=25, max_len=500, n_rate=0.08, paralog_rate=0.12,
                                   shuffle_columns=5, drop=(2,), sep=(4, 5), wrap=60)
    pangenomes["synth_a"] = files
    for nm, o in [("synth_a_0_0", {}), ("synth_a_50_30", dict(up=50, down=30)), ("synth_a_200_200", dict(up=200, down=200)),
                  ("synth_a_20_40_dsc", dict(up=20, down=40, dsc=True)), ("synth_a_0_10_dsc", dict(down=10, dsc=True)),
                  ("synth_a_5000_5000", dict(up=5000, down=5000)), ("synth_a_raise", dict(raise_missing=True)),
                  ("synth_a_embedded_fasta_only", dict(up=7, down=9, fasta_dir=False))]:
        if nm == "synth_a_embedded_fasta_only":
            # without a fasta directory every GFF needs its own ##FASTA section: leave out the two that have none
            sub = {k: v for k, v in files.items() if not any(k.startswith(f"gffs/{names[i]}.") for i in (4, 5))}
            pangenomes["synth_a_embedded"] = sub
            add(nm, "synth_a_embedded", **o)
        else:
            add(nm, "synth_a", **o)
    table = [ln.split(",")[0] for ln in files["gene_presence_absence.csv"].split("\n")[1:] if ln]
    add("synth_a_genes", "synth_a", up=10, down=10, genes=[table[3], table[9], "not_a_cluster"], targets=[names[0], names[7]])

    files, names = synth_pangenome(4, 9, 17, flank=15, mean_len=90, min_len=3, max_len=300, n_rate=0.3, paralog_rate=0.25,
                                   wrap=10, missing_gene_rate=0.1, lower_rate=0.3)
    pangenomes["synth_b"] = files
    for nm, o in [("synth_b_15_15", dict(up=15, down=15)), ("synth_b_7_150", dict(up=7, down=150)),
                  ("synth_b_150_3_dsc", dict(up=150, down=3, dsc=True))]:
        add(nm, "synth_b", **o)

    out = os.path.join(REPO, "tests", "golden", "n1.json.gz")
    payload = json.dumps({"pangenomes": pangenomes, "cases": cases}, sort_keys=True, separators=(",", ":")).encode()
    with open(out, "wb") as raw:
        with gzip.GzipFile(fileobj=raw, mode="wb", mtime=0, filename="") as fh:
            fh.write(payload)
    nrec = sum(len(c["expect"]["records"]) for c in cases)
    nseq = sum(len(s[1]) for c in cases for r in c["expect"]["records"] for s in r["strains"])
    nwarn = sum(len(c["expect"]["warnings"]) for c in cases)
    print(f"{out}: {len(cases)} cases over {len(pangenomes)} pangenomes, {nrec} records, {nseq} sequences, "
          f"{nwarn} warnings, {os.path.getsize(out)} bytes")


if __name__ == "__main__":
    main()
