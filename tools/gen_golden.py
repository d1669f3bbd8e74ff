#!/usr/bin/env python3
"""Generate tests/golden/*.json.gz by running the REFERENCE hot path in this container.

Runs only where /root/reference exists (the build container); the fixtures it writes are
plain data (input strings + flags, expected TSV texts) and are what travels.

The reference's `panfeed/panfeed.py` imports `.input`, which imports `pyfaidx` at module
top (input.py:9); pyfaidx is not installed here and the hot-path functions never touch it
(they consume plain-str Seqinfo records), so an empty module named `pyfaidx` is registered
before the import -- the procedure SURVEY.md section 8(c) records.  Nothing of the reference is
copied: only its outputs on our inputs are stored.

Usage: python tools/gen_golden.py            (rewrites tests/golden/)
"""
import gzip
import io
import json
import os
import shutil
import sys
import tempfile
import types

import numpy as np
import pandas as pd

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

_stub = types.ModuleType("pyfaidx")
_stub.Fasta = object
sys.modules.setdefault("pyfaidx", _stub)
sys.path.insert(0, "/root/reference")
from panfeed.panfeed import cluster_cutter, pattern_hasher, write_headers  # noqa: E402  (reference)
from panfeed.classes import Seqinfo as RefSeqinfo  # noqa: E402

from panfeed_amd import synth  # noqa: E402

_COMP = str.maketrans("ACGTN", "TGCAN")


def si(seq, id="g", chrom="c1", start=1, strand=1, offset=0, comp=None):
    comp = seq.translate(_COMP) if comp is None else comp
    return (seq, comp, id, chrom, start, start + len(seq) - 1, strand, offset)


def cluster_to_json(rec):
    gs, idx, presab = rec
    return {"idx": idx, "presab": [int(x) for x in presab],
            "strains": [[name, [list(s) for s in seqs]] for name, seqs in gs.items()]}


def json_to_ref_record(cj):
    gs = {name: [RefSeqinfo(*s) for s in seqs] for name, seqs in cj["strains"]}
    return gs, cj["idx"], np.array(cj["presab"], dtype=int)


def run_reference(case):
    o = case["opts"]
    stroi = set(o["stroi"]) if o["stroi"] is not None else ""   # input.py:194-202
    clusters = [json_to_ref_record(c) for c in case["clusters"]]
    all_names = case["all_strains"]
    genepres = pd.DataFrame(columns=all_names)
    if not o["multiple_files"]:
        ks, hp, kh = io.StringIO(), io.StringIO(), io.StringIO()
        ks.write("cluster\tstrain\tfeature_id\tcontig\tfeature_strand\tcontig_start\tcontig_end\t"
                 "gene_start\tgene_end\tstrand\tk-mer\n")     # input.py:243 (create_kmer_stroi)
        write_headers(hp, kh, genepres)
        patterns = set()
        for x in clusters:                                     # __main__.py:350-356
            ret = cluster_cutter(x, o["klength"], stroi, False, o["canon"], o["consider_missing"], "unused")
            patterns = pattern_hasher((ret,), ks, hp, kh, genepres, o["patfilt"], o["maf"], "unused",
                                      patterns=patterns, consider_missing_cluster=o["consider_missing"])
        return {"kmers.tsv": ks.getvalue(), "kmers_to_hashes.tsv": kh.getvalue(),
                "hashes_to_patterns.tsv": hp.getvalue(), "n_patterns": len(patterns)}
    out = tempfile.mkdtemp(prefix="golden_mf_")
    try:
        patterns = set()
        for x in clusters:
            ret = cluster_cutter(x, o["klength"], stroi, True, o["canon"], o["consider_missing"], out)
            patterns = pattern_hasher((ret,), None, None, None, genepres, o["patfilt"], o["maf"], out,
                                      patterns=patterns, consider_missing_cluster=o["consider_missing"])
        res = {}
        for d in sorted(os.listdir(out)):
            res[d] = {}
            for f in sorted(os.listdir(os.path.join(out, d))):
                # files of the last cluster may be unflushed handles; the reference leaves them to GC
                with open(os.path.join(out, d, f)) as fh:
                    res[d][f] = fh.read()
        return {"dirs": res}
    finally:
        import gc
        gc.collect()
        shutil.rmtree(out, ignore_errors=True)


def opts(klength=31, stroi=None, canon=True, consider_missing=False, patfilt=True, maf=0.01,
         multiple_files=False):
    return dict(klength=klength, stroi=stroi, canon=canon, consider_missing=consider_missing,
                patfilt=patfilt, maf=maf, multiple_files=multiple_files)


def make_case(name, records, all_strains, **o):
    case = {"name": name, "opts": opts(**o), "all_strains": list(all_strains),
            "clusters": [cluster_to_json(r) for r in records]}
    case["expect"] = run_reference(case)
    return case


def handmade():
    cases = []
    # the toy of SURVEY 8(c): k=5, one - strand target strain, one absent strain
    toy = ({"s2": [si("ACGTACGTTG", "g2", "c2", 3, 1, 0)],
            "s1": [si("ACGTACGATG", "g1", "c1", 10, -1, 0)], "s3": []}, "grp1", [1, 1, 0])
    for nm, kw in [("toy_canon", dict(klength=5, stroi=["s1"])),
                   ("toy_noncanon", dict(klength=5, stroi=["s1", "s2"], canon=False)),
                   ("toy_nofilter", dict(klength=5, stroi=None, patfilt=False)),
                   ("toy_missing", dict(klength=5, stroi=["s2"], consider_missing=True)),
                   ("toy_maf0", dict(klength=5, stroi=None, maf=0.0)),
                   ("toy_maf0_nofilter", dict(klength=5, stroi=None, maf=0.0, patfilt=False)),
                   ("toy_maf05", dict(klength=5, stroi=None, maf=0.5))]:
        cases.append(make_case(nm, [toy], ["s2", "s1", "s3"], **kw))
    # edge cases: shorter than k, exactly k, empty cluster, palindromes (tie -> forward), N, paralogs, repeats
    e1 = ({"b": [si("ACG", "gb0"), si("ACGTA", "gb1", start=50, strand=-1, offset=2)],
           "a": [si("ACGTTGCAAC", "ga0", offset=1), si("AAAAAAAAAA", "ga1", start=7)],
           "d": [si("TTTTTTTTTT", "gd0"), si("ACGTNACGTAC", "gd1", strand=-1)],
           "c": []}, "edge1", [1, 1, 0, 1])
    e2 = ({"a": [], "b": [], "c": [], "d": []}, "empty", [0, 0, 0, 0])
    e3 = ({"c": [si("GATTACAGATTACA", "gc")], "a": [si("GATTACAGATTACA", "ga")],
           "b": [si("GATTACAGATTACA", "gb")], "d": [si("TGTAATCTGTAATC", "gd")]}, "allsame", [1, 1, 1, 1])
    e4 = ({"a": [si("ACGTACGT", "ga")], "b": [si("ACGAACGT", "gb")], "c": [], "d": []}, "half", [1, 1, 0, 0])
    for nm, kw in [("edge_k5", dict(klength=5, stroi=["a", "d"])),
                   ("edge_k4", dict(klength=4, stroi=["a", "b"])),
                   ("edge_k4_noncanon", dict(klength=4, stroi=["b", "d"], canon=False)),
                   ("edge_k1", dict(klength=1, stroi=None)),
                   ("edge_k5_missing_nofilter", dict(klength=5, stroi=None, consider_missing=True, patfilt=False)),
                   ("edge_k5_nofilter_maf0", dict(klength=5, stroi=None, patfilt=False, maf=0.0)),
                   ("edge_k8", dict(klength=8, stroi=["c"])),
                   ("edge_k5_mf", dict(klength=5, stroi=["a"], multiple_files=True))]:
        cases.append(make_case(nm, [e1, e2, e3, e4], ["b", "a", "d", "c"], **kw))
    # a present strain without genome data: the cluster dict is shorter than clusterpresab (input.py:384-385)
    m1 = ({"s1": [si("ACGTACGTAA", "g1")], "s3": [si("ACGTACGTCA", "g3")], "s4": []}, "short_dict", [1, 1, 1, 0])
    cases.append(make_case("missing_gff", [m1], ["s1", "s2", "s3", "s4"], klength=5, stroi=None))
    cases.append(make_case("missing_gff_nofilter", [m1], ["s1", "s2", "s3", "s4"], klength=5, stroi=None,
                           patfilt=False, maf=0.0))
    return cases


def seeded():
    cases = []

    def recs(n_clusters, n_samples, first=0, shuffle=None, **kw):
        cl = synth.generate(n_clusters, n_samples, first=first, shuffle_columns=shuffle, **kw)
        return [c.record() for c in cl], cl[0].names

    small = dict(mean_len=120, min_len=20, max_len=400, n_rate=0.02, paralog_rate=0.05)
    r, names = recs(8, 12, **small)
    tg = [names[1], names[5]]
    for nm, kw in [("rand12_basic", dict(stroi=tg)),
                   ("rand12_notargets", dict(stroi=None)),
                   ("rand12_noncanon", dict(stroi=tg, canon=False)),
                   ("rand12_nofilter", dict(stroi=tg, patfilt=False)),
                   ("rand12_maf01", dict(stroi=tg, maf=0.1)),
                   ("rand12_missing", dict(stroi=tg, consider_missing=True)),
                   ("rand12_missing_nofilter_maf0", dict(stroi=None, consider_missing=True, patfilt=False, maf=0.0)),
                   ("rand12_mf", dict(stroi=tg, multiple_files=True)),
                   ("rand12_k21", dict(stroi=tg, klength=21)),
                   ("rand12_k32", dict(stroi=tg, klength=32)),
                   ("rand12_k32_noncanon", dict(stroi=None, klength=32, canon=False)),
                   ("rand12_k33", dict(stroi=tg, klength=33)),
                   ("rand12_k51", dict(stroi=tg, klength=51)),
                   ("rand12_k51_noncanon", dict(stroi=None, klength=51, canon=False)),
                   ("rand12_k63", dict(stroi=None, klength=63)),
                   ("rand12_k64", dict(stroi=None, klength=64)),
                   ("rand12_k7", dict(stroi=None, klength=7))]:
        cases.append(make_case(nm, r, names, **kw))
    # flanks (upstream/downstream only change the input records: offset + coordinates)
    r, names = recs(6, 10, first=100, flank=30, **small)
    cases.append(make_case("rand10_flank", r, names, stroi=[names[0], names[9]]))
    # shuffled CSV column order (iteration order != sorted order) and >32 / >64 samples (multi-word rows)
    r, names = recs(6, 40, first=200, shuffle=7, **small)
    cases.append(make_case("rand40_shuffled", r, names, stroi=[names[3]]))
    cases.append(make_case("rand40_shuffled_missing", r, names, stroi=None, consider_missing=True))
    r, names = recs(5, 70, first=300, shuffle=11, **small)
    cases.append(make_case("rand70_shuffled", r, names, stroi=None))
    cases.append(make_case("rand70_maf01_nofilter", r, names, stroi=None, maf=0.1, patfilt=False))
    # MAF boundary at S=100/200 with maf 0.01 (counts 1, 2, 98, 99, 198 ...)
    r, names = recs(3, 100, first=400, mean_len=80, min_len=40, max_len=120, n_rate=0.0, paralog_rate=0.01,
                    sub_rate=0.02)
    cases.append(make_case("rand100_mafedge", r, names, stroi=None))
    cases.append(make_case("rand100_maf005", r, names, stroi=None, maf=0.05))
    return cases


def wide():
    """round 4: the reference itself on the shapes that were pinned only through the oracle before -- 130 and 260 samples
    (5 and 9 column chunks of 32), keys of three and four 63-bit words (k = 65, 95, 126), and clusters of more than 32 /
    more than 64 distinct sequences (the two-word and the wide allele-mask paths, several key partitions at the
    library's table size)"""
    cases = []

    def recs(n_clusters, n_samples, first=0, shuffle=None, **kw):
        cl = synth.generate(n_clusters, n_samples, first=first, shuffle_columns=shuffle, **kw)
        return [c.record() for c in cl], cl[0].names

    mid = dict(mean_len=260, min_len=140, max_len=500, n_rate=0.01, paralog_rate=0.03)
    r, names = recs(3, 130, first=500, shuffle=13, **mid)
    tg = [names[2], names[77]]
    for nm, kw in [("wide130_k31", dict(stroi=tg)),
                   ("wide130_k65", dict(stroi=tg, klength=65)),
                   ("wide130_k95_noncanon", dict(stroi=None, klength=95, canon=False)),
                   ("wide130_k126", dict(stroi=tg, klength=126)),
                   ("wide130_k126_missing_nofilter", dict(stroi=None, klength=126, consider_missing=True, patfilt=False, maf=0.0))]:
        cases.append(make_case(nm, r, names, **kw))
    r, names = recs(2, 260, first=600, shuffle=17, **mid)
    for nm, kw in [("wide260_k31", dict(stroi=[names[200]])),
                   ("wide260_k65_maf005", dict(stroi=None, klength=65, maf=0.05)),
                   ("wide260_k95", dict(stroi=None, klength=95)),
                   ("wide260_k126_noncanon", dict(stroi=None, klength=126, canon=False))]:
        cases.append(make_case(nm, r, names, **kw))
    # many distinct sequences per cluster: ~45 (two mask words) and ~90 (the wide class), related and SURVEY's alleles
    many = dict(mean_len=420, min_len=300, max_len=600, n_rate=0.0, paralog_rate=0.02, allele_decay=1.0)
    r, names = recs(2, 130, first=700, flank=20, mean_alleles=45.0, allele_model="tree", **many)
    cases.append(make_case("alleles45_tree_k31", r, names, stroi=[names[5]]))
    cases.append(make_case("alleles45_tree_k65", r, names, stroi=None, klength=65))
    r, names = recs(2, 260, first=800, flank=20, mean_alleles=90.0, allele_model="star", sub_rate=0.02, **many)
    cases.append(make_case("alleles90_star_k31", r, names, stroi=None))
    cases.append(make_case("alleles90_star_k31_missing", r, names, stroi=None, consider_missing=True))
    r, names = recs(2, 260, first=900, flank=20, mean_alleles=90.0, allele_model="tree", **many)
    cases.append(make_case("alleles90_tree_k31", r, names, stroi=[names[100]]))
    cases.append(make_case("alleles90_tree_k95", r, names, stroi=None, klength=95))
    return cases


def main():
    outdir = os.path.join(REPO, "tests", "golden")
    os.makedirs(outdir, exist_ok=True)
    for fname, cases in [("handmade.json.gz", handmade()), ("seeded.json.gz", seeded()), ("wide.json.gz", wide())]:
        path = os.path.join(outdir, fname)
        with gzip.GzipFile(path, "wb", mtime=0) as fh:
            fh.write(json.dumps({"generator": "tools/gen_golden.py", "cases": cases},
                                sort_keys=True).encode())
        print(f"{path}: {len(cases)} cases, {os.path.getsize(path)} bytes")
    # known-answer vectors for the hash image (panfeed.py:175-176, 206-207)
    import binascii
    import hashlib
    kats = []
    for dtype, vals in [("int64", [1, 0]), ("float64", [1.0, 0.0]), ("float64", [1.0, float("nan")]),
                        ("float64", [0.0] * 9), ("int64", [0] * 9), ("float64", [1.0] * 130), ("int64", [])]:
        a = np.array(vals, dtype=dtype)
        h = binascii.b2a_base64(hashlib.md5(a.view(np.uint8)).digest()).decode()[:24]
        kats.append({"dtype": dtype, "hex": a.tobytes().hex(), "b64": h})
    with open(os.path.join(outdir, "hash_kat.json"), "w") as fh:
        json.dump(kats, fh, indent=1)


if __name__ == "__main__":
    main()
