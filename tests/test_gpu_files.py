"""File surface through the mirror API with real files: names, headers, gzip mode, per-cluster directories
(/root/reference/panfeed/input.py:235-259, panfeed.py:35-43,153-167) against the reference's own outputs."""
import gzip
import os

import pandas as pd
import pytest

from conftest import all_cases, case_records

pytestmark = pytest.mark.gpu

CASES = {c["name"]: c for c in all_cases()}


def _run_files(case, outdir, compress):
    from panfeed_amd.output import create_hash_files, create_kmer_stroi
    from panfeed_amd.panfeed import cluster_cutter, pattern_hasher, write_headers
    o = case["opts"]
    stroi = set(o["stroi"]) if o["stroi"] is not None else ""
    genepres = pd.DataFrame(columns=case["all_strains"])
    mf = o["multiple_files"]
    if mf:
        ks = hp = kh = None
    else:
        ks = create_kmer_stroi(outdir, compress)
        hp, kh = create_hash_files(outdir, compress)
        write_headers(hp, kh, genepres)
    rets = [cluster_cutter(x, o["klength"], stroi, mf, o["canon"], o["consider_missing"], outdir, compress)
            for x in case_records(case)]
    # two calls: the run-global patterns must carry over (and reset per cluster under --multiple-files)
    half = len(rets) // 2
    patterns = pattern_hasher(rets[:half], ks, hp, kh, genepres, o["patfilt"], o["maf"], outdir, patterns=None,
                              consider_missing_cluster=o["consider_missing"], compress=compress)
    patterns = pattern_hasher(rets[half:], ks, hp, kh, genepres, o["patfilt"], o["maf"], outdir, patterns=patterns,
                              consider_missing_cluster=o["consider_missing"], compress=compress)
    for f in (ks, hp, kh):
        if f is not None:
            f.close()
    return patterns


def _read(path, compress):
    if compress:
        with gzip.open(path + ".gz", "rt") as fh:
            return fh.read()
    with open(path) as fh:
        return fh.read()


@pytest.mark.parametrize("compress", [False, True], ids=["plain", "gzip"])
@pytest.mark.parametrize("name", ["rand12_basic", "rand12_noncanon", "rand40_shuffled_missing", "edge_k5", "rand12_k51"])
def test_single_directory(tmp_path, name, compress):
    case = CASES[name]
    out = str(tmp_path / "panfeed")
    os.mkdir(out)
    patterns = _run_files(case, out, compress)
    exp = case["expect"]
    for f in ("kmers.tsv", "kmers_to_hashes.tsv", "hashes_to_patterns.tsv"):
        assert _read(os.path.join(out, f), compress) == exp[f], f
    assert len(patterns) == exp["n_patterns"]
    assert sorted(os.listdir(out)) == sorted(f + (".gz" if compress else "") for f in
                                             ("kmers.tsv", "kmers_to_hashes.tsv", "hashes_to_patterns.tsv"))


@pytest.mark.parametrize("compress", [False, True], ids=["plain", "gzip"])
@pytest.mark.parametrize("name", ["rand12_mf", "edge_k5_mf"])
def test_multiple_files(tmp_path, name, compress):
    case = CASES[name]
    out = str(tmp_path / "panfeed")
    os.mkdir(out)
    _run_files(case, out, compress)
    exp = case["expect"]["dirs"]
    assert sorted(os.listdir(out)) == sorted(exp)
    for d in exp:
        for f in exp[d]:
            assert _read(os.path.join(out, d, f), compress) == exp[d][f], (d, f)
