"""panfeed-get-clusters / panfeed-get-kmers over the files of golden cases, against the outputs of
the reference's own `main()`s (tests/golden/n4.json.gz, made by tools/gen_golden_n4.py).  The row filter runs on the
GPU (pf_rowfilter_scan).

The reference iterates over Python sets of cluster names: the order of its printed clusters (get_clusters) and of its
per-bunch blocks (get_kmers with more clusters than --clusters-per-iteration) is arbitrary, so those are compared as
multisets of lines; everything else byte for byte."""
import gzip
import io
import json
import os

import pytest

from conftest import GOLDEN, all_cases

pytestmark = pytest.mark.gpu

with gzip.open(os.path.join(GOLDEN, "n4.json.gz"), "rb") as _fh:
    FIX = json.loads(_fh.read().decode())["fixtures"]
CASES = {c["name"]: c for c in all_cases()}
RUNS = [(f["case"], i) for f in FIX for i in range(len(f["runs"]))]


def _files(tmp_path, fx, gz=False):
    exp = CASES[fx["case"]]["expect"]
    paths = {}
    for name in ("kmers.tsv", "kmers_to_hashes.tsv"):
        p = tmp_path / (name + (".gz" if gz else ""))
        if gz:
            from panfeed_amd.output import ParallelGzipWriter
            w = ParallelGzipWriter(str(p), chunk_bytes=65536)     # several members
            w.write(exp[name])
            w.close()
        else:
            p.write_text(exp[name])
        paths[name] = str(p)
    pa = tmp_path / "assoc.tsv"
    pa.write_text(fx["associations"])
    return paths, str(pa)


def _run(tool, argv):
    from panfeed_amd import downstream
    out = io.StringIO()
    rc = 0
    try:
        rc = (downstream.get_clusters if tool == "get_clusters" else downstream.get_kmers)(argv, out=out)
    except SystemExit as e:
        rc = int(e.code or 0)
    return out.getvalue(), rc


@pytest.mark.parametrize("case,i", RUNS, ids=[f"{c}-{i}" for c, i in RUNS])
def test_downstream_tools_equal_reference(tmp_path, case, i):
    fx = next(f for f in FIX if f["case"] == case)
    run = fx["runs"][i]
    paths, pa = _files(tmp_path, fx)
    po = str(tmp_path / "filtered.tsv")
    argv = ["-a", pa, "-p", paths["kmers_to_hashes.tsv"], "-o", po] + run["args"]
    if run["tool"] == "get_kmers":
        argv += ["-k", paths["kmers.tsv"]]
    got, rc = _run(run["tool"], argv)
    assert rc == run["rc"]
    if run["filtered"] is None:
        assert not os.path.exists(po)
    else:
        assert open(po).read() == run["filtered"]
    exp = run["stdout"]
    if run["tool"] == "get_clusters":
        assert sorted(got.splitlines()) == sorted(exp.splitlines())
        return
    gl, el = got.splitlines(), exp.splitlines()
    assert gl[:1] == el[:1]                                    # the header, once
    assert sorted(gl[1:]) == sorted(el[1:])
    n_clusters = len({ln.split("\t")[0] for ln in el[1:]})
    cpi = int(run["args"][run["args"].index("--clusters-per-iteration") + 1]) if "--clusters-per-iteration" in run["args"] else 15
    if n_clusters <= cpi and "--only-passing" not in run["args"]:
        assert got == exp                                      # one bunch: kmers.tsv order, byte for byte


def test_gzip_inputs_and_blocks(tmp_path):
    """.gz inputs (several gzip members) and blocks far smaller than the file: lines cut by block ends are carried over"""
    from panfeed_amd import downstream
    fx = FIX[0]
    paths, pa = _files(tmp_path, fx, gz=True)
    run = next(r for r in fx["runs"] if r["tool"] == "get_kmers" and r["args"] == ["-t", "0.01"])
    old = downstream.BLOCK_BYTES
    try:
        for blk in (257, 4096, old):
            downstream.BLOCK_BYTES = blk
            got, rc = _run("get_kmers", ["-a", pa, "-p", paths["kmers_to_hashes.tsv"], "-k", paths["kmers.tsv"], "-t", "0.01"])
            assert rc == 0 and got == run["stdout"], blk
    finally:
        downstream.BLOCK_BYTES = old


def test_rowfilter_exact_keys(tmp_path):
    """first-field and last-field keys, keys that are prefixes of one another, an unterminated last line, no key at all"""
    from panfeed_amd.downstream import RowFilter
    text = b"h1\tx\tAAA\nab\t1\tkey\nabc\t2\tkey2\nab\t3\tkex\n\t\tkey\nabc\tlast\tkey"
    p = tmp_path / "t.tsv"
    p.write_bytes(text)
    f = RowFilter(["key"], first_field=False)
    assert f.filter_file(str(p)) == (b"h1\tx\tAAA\n", b"ab\t1\tkey\n\t\tkey\nabc\tlast\tkey\n")
    f.close()
    f = RowFilter(["ab", ""], first_field=True)
    assert f.filter_file(str(p))[1] == b"ab\t1\tkey\nab\t3\tkex\n\t\tkey\n"
    f.close()
    f = RowFilter([], first_field=True)
    assert f.filter_file(str(p))[1] == b""
    f.close()
