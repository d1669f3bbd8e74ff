#!/usr/bin/env python3
"""Generate tests/golden/n4.json.gz: outputs of the REFERENCE's downstream tools,

    panfeed-get-clusters  /root/reference/panfeed/get_clusters.py:71-101
    panfeed-get-kmers     /root/reference/panfeed/get_kmers.py:88-145

run in this container (they need pandas only) on the three files of a few golden cases (tests/golden/*.json.gz: the
reference's own outputs) plus seeded pyseer-style association tables made here.  Only inputs we made and the
reference's outputs on them are stored; nothing of the reference is copied.

Both tools iterate over a Python `set` of cluster names, so the ORDER of their output blocks / lines changes with
PYTHONHASHSEED; the fixtures keep one run's text and the tests compare what is order-free (see tests/test_gpu_n4.py).

Usage: python tools/gen_golden_n4.py            (rewrites tests/golden/n4.json.gz)
"""
import contextlib
import gzip
import io
import json
import os
import sys
import tempfile

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
sys.path.insert(0, "/root/reference")
from panfeed import get_clusters as ref_get_clusters  # noqa: E402  (reference)
from panfeed import get_kmers as ref_get_kmers  # noqa: E402

from conftest import all_cases  # noqa: E402


def associations(kh_text, seed, nan_rate=0.05):
    """a pyseer-like table over the distinct hashes of a kmers_to_hashes.tsv: variant, af, filter-pvalue, lrt-pvalue,
    beta, beta-std-err, intercept, notes (pyseer prints floats as %.2E; some p-values missing)"""
    rng = np.random.default_rng(seed)
    hashes = []
    seen = set()
    for line in kh_text.split("\n")[1:]:
        if line:
            h = line.rsplit("\t", 1)[1]
            if h not in seen:
                seen.add(h)
                hashes.append(h)
    order = rng.permutation(len(hashes))
    rows = ["variant\taf\tfilter-pvalue\tlrt-pvalue\tbeta\tbeta-std-err\tintercept\tnotes"]
    for i in order:
        p = float(10 ** rng.uniform(-8, 0))
        lrt = "" if rng.random() < nan_rate else f"{p:.2E}"
        note = "bad-chisq" if rng.random() < 0.1 else ""
        rows.append(f"{hashes[i]}\t{rng.uniform(0.01, 0.5):.2E}\t{10 ** rng.uniform(-6, 0):.2E}\t{lrt}\t"
                    f"{rng.normal():.2E}\t{abs(rng.normal()) + 0.01:.2E}\t{rng.normal():.2E}\t{note}")
    # a few hashes that occur in no file
    for j in range(3):
        rows.append(f"{'Z' * 21}{j}==\t1.00E-01\t1.00E-03\t1.00E-09\t1.00E+00\t1.00E-01\t0.00E+00\t")
    return "\n".join(rows) + "\n"


def run_main(mod, argv):
    out = io.StringIO()
    old = sys.argv
    sys.argv = argv
    try:
        with contextlib.redirect_stdout(out):
            try:
                mod.main()
            except SystemExit as e:
                return out.getvalue(), int(e.code or 0)
    finally:
        sys.argv = old
        import logging
        logging.getLogger("panfeed").handlers.clear()
    return out.getvalue(), 0


def main():
    cases = {c["name"]: c for c in all_cases()}
    fixtures = []
    for name, seed in (("rand40_shuffled", 11), ("rand70_shuffled", 12), ("rand12_k51_noncanon", 13), ("rand12_basic", 14)):
        exp = cases[name]["expect"]
        assoc = associations(exp["kmers_to_hashes.tsv"], seed)
        with tempfile.TemporaryDirectory() as d:
            paths = {}
            for f in ("kmers.tsv", "kmers_to_hashes.tsv"):
                paths[f] = os.path.join(d, f)
                with open(paths[f], "w") as fh:
                    fh.write(exp[f])
            pa = os.path.join(d, "assoc.tsv")
            with open(pa, "w") as fh:
                fh.write(assoc)
            runs = []
            for tool, extra in (("get_clusters", ["-t", "1"]), ("get_clusters", ["-t", "0.01"]),
                                ("get_clusters", ["-t", "1e-12"]), ("get_clusters", ["-t", "0.05", "-c", "filter-pvalue"]),
                                ("get_clusters", ["-c", "no-such-column"]),
                                ("get_kmers", ["-t", "1"]), ("get_kmers", ["-t", "0.01"]),
                                ("get_kmers", ["-t", "0.01", "--only-passing"]),
                                ("get_kmers", ["-t", "0.3", "--clusters-per-iteration", "2"]),
                                ("get_kmers", ["-t", "0.3", "--clusters-per-iteration", "2", "--only-passing"]),
                                ("get_kmers", ["-t", "1e-12"])):
                po = os.path.join(d, "filtered.tsv")
                if os.path.exists(po):
                    os.remove(po)
                argv = [tool, "-a", pa, "-p", paths["kmers_to_hashes.tsv"], "-o", po] + extra
                if tool == "get_kmers":
                    argv += ["-k", paths["kmers.tsv"]]
                    out, rc = run_main(ref_get_kmers, argv)
                else:
                    out, rc = run_main(ref_get_clusters, argv)
                filt = open(po).read() if os.path.exists(po) else None
                runs.append({"tool": tool, "args": extra, "stdout": out, "rc": rc, "filtered": filt})
        fixtures.append({"case": name, "associations": assoc, "runs": runs})
    path = os.path.join(REPO, "tests", "golden", "n4.json.gz")
    with gzip.GzipFile(path, "wb", mtime=0) as fh:
        fh.write(json.dumps({"fixtures": fixtures}, indent=0).encode())
    print(path, sum(len(f["runs"]) for f in fixtures), "runs")


if __name__ == "__main__":
    main()
