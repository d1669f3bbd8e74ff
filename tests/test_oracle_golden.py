"""The CPU oracle (oracle/panfeed_oracle.c) against outputs of the reference itself
(tests/golden/, written by tools/gen_golden.py).  Byte-for-byte; this is what pins the oracle."""
import json
import os

import pytest

from conftest import GOLDEN, all_cases, case_ids, case_records
from oracle import oracle as po

CASES = all_cases()


def run_oracle(case, threads=1):
    o = case["opts"]
    stroi = set(o["stroi"]) if o["stroi"] is not None else ""
    run = po.OracleRun(klength=o["klength"], stroi=stroi, canon=o["canon"],
                       consider_missing=o["consider_missing"], patfilt=o["patfilt"], maf=o["maf"],
                       multiple_files=o["multiple_files"], threads=threads)
    recs = case_records(case)
    if not o["multiple_files"]:
        run.feed(recs)
        k, kh, hp = run.texts()
        return {"kmers.tsv": po.kmers_tsv_header() + k,
                "kmers_to_hashes.tsv": po.kmers_to_hashes_header() + kh,
                "hashes_to_patterns.tsv": po.hashes_to_patterns_header(case["all_strains"]) + hp,
                "n_patterns": run.stats()["patterns"]}
    dirs = {}
    for rec in recs:      # one directory per cluster, headers rewritten each time (panfeed.py:153-167)
        run.clear_text()
        run.feed([rec])
        k, kh, hp = run.texts()
        dirs[rec[1]] = {"kmers.tsv": po.kmers_tsv_header() + k,
                        "kmers_to_hashes.tsv": po.kmers_to_hashes_header() + kh,
                        "hashes_to_patterns.tsv": po.hashes_to_patterns_header(case["all_strains"]) + hp}
    return {"dirs": dirs}


@pytest.mark.parametrize("case", CASES, ids=case_ids(CASES))
def test_oracle_matches_reference(case):
    got = run_oracle(case)
    exp = case["expect"]
    if "dirs" in exp:
        assert got["dirs"].keys() == exp["dirs"].keys()
        for d in exp["dirs"]:
            for f in exp["dirs"][d]:
                assert got["dirs"][d][f] == exp["dirs"][d][f], (d, f)
    else:
        for f in ("kmers_to_hashes.tsv", "hashes_to_patterns.tsv", "kmers.tsv"):
            assert got[f] == exp[f], f
        # len(patterns) counts distinct hashes; the oracle counts rows written: equal by construction
        assert got["n_patterns"] == exp["n_patterns"]


@pytest.mark.parametrize("case", [c for c in CASES if not c["opts"]["multiple_files"]][::5],
                         ids=lambda c: c["name"])
def test_oracle_threaded_is_identical(case):
    assert run_oracle(case, threads=4) == run_oracle(case, threads=1)


def test_md5_rfc1321_suite():
    kat = {b"": "d41d8cd98f00b204e9800998ecf8427e", b"a": "0cc175b9c0f1b6a831c399e269772661",
           b"abc": "900150983cd24fb0d6963f7d28e17f72", b"message digest": "f96b697d7cb7938d525a2f31aaf161d0",
           b"abcdefghijklmnopqrstuvwxyz": "c3fcd3d76192e4007dfb496cca67e13b",
           b"12345678901234567890123456789012345678901234567890123456789012345678901234567890":
               "57edf4a22be3c955ac49da2e2107b67a"}
    for msg, hx in kat.items():
        assert po.md5_hex(msg) == hx


def test_hash_image_kats():
    with open(os.path.join(GOLDEN, "hash_kat.json")) as fh:
        for k in json.load(fh):
            assert po.md5_b64(bytes.fromhex(k["hex"])) == k["b64"], k
