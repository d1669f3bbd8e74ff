import gzip
import json
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

from panfeed_amd.classes import Seqinfo  # noqa: E402

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_cases(fname):
    with gzip.open(os.path.join(GOLDEN, fname), "rb") as fh:
        return json.loads(fh.read().decode())["cases"]


def case_records(case):
    """fixture clusters -> reference-shaped (gene_sequences, idx, clusterpresab) records"""
    out = []
    for cj in case["clusters"]:
        gs = {name: [Seqinfo(*s) for s in seqs] for name, seqs in cj["strains"]}
        out.append((gs, cj["idx"], np.array(cj["presab"], dtype=np.int64)))
    return out


def all_cases():
    return load_cases("handmade.json.gz") + load_cases("seeded.json.gz") + load_cases("wide.json.gz")


def case_ids(cases):
    return [c["name"] for c in cases]
