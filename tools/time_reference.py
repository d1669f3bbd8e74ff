#!/usr/bin/env python3
"""Time the REFERENCE hot path (serial loop of __main__.py:350-356 and the reader/worker/writer topology of
__main__.py:299-344) on a few clusters of the bench workload.  Build container only (/root/reference);
numbers go to DESIGN.md / profiles/, nothing here ships.  Same pyfaidx note as tools/gen_golden.py."""
import io
import json
import multiprocessing as mp
import os
import sys
import time
import types

import numpy as np
import pandas as pd

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
_stub = types.ModuleType("pyfaidx")
_stub.Fasta = object
sys.modules.setdefault("pyfaidx", _stub)
sys.path.insert(0, "/root/reference")
from functools import partial  # noqa: E402

from panfeed.panfeed import cluster_cutter, pattern_hasher, write_headers  # noqa: E402
from panfeed.classes import Seqinfo as RefSeqinfo  # noqa: E402
from panfeed.__main__ import reader, worker, writer  # noqa: E402

from panfeed_amd import synth  # noqa: E402


def ref_records(n, S, flank):
    cl = synth.generate(n, S, first=0, flank=flank, n_rate=0.0)
    recs = []
    for c in cl:
        gs, idx, presab = c.record()
        recs.append(({k: [RefSeqinfo(*s) for s in v] for k, v in gs.items()}, idx, np.asarray(presab, dtype=int)))
    return recs, cl[0].names, sum(c.n_instances(31) for c in cl)


def serial(recs, names):
    hp, kh, ks = io.StringIO(), io.StringIO(), io.StringIO()
    genepres = pd.DataFrame(columns=names)
    write_headers(hp, kh, genepres)
    patterns = set()
    t0 = time.time()
    for x in recs:
        ret = cluster_cutter(x, 31, "", False, True, False, "unused")
        patterns = pattern_hasher((ret,), ks, hp, kh, genepres, True, 0.01, "unused", patterns=patterns)
    return time.time() - t0


def topology(recs, names, cores, outdir):
    """--cores N: 1 reader, N-2 workers, 1 writer (fork context), as __main__.py:299-344 wires them"""
    os.makedirs(outdir, exist_ok=True)
    hp = open(os.path.join(outdir, "hp.tsv"), "w")
    kh = open(os.path.join(outdir, "kh.tsv"), "w")
    ks = open(os.path.join(outdir, "ks.tsv"), "w")
    genepres = pd.DataFrame(columns=names)
    write_headers(hp, kh, genepres)
    ctx = mp.get_context("fork")
    ql = 3
    read_q = ctx.Queue(maxsize=ql * cores)
    write_q = ctx.Queue(maxsize=ql)
    iter_o = partial(cluster_cutter, klength=31, stroi="", multiple_files=False, canon=True,
                     consider_missing_cluster=False, output="unused")
    func_w = partial(pattern_hasher, kmer_stroi=ks, hash_pat=hp, kmer_hash=kh, genepres=genepres, patfilt=True,
                     maf=0.01, output="unused")
    procs = [ctx.Process(target=reader, args=(iter(recs), read_q, cores - 2))]
    procs += [ctx.Process(target=worker, args=(iter_o, read_q, write_q)) for _ in range(cores - 2)]
    procs += [ctx.Process(target=writer, args=(func_w, write_q, cores - 2))]
    t0 = time.time()
    for p in procs:
        p.start()
    for p in procs:
        p.join()
    return time.time() - t0


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    recs, names, ninst = ref_records(n, 1000, 100)
    out = {"clusters": n, "instances": ninst}
    dt = serial(recs[:2], names)
    n2 = sum(max(len(s.sequence) - 30, 0) for r in recs[:2] for v in r[0].values() for s in v)
    out["cores1"] = {"clusters": 2, "instances": n2, "seconds": dt, "instances_per_s": n2 / dt}
    cores = os.cpu_count()
    dt = topology(recs, names, cores, "/tmp/ref_topology")
    out[f"cores{cores}"] = {"workers": cores - 2, "seconds": dt, "instances_per_s": ninst / dt}
    print(json.dumps(out, indent=1))
