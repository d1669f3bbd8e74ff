"""files -> three files (pipeline.run_files), alternating on one box: the one-pass ingest (genomes to the GPU as their files
are read, pf_pangenome_open_device) against the two-step start-up (read into host strings, then upload; with and without
its overlap).  usage: python tools/e2e_ab.py [clusters] [rounds] [workers]"""
import json
import os
import shutil
import sys
import tempfile
import time

sys.path.insert(0, ".")
from panfeed_amd import synth  # noqa: E402
from panfeed_amd.pipeline import run_files  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
workers = int(sys.argv[3]) if len(sys.argv) > 3 else 0
S, k, up = 1000, 31, 100
scratch = tempfile.mkdtemp(prefix="pf_e2e_ab_")
try:
    cl = synth.generate(n, S, first=0, flank=up, n_rate=0.001)
    ninst = sum(c.n_instances(k) for c in cl)
    csvp, gffs, _ = synth.write_pangenome(scratch, cl, missing_gene_rate=0.0, workers=workers)
    del cl
    for r in range(rounds):
        for one_pass, overlap in ((True, True), (False, True), (False, False)):
            od = os.path.join(scratch, "out")
            # (the reader of the run before gives its memory back on a thread of its own -- gigabytes of contig strings after
            # a two-step run -- and the next open waits for that thread: a real run is a fresh process)
            time.sleep(2.0)
            t0 = time.time()
            st = run_files(csvp, os.path.join(scratch, "gffs"), od, klength=k, upstream=up, downstream=up, batch_clusters=256,
                           device_text=True, overlap=overlap, one_pass=one_pass)
            dt = time.time() - t0
            shutil.rmtree(od)
            print(json.dumps({"one_pass": one_pass, "overlap": overlap, "seconds": round(dt, 4), "inst_per_s": float("%.3e" % (ninst / dt)),
                              "stages": {a: round(b, 4) for a, b in st["stages"].items()}}), flush=True)
finally:
    shutil.rmtree(scratch, ignore_errors=True)
