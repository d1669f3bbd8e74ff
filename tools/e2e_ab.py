"""files -> three files (pipeline.run_files) with and without the overlapped start-up (context made while the reader opens,
genomes uploaded while the packer works), alternating, on one box.  usage: python tools/e2e_ab.py [clusters] [rounds]"""
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
S, k, up = 1000, 31, 100
scratch = tempfile.mkdtemp(prefix="pf_e2e_ab_")
try:
    cl = synth.generate(n, S, first=0, flank=up, n_rate=0.001)
    ninst = sum(c.n_instances(k) for c in cl)
    csvp, gffs, _ = synth.write_pangenome(scratch, cl, missing_gene_rate=0.0)
    del cl
    for r in range(rounds):
        for overlap in (True, False):
            od = os.path.join(scratch, "out")
            t0 = time.time()
            st = run_files(csvp, os.path.join(scratch, "gffs"), od, klength=k, upstream=up, downstream=up, batch_clusters=256,
                           device_text=True, overlap=overlap)
            dt = time.time() - t0
            shutil.rmtree(od)
            print(json.dumps({"overlap": overlap, "seconds": round(dt, 4), "inst_per_s": float("%.3e" % (ninst / dt)),
                              "stages": {a: round(b, 4) for a, b in st["stages"].items()}}), flush=True)
finally:
    shutil.rmtree(scratch, ignore_errors=True)
