#!/usr/bin/env python3
"""End-to-end leg on the GPU box: reference-shaped records (host strings) -> the three TSV files, through the
pipelined driver (native packer -> H2D -> kernels -> D2H -> native renderers -> file writes).  The record
generation is outside the timed region.  Prints one JSON line; numbers go to DESIGN.md section 5/6."""
import json
import os
import sys
import tempfile
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from panfeed_amd import synth  # noqa: E402
from panfeed_amd.engine import Engine, KMERS_TSV_HEADER, KMERS_TO_HASHES_HEADER, hashes_to_patterns_header  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    S, k = 1000, 31
    cl = synth.generate(n, S, first=0, flank=100, n_rate=0.001)
    names = cl[0].names
    recs = [c.record() for c in cl]
    ninst = sum(c.n_instances(k) for c in cl)
    stroi = {names[10]}
    out = tempfile.mkdtemp(prefix="pf_e2e_")
    eng = Engine(klength=k, max_strains=1024, stroi=stroi)
    # warm up the context (first kernel launches, allocations)
    for _ in eng.run_stream(iter(recs[:8]), batch_clusters=8):
        pass
    eng.close()
    eng = Engine(klength=k, max_strains=1024, stroi=stroi)
    t0 = time.time()
    nbytes = 0
    with open(os.path.join(out, "kmers.tsv"), "w") as ks, open(os.path.join(out, "kmers_to_hashes.tsv"), "w") as kh, \
            open(os.path.join(out, "hashes_to_patterns.tsv"), "w") as hp:
        ks.write(KMERS_TSV_HEADER)
        kh.write(KMERS_TO_HASHES_HEADER)
        hp.write(hashes_to_patterns_header(names))
        dev_ms = 0.0
        for o in eng.run_stream(iter(recs), batch_clusters=128):
            ks.write(o.kmers_tsv)
            kh.write(o.kmers_to_hashes)
            hp.write(o.hashes_to_patterns)
            nbytes += len(o.kmers_tsv) + len(o.kmers_to_hashes) + len(o.hashes_to_patterns)
            dev_ms += o.timing["total_ms"]
    dt = time.time() - t0
    print(json.dumps({"clusters": n, "samples": S, "instances": ninst, "seconds": dt, "instances_per_s": ninst / dt,
                      "output_bytes": nbytes, "device_ms_total": dev_ms,
                      "note": "host strings -> three files; 1 target strain; record generation excluded"}))
    eng.close()
    # the same with the two big files' text written by the GPU (bytes into binary files; kmers.tsv from the library's
    # renderer, handed over without a copy)
    from panfeed_amd.engine import OwnedText
    eng = Engine(klength=k, max_strains=1024, stroi=stroi)
    t0 = time.time()
    nbytes = 0
    with open(os.path.join(out, "kmers.tsv"), "wb") as ks, open(os.path.join(out, "kmers_to_hashes.tsv"), "wb") as kh, \
            open(os.path.join(out, "hashes_to_patterns.tsv"), "wb") as hp:
        ks.write(KMERS_TSV_HEADER.encode())
        kh.write(KMERS_TO_HASHES_HEADER.encode())
        hp.write(hashes_to_patterns_header(names).encode())
        for o in eng.run_stream(iter(recs), batch_clusters=128, device_text=True):
            kt = o.kmers_tsv
            if isinstance(kt, OwnedText):
                ks.write(kt.view)
                nbytes += len(kt)
                kt.release()
            else:
                ks.write(kt.encode() if isinstance(kt, str) else kt)
                nbytes += len(kt)
            kh.write(o.kmers_to_hashes)
            hp.write(o.hashes_to_patterns)
            nbytes += len(o.kmers_to_hashes) + len(o.hashes_to_patterns)
    dt = time.time() - t0
    print(json.dumps({"clusters": n, "samples": S, "instances": ninst, "seconds": dt, "instances_per_s": ninst / dt,
                      "output_bytes": nbytes, "note": "the same, text of the two big files written on the device"}))
    eng.close()


if __name__ == "__main__":
    main()
