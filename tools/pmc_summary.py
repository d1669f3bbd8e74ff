#!/usr/bin/env python3
"""rocprofv3 --pmc output directories -> one JSON of HBM bytes per kernel and step (profiles/rNN/*_pmc_traffic.json).

usage: pmc_summary.py STEPS_PROFILED OUT.json DIR [DIR ...]
Each DIR is the -d directory of one `rocprofv3 --kernel-trace --pmc <COUNTERS> -- python bench.py ...` pass
(FETCH_SIZE and WRITE_SIZE in separate passes, as MI355X_MICROARCH.md prescribes; a third pass for SQ counters).
FETCH_SIZE / WRITE_SIZE are KiB.  On gfx950 FETCH_SIZE counts 64 B per 128-B request for WIDE coalesced streaming
reads (16 B per lane): that correction (x2) is applied to the kernels whose reads are of that kind -- cluster_dedup_kernel
(the packed input, 16 B per lane), synth / gather copies -- and NOT to kernels that read 4 or 8 bytes per lane
(finish, scan, md5, rows, emit: uncalibrated widths, the guide says; both readings are kept in the JSON)."""
import csv
import glob
import json
import os
import re
import sys

steps = int(sys.argv[1])
out_path = sys.argv[2]
kern = {}
for d in sys.argv[3:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                name = re.sub(r"\(.*$", "", r["Kernel_Name"]).replace("void ", "").strip()
                if not name.startswith("pf::"):
                    continue
                k = kern.setdefault(name, {"dispatch_ids": {}, "counters": {}})
                k["counters"][r["Counter_Name"]] = k["counters"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
                k["dispatch_ids"].setdefault(r["Counter_Name"], set()).add(r["Dispatch_Id"])
res = {}
for name, k in sorted(kern.items()):
    c = k["counters"]
    nd = max(len(v) for v in k["dispatch_ids"].values())
    fetch, write = c.get("FETCH_SIZE", 0.0), c.get("WRITE_SIZE", 0.0)
    wide = any(t in name for t in ("cluster_dedup_kernel", "synth_expand_kernel", "gather_segments_kernel"))
    rd = (2.0 if wide else 1.0) * fetch * 1024.0 / steps
    wr = write * 1024.0 / steps
    res[name] = {"dispatches": nd, "dispatches_per_step": nd / steps, "FETCH_SIZE_KiB_total": fetch,
                 "WRITE_SIZE_KiB_total": write, "fetch_x2_applied": wide, "hbm_read_bytes_per_step": rd,
                 "hbm_read_bytes_per_step_if_x2": 2.0 * fetch * 1024.0 / steps,
                 "hbm_write_bytes_per_step": wr, "hbm_bytes_per_step": rd + wr}
    for cn, v in c.items():
        if cn not in ("FETCH_SIZE", "WRITE_SIZE"):
            res[name][cn + "_per_step"] = v / steps
import hashlib  # noqa: E402
_h = hashlib.sha256()
_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _f in ("pf_kernels.h", "pf_api.hip"):
    with open(os.path.join(_root, "panfeed_amd", "csrc", _f), "rb") as _fh:
        _h.update(_fh.read())
json.dump({"steps_profiled": steps, "commit": os.environ.get("PF_COMMIT"), "kernel_source_sha16": _h.hexdigest()[:16], "kernels": res,
           "note": "FETCH_SIZE/WRITE_SIZE are KiB as rocprofv3 reports them. On gfx950 FETCH_SIZE counts 64 B per 128-B "
                   "request for wide coalesced streaming reads (MI355X_MICROARCH.md, HBM): hbm_read_bytes = 2 * FETCH_SIZE "
                   "* 1024 where fetch_x2_applied (16-B-per-lane streaming reads), FETCH_SIZE * 1024 elsewhere (narrower "
                   "reads are uncalibrated: hbm_read_bytes_per_step_if_x2 is the other reading); WRITE_SIZE is exact for "
                   "16-B stores. SQ_* counters: sums over the dispatches of a step."}, open(out_path, "w"), indent=1)
print(out_path, {n: round(v["hbm_bytes_per_step"] / 1e9, 3) for n, v in res.items()})
