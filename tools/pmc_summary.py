#!/usr/bin/env python3
"""rocprofv3 --pmc output directories -> one JSON of HBM bytes per kernel and step (profiles/rNN/*_pmc_traffic.json).

usage: pmc_summary.py STEPS_PROFILED OUT.json DIR [DIR ...]
Each DIR is the -d directory of one `rocprofv3 --kernel-trace --pmc <COUNTER> -- python bench.py ...` pass
(FETCH_SIZE and WRITE_SIZE in separate passes, as MI355X_MICROARCH.md prescribes).  FETCH_SIZE / WRITE_SIZE are KiB;
on gfx950 FETCH_SIZE counts 64 B per 128-B request for wide coalesced reads, hence the x2 on the read side."""
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
    rd = 2.0 * fetch * 1024.0 / steps
    wr = write * 1024.0 / steps
    res[name] = {"dispatches": nd, "dispatches_per_step": nd / steps, "FETCH_SIZE_KiB_total": fetch,
                 "WRITE_SIZE_KiB_total": write, "hbm_read_bytes_per_step_x2corrected": rd,
                 "hbm_write_bytes_per_step": wr, "hbm_bytes_per_step": rd + wr}
    for cn, v in c.items():
        if cn not in ("FETCH_SIZE", "WRITE_SIZE"):
            res[name][cn + "_per_step"] = v / steps
json.dump({"steps_profiled": steps, "kernels": res,
           "note": "FETCH_SIZE/WRITE_SIZE are KiB as rocprofv3 reports them. On gfx950 FETCH_SIZE counts 64 B per 128-B "
                   "request for wide coalesced streaming reads (MI355X_MICROARCH.md, HBM): hbm_read_bytes = 2 * FETCH_SIZE "
                   "* 1024; WRITE_SIZE is exact for 16-B stores."}, open(out_path, "w"), indent=1)
print(out_path, {n: round(v["hbm_bytes_per_step"] / 1e9, 3) for n, v in res.items()})
