"""PCIe-inclusive rate of the hot path: pf_submit on a HOST-resident packed batch (the boundary hands over host buffers:
packed segments + metadata go up over PCIe inside the call) next to the same batch resident in HBM."""
import json, sys, time
sys.path.insert(0, ".")
from panfeed_amd import devbatch, synth
from panfeed_amd.engine import Engine
from panfeed_amd.packing import build_batch_native
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
S, k = 1000, 31
cl = synth.generate(n, S, flank=100, n_rate=0.0)
recs = [c.record() for c in cl]
eng = Engine(klength=k, max_strains=1024, max_items=16384, pattern_capacity=1 << 23)
hb = build_batch_native(recs, k, True, eng.W)
del recs
out = dict(clusters=n, instances=int(hb.n_instances), packed_bytes=int(hb.packed.nbytes))
for name, f in (("host_resident", lambda: eng.submit_host_batch(hb)),):
    f()
    ts = []
    for _ in range(3):
        eng.L.pf_reset_patterns(eng.ctx)
        t = time.time(); f(); ts.append(time.time() - t)
    out[name + "_s"] = min(ts)
    out[name + "_inst_per_s"] = out["instances"] / min(ts)
db = devbatch.from_host_batch(eng, hb)
db.submit()
ts = []
for _ in range(3):
    eng.L.pf_reset_patterns(eng.ctx)
    t = time.time(); db.submit(); ts.append(time.time() - t)
out["hbm_resident_s"] = min(ts)
out["hbm_resident_inst_per_s"] = out["instances"] / min(ts)
out["upload_GBps"] = out["packed_bytes"] / max(out["host_resident_s"] - out["hbm_resident_s"], 1e-9) / 1e9
print(json.dumps(out))
