"""Where the end-to-end time goes (serial, no overlap): pack / submit / fetch / render / write, per stage."""
import json, os, sys, tempfile, time
sys.path.insert(0, ".")
from panfeed_amd import synth
from panfeed_amd.engine import Engine
from panfeed_amd.packing import build_batch_native

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
S, k, bc = 1000, 31, 128
cl = synth.generate(n, S, flank=100, n_rate=0.001)
recs = [c.record() for c in cl]
ninst = sum(c.n_instances(k) for c in cl)
eng = Engine(klength=k, max_strains=1024)
eng.run(recs[:8]); eng.close()
eng = Engine(klength=k, max_strains=1024)
T = dict(pack=0.0, submit=0.0, fetch=0.0, render=0.0, write=0.0)
out = tempfile.mkdtemp()
nb = 0
with open(os.path.join(out, "a"), "w") as fa, open(os.path.join(out, "b"), "w") as fb:
    for i in range(0, n, bc):
        t = time.time(); hb = build_batch_native(recs[i:i + bc], k, True, eng.W, first_ordinal=i); T["pack"] += time.time() - t
        t = time.time(); eng.submit_host_batch(hb); T["submit"] += time.time() - t
        t = time.time(); res = eng.fetch(); T["fetch"] += time.time() - t
        t = time.time(); o = eng._render(hb, res); T["render"] += time.time() - t
        t = time.time(); fa.write(o.kmers_to_hashes); fb.write(o.hashes_to_patterns); T["write"] += time.time() - t
        nb += len(o.kmers_to_hashes) + len(o.hashes_to_patterns)
tot = sum(T.values())
print(json.dumps(dict(clusters=n, instances=ninst, out_bytes=nb, total_s=tot, inst_per_s=ninst / tot, **{k_: round(v, 3) for k_, v in T.items()})))
