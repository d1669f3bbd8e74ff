"""Device-side text rendering at scale: one submit of N clusters x 1000 samples, then pf_render_device (text kernels +
copy of the text to pinned host memory) against pf_fetch + the host renderers."""
import ctypes as C, json, sys, time
sys.path.insert(0, ".")
from panfeed_amd import _lib, devbatch, synth
from panfeed_amd.engine import Engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
S, k = 1000, 31
eng = Engine(klength=k, max_strains=1024, max_items=16384, pattern_capacity=1 << 23)
cl = synth.generate(n, S, flank=100, n_rate=0.0)
db = devbatch.from_synth(eng, cl, k)
db.submit()
names = (C.c_char_p * n)(*[c.idx.encode() for c in cl])
kh, kn, hp, hn = C.c_void_p(), C.c_uint64(), C.c_void_p(), C.c_uint64()
out = {}
for rep in range(3):
    t = time.time()
    _lib.check(eng.L.pf_render_device(eng.ctx, names, None, 0, C.byref(kh), C.byref(kn), C.byref(hp), C.byref(hn)))
    out["device_render_s"] = time.time() - t
nbytes = kn.value + hn.value
t = time.time(); res = eng.fetch(); out["fetch_s"] = time.time() - t
buf, nb = C.c_void_p(), C.c_uint64()
t = time.time()
_lib.check(eng.L.pf_render_kmers_to_hashes(eng.ctx, names, None, C.byref(buf), C.byref(nb), None))
b2, n2 = C.c_void_p(), C.c_uint64()
_lib.check(eng.L.pf_render_hashes_to_patterns(eng.ctx, C.byref(b2), C.byref(n2)))
out["host_render_s"] = time.time() - t
same = C.string_at(kh, kn.value) == C.string_at(buf, nb.value) and C.string_at(hp, hn.value) == C.string_at(b2, n2.value)
out.update(clusters=n, text_bytes=nbytes, identical=bool(same), device_GBps=nbytes / out["device_render_s"] / 1e9,
           host_GBps=nbytes / (out["fetch_s"] + out["host_render_s"]) / 1e9)
print(json.dumps(out))
