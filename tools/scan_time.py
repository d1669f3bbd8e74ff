"""Device times of one submit of N synthetic clusters (1 000 samples, k = 31, +-100 bp), second run of two.
usage: python tools/scan_time.py [clusters]"""
import sys

sys.path.insert(0, ".")
from panfeed_amd import devbatch, synth  # noqa: E402
from panfeed_amd.engine import Engine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
eng = Engine(klength=31, max_strains=1024, pattern_capacity=1 << 24)
cl = synth.generate(n, 1000, flank=100, n_rate=0.0)
db = devbatch.from_synth(eng, cl, 31)
db.submit()
eng.L.pf_reset_patterns(eng.ctx)
db.submit()
t = eng.timing()
print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in t.items()})
