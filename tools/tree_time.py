"""Device times of one submit of N synthetic clusters whose alleles descend from one another (synth's "tree" model),
second run of two.  usage: python tools/tree_time.py [clusters] [mean_alleles] [star|tree] [max_items]"""
import sys

import numpy as np

sys.path.insert(0, ".")
from panfeed_amd import devbatch, synth  # noqa: E402
from panfeed_amd.engine import Engine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
ma = float(sys.argv[2]) if len(sys.argv) > 2 else 150
model = sys.argv[3] if len(sys.argv) > 3 else "tree"
max_items = int(sys.argv[4]) if len(sys.argv) > 4 else 32768
eng = Engine(klength=31, max_strains=1024, pattern_capacity=1 << 24, max_items=max_items)
cl = synth.generate(n, 1000, flank=100, n_rate=0.0, mean_alleles=ma, allele_decay=1.0, allele_model=model)
db = devbatch.from_synth(eng, cl, 31)
db.submit()
eng.L.pf_reset_patterns(eng.ctx)
db.submit()
t = eng.timing()
print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in t.items()})
