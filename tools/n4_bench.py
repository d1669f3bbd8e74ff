"""Row N4: the row filter of panfeed-get-clusters / panfeed-get-kmers over a large kmers_to_hashes.tsv, timed: the GPU
filter (panfeed_amd.downstream.RowFilter: file -> HBM -> rowfilter_kernel -> matching lines) against the statement
the reference runs, restated here for timing (get_clusters.py:89-94: pandas, 100 000-row chunks, `isin`).
    python tools/n4_bench.py [rows_in_millions] [passing_hashes]"""
import base64
import hashlib
import os
import sys
import tempfile
import time

import numpy as np
import pandas as pd

sys.path.insert(0, ".")
from panfeed_amd.downstream import RowFilter  # noqa: E402

mrows = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
n_pass = int(sys.argv[2]) if len(sys.argv) > 2 else 200
rng = np.random.default_rng(5)
n = int(mrows * 1e6)
# a pool of 200 000 pattern hashes; rows "cluster\tk-mer\thash"
pool = [base64.b64encode(hashlib.md5(str(i).encode()).digest()).decode() for i in range(200000)]
passing = set(rng.choice(len(pool), n_pass, replace=False).tolist())
tmp = tempfile.mkdtemp(prefix="n4bench_")
path = os.path.join(tmp, "kmers_to_hashes.tsv")
t0 = time.time()
acgt = np.frombuffer(b"ACGT", np.uint8)
with open(path, "wb") as fh:
    fh.write(b"cluster\tk-mer\thashed_pattern\n")
    step = 500000
    for a in range(0, n, step):
        m = min(step, n - a)
        km = acgt[rng.integers(0, 4, (m, 31))]
        hs = rng.integers(0, len(pool), m)
        cl = (a + np.arange(m)) // 3000
        lines = [b"group_%d\t%s\t%s\n" % (int(cl[i]), km[i].tobytes(), pool[int(hs[i])].encode()) for i in range(m)]
        fh.write(b"".join(lines))
size = os.path.getsize(path)
print(f"generated {n} rows, {size / 1e9:.2f} GB in {time.time() - t0:.0f} s", flush=True)
keys = [pool[i] for i in sorted(passing)]

# --- GPU row filter (file read from the page cache both times: the file was just written)
f = RowFilter(keys, first_field=False)
f.filter_file(path)                      # warm-up: first-use allocations
t0 = time.time()
header, body = f.filter_file(path)
t_gpu = time.time() - t0
st = f.stats()
f.close()
rows_gpu = body.count(b"\n")

# --- the reference's statement, restated: chunked pandas + isin
t0 = time.time()
kept = []
for x in pd.read_csv(path, sep="\t", chunksize=100000):
    kept.append(x[x["hashed_pattern"].isin(set(keys))])
kept = pd.concat(kept)
t_pd = time.time() - t0
assert len(kept) == rows_gpu, (len(kept), rows_gpu)
exp = "".join(f"{a}\t{b}\t{c}\n" for a, b, c in zip(kept["cluster"], kept["k-mer"], kept["hashed_pattern"])).encode()
assert exp == body, "the filtered rows differ"
print({"rows": n, "file_GB": round(size / 1e9, 3), "passing_hashes": n_pass, "kept_rows": rows_gpu,
       "gpu_filter_s": round(t_gpu, 3), "gpu_GBps_end_to_end": round(size / t_gpu / 1e9, 2),
       "gpu_device_ms_last_two_passes": round(st["device_ms"], 1),
       "pandas_chunks_s": round(t_pd, 2), "pandas_GBps": round(size / t_pd / 1e9, 3),
       "speedup": round(t_pd / t_gpu, 1)})
os.remove(path)
os.rmdir(tmp)
