"""Host stages of the files -> files run without a GPU: open + parse, then the reader + packer loop with by-reference
records (what the packer thread of pipeline.run_files does), timed per stage.  usage: python tools/host_stages.py [clusters] [samples]"""
import ctypes as C
import sys
import tempfile
import time

sys.path.insert(0, ".")
from panfeed_amd import _lib, native_input as ni, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
cl = synth.generate(n, S, flank=100, n_rate=0.001)
d = tempfile.mkdtemp()
t0 = time.time()
csvp, gffs, fas = synth.write_pangenome(d, cl, missing_gene_rate=0.0)
print("write_pangenome", round(time.time() - t0, 2), "s")
L = _lib.load()
for rep in range(2):
    t0 = time.time()
    pg = ni.Pangenome(csvp, d + "/gffs", None, 100, 100)
    t_open = time.time() - t0
    t0 = time.time()
    pg.assign_store()
    t_store = time.time() - t0
    t0 = time.time()
    nb = 0
    for hb in pg.batches(31, True, (S + 31) // 32, max_clusters=256):
        nb += 1
    t_pack = time.time() - t0
    t0 = time.time()
    pg.close()
    print(f"open {t_open:.3f}  store {t_store:.4f}  batches {t_pack:.3f} ({nb})  close {time.time() - t0:.3f}")
