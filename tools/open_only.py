"""time Pangenome open alone, both ways (no batches): python open_only.py dir"""
import sys, time, os
sys.path.insert(0, ".")
from panfeed_amd import native_input as ni
from panfeed_amd.engine import Engine
d = sys.argv[1]
csvp = os.path.join(d, "gene_presence_absence.csv")
for rep in range(3):
    for mode in ("two_step", "one_pass"):
        eng = Engine(klength=31, max_strains=1024, max_items=512)
        t0 = time.time()
        pg = ni.Pangenome(csvp, os.path.join(d, "gffs"), None, 100, 100, engine=eng if mode == "one_pass" else None)
        t1 = time.time()
        if mode == "two_step":
            pg.make_resident(eng)
        t2 = time.time()
        print(mode, "open %.3f resident %.3f" % (t1 - t0, t2 - t1), flush=True)
        pg.close(); eng.close()
