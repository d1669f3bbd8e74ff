"""cProfile of the resident-genome end-to-end loop (files -> text), to see what the host still does per batch."""
import cProfile, pstats, sys, tempfile, time
sys.path.insert(0, ".")
from panfeed_amd import native_input as ni, synth
from panfeed_amd.engine import Engine
C_, S, k = (int(sys.argv[1]) if len(sys.argv) > 1 else 300), 1000, 31
cl = synth.generate(C_, S, flank=100)
d = tempfile.mkdtemp()
csvp, gffs, fas = synth.write_pangenome(d, cl, missing_gene_rate=0.0)
gn = sorted(gffs)
eng = Engine(klength=k, max_strains=1024)
pg = ni.Pangenome(csvp, None, None, 100, 100, genome_names=gn, gff_paths=[gffs[n] for n in gn]).make_resident(eng)
for o in eng.run_pangenome(pg, batch_clusters=64):
    pass
pg.close()
pg = ni.Pangenome(csvp, None, None, 100, 100, genome_names=gn, gff_paths=[gffs[n] for n in gn]).make_resident(eng)
eng.L.pf_reset_patterns(eng.ctx)
pr = cProfile.Profile(); pr.enable(); t = time.time()
nb = 0
for o in eng.run_pangenome(pg, batch_clusters=64):
    nb += len(o.kmers_to_hashes) + len(o.hashes_to_patterns)
dt = time.time() - t; pr.disable()
print("e2e", dt, "bytes", nb)
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
