"""Row N1 timing: files on disk -> native reader -> packer -> GPU -> TSV text, per stage.
usage: python tools/n1_bench.py [clusters] [samples]   (needs a GPU unless --host-only)"""
import json
import sys
import tempfile
import time

sys.path.insert(0, ".")
from panfeed_amd import native_input as ni, synth  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
host_only = "--host-only" in sys.argv
C_, S = (int(args[0]) if args else 300), (int(args[1]) if len(args) > 1 else 1000)
k, up, down = 31, 100, 100
cl = synth.generate(C_, S, flank=up)          # BASELINE configs[1] shape: allele-specific 100-base flanks
d = tempfile.mkdtemp()
t = time.time(); csvp, gffs, fas = synth.write_pangenome(d, cl, missing_gene_rate=0.0); t_write = time.time() - t
gn = sorted(gffs)
t = time.time()
pg = ni.Pangenome(csvp, None, None, up, down, genome_names=gn, gff_paths=[gffs[n] for n in gn])
t_open = time.time() - t
W = (S + 31) // 32
t = time.time(); inst = 0
for hb in pg.batches(k, True, W, max_clusters=64):
    inst += hb.n_instances
t_pack = time.time() - t
pg.close()
out = dict(clusters=C_, samples=S, instances=inst, write_s=round(t_write, 2), open_parse_s=round(t_open, 3),
           read_pack_s=round(t_pack, 3), read_pack_inst_per_s=inst / t_pack)
if not host_only:
    from panfeed_amd.engine import Engine
    for mode in ("text", "resident", "resident_device_text"):
        eng = Engine(klength=k, max_strains=W * 32)
        pg = ni.Pangenome(csvp, None, None, up, down, genome_names=gn, gff_paths=[gffs[n] for n in gn])
        t_up = 0.0
        if mode == "resident_device_text":
            pg.make_resident(eng)
        if mode == "resident":
            t = time.time(); pg.make_resident(eng); t_up = time.time() - t
            t = time.time(); ninst2 = 0
            for hb in pg.batches(k, True, W, max_clusters=64):
                ninst2 += hb.n_instances
            out["resident_read_pack_s"] = round(time.time() - t, 3)
            assert ninst2 == inst
            pg.close()
            pg = ni.Pangenome(csvp, None, None, up, down, genome_names=gn, gff_paths=[gffs[n] for n in gn])
            pg.make_resident(eng)
        t = time.time(); nb = 0; dev = 0.0
        for o in eng.run_pangenome(pg, batch_clusters=64, device_text=mode.endswith("device_text")):
            nb += len(o.kmers_to_hashes); dev += o.timing["total_ms"]
        t_e2e = time.time() - t
        out[mode] = dict(e2e_s=round(t_e2e, 3), e2e_inst_per_s=inst / t_e2e, device_ms=round(dev, 1), text_bytes=nb,
                         genome_upload_s=round(t_up, 3))
        pg.close(); eng.close()
if not host_only:
    # files -> files through pipeline.run_files (writer thread; gzip = level 9 members on all host threads)
    import os, shutil
    from panfeed_amd.pipeline import run_files
    for compress in (False, True):
        od = os.path.join(tempfile.mkdtemp(), "panfeed")      # run_files refuses an existing directory, as the reference does
        t = time.time()
        st = run_files(csvp, os.path.join(d, "gffs"), od, klength=k, upstream=up, downstream=down, compress=compress,
                       batch_clusters=64)
        dt = time.time() - t
        size = sum(os.path.getsize(os.path.join(od, f)) for f in os.listdir(od))
        out["files_to_files_gzip" if compress else "files_to_files"] = dict(
            seconds=round(dt, 3), inst_per_s=st["instances"] / dt, text_bytes=st["bytes"], file_bytes=size)
        shutil.rmtree(os.path.dirname(od))
print(json.dumps(out))
