"""One rank of a sharded run, started by tests/test_sharded_cpu.py (mode oracle: no GPU, the CPU oracle stands in
for the engine so that the product's merge / part-file / assembly code runs under gloo) and tests/test_gpu_sharded.py
(mode gpu: a real Engine per rank, all on cuda:0, gloo collectives on host tensors).

    python tests/sharded_worker.py MODE RANK WORLD PORT OUTDIR CASE [compress] [method]

CASE: a golden case name, seeded:<n_clusters>:<samples>:<k>:<flank>:<seed>, or (mode gpu)
files:<table.csv>:<gff dir>:<k>:<upstream>:<downstream>:<target,target> for `run_files_sharded`.
"""
import base64
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))


def load_case(spec):
    """(records, all strains, options dict)"""
    if spec.startswith("seeded:"):
        from panfeed_amd import synth
        _, n, S, k, flank, seed = spec.split(":")
        cl = synth.generate(int(n), int(S), first=int(seed), flank=int(flank), mean_len=500, min_len=90, max_len=1800,
                            n_rate=0.004, paralog_rate=0.02)
        names = cl[0].names
        opts = {"klength": int(k), "canon": True, "consider_missing": False, "patfilt": True, "maf": 0.01,
                "multiple_files": False, "stroi": [names[1], names[len(names) // 2]]}
        recs = [c.record() for c in cl]
        # the first two clusters once more at the end of the run (other names): every pattern of theirs has been seen
        # by an earlier rank, which is what the exchange has to find out
        recs += [(r[0], r[1] + "_again", r[2]) for r in recs[:2]]
        return recs, names, opts
    from conftest import all_cases, case_records
    case = {c["name"]: c for c in all_cases()}[spec]
    return case_records(case), case["all_strains"], case["opts"]


class OracleShard:
    """PatternSource stand-in: the oracle's texts of this rank's clusters"""

    def __init__(self, records, start, opts):
        from oracle import oracle as po
        run = po.OracleRun(klength=opts["klength"], stroi=set(opts["stroi"]) if opts["stroi"] else (), canon=opts["canon"],
                           consider_missing=opts["consider_missing"], patfilt=opts["patfilt"], maf=opts["maf"])
        run.feed(records)
        self.kmers_tsv, self.kmers_to_hashes, hp = run.texts()
        self.rows = [r + "\n" for r in hp.split("\n") if r]
        self.start = start

    def export_patterns(self):
        md5 = np.zeros((len(self.rows), 16), dtype=np.uint8)
        for i, r in enumerate(self.rows):
            md5[i] = np.frombuffer(base64.b64decode(r[:24]), dtype=np.uint8)
        # local first-seen order inside the rank's ordinal range
        fs = (np.uint64(self.start) << np.uint64(32)) + np.arange(len(self.rows), dtype=np.uint64)
        return md5, fs

    def render_pattern_rows(self, ids):
        yield "".join(self.rows[int(i)] for i in ids).encode()


def main():
    mode, rank, world, port, outdir, spec = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5], sys.argv[6]
    compress = len(sys.argv) > 7 and sys.argv[7] == "1"
    method = sys.argv[8] if len(sys.argv) > 8 else "owner"
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = port
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    d = dist if world > 1 else None
    if spec.startswith("files:"):
        from panfeed_amd import sharded
        _, csvp, gffdir, k, up, down, tg = spec.split(":")
        stats = sharded.run_files_sharded(csvp, gffdir, outdir, rank, world, d, None, klength=int(k), upstream=int(up),
                                          downstream=int(down), targets=tuple(t for t in tg.split(",") if t),
                                          compress=compress, batch_clusters=5, method=method)
        stats.pop("log", None)
        if d:
            d.barrier()
            d.destroy_process_group()
        print("STATS " + json.dumps(stats), flush=True)
        return
    records, strains, opts = load_case(spec)
    from panfeed_amd import sharded
    from panfeed_amd.distributed import shard_range
    if mode == "oracle":
        start, stop = shard_range(len(records), rank, world)
        src = OracleShard(records[start:stop], start, opts)
        w = sharded.ShardWriter(outdir, rank, compress)
        w.write_batch(src.kmers_tsv, src.kmers_to_hashes)
        rows, n_global = sharded.finish_shard(src, w, d, None, method)
        w.close()
        if d:
            d.barrier()
        if rank == 0:
            sharded.assemble(outdir, world, strains, compress)
        stats = {"rank": rank, "pattern_rows": rows, "patterns": n_global, "range": [start, stop]}
    else:
        stats = sharded.run_records_sharded(records, outdir, strains, rank, world, d, None, klength=opts["klength"],
                                            canon=opts["canon"], consider_missing=opts["consider_missing"],
                                            patfilt=opts["patfilt"], maf=opts["maf"], targets=opts["stroi"] or (),
                                            compress=compress, multiple_files=opts["multiple_files"], batch_clusters=3,
                                            method=method)
    if d:
        d.barrier()
        d.destroy_process_group()
    print("STATS " + json.dumps(stats), flush=True)


if __name__ == "__main__":
    main()
