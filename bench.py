#!/usr/bin/env python3
"""bench.py -- throughput of the panfeed hot path (k-mer extraction + pattern hashing) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of cluster_cutter + pattern_hasher (device path: pf_submit) over the
whole synthetic pangenome, packed input already resident in HBM, results left in HBM.
Workload at every N: BASELINE.json configs[2] -- 50k clusters x 1k samples, k=31, +-100 bp
flanks (SURVEY 8d generator, pure-ACGT variant) -- PER GPU ("weak" scaling: gene clusters are
independent, each rank owns a contiguous range; the only exchange is the all-gather of
(md5, first_seen) pairs for the run-global pattern dedup, inside the timed region).

Prints ONE JSON line on rank 0.  `value` = k-mer instances/s over all ranks (trip count of
/root/reference/panfeed/panfeed.py:64); `patterns_per_s` = unique patterns/s (rows of
hashes_to_patterns.tsv).  `roofline` prices the dominant kernel (kmer_scan_kernel) against HBM
with SURVEY 8(d)'s algorithmic bytes; `cpu_baseline` is the CPU oracle (a port, not the
reference) on a bounded sample of the same workload, rank 0 at N=1 only.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8 TB/s HBM3E peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--clusters", type=int, default=50000, help="gene clusters per GPU")
    ap.add_argument("--samples", type=int, default=1000)
    ap.add_argument("--flank", type=int, default=100)
    ap.add_argument("--k", type=int, default=31)
    ap.add_argument("--max-items", type=int, default=65536)
    ap.add_argument("--cpu-clusters", type=int, default=0, help="clusters in the CPU-baseline sample (0 = auto)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dedup", action="store_true", help="scan every copy of identical sequences (PF_FLAG_NO_DEDUP)")
    ap.add_argument("--no-every-copy-leg", action="store_true", help="skip the extra scan-every-copy step")
    return ap.parse_args()


def algorithmic_bytes(packed_bytes, n_kept, n_new_patterns, S, k):
    """sum ceil(len/4) + U*(Kb+16) + U*W8 + P_new*W8."""
    kb = 8 if k <= 32 else 16
    w8 = (S + 7) // 8
    return packed_bytes + n_kept * (kb + 16) + n_kept * w8 + n_new_patterns * w8


def cpu_baseline(args, threads):
    """the CPU oracle on the first clusters of the same workload, ~10-30 s of CPU work"""
    from oracle import oracle as po
    from panfeed_amd import synth
    n = args.cpu_clusters or max(8, min(args.clusters, 80 * threads))   # ~15-25 s at ~5e6 instances/s/thread
    cl = synth.generate(n, args.samples, first=0, flank=args.flank, n_rate=0.0)
    recs = [c.record() for c in cl]
    ninst = sum(c.n_instances(args.k) for c in cl)
    run = po.OracleRun(klength=args.k, want_kmers_tsv=False, threads=threads)
    prep = run.prepare(recs)
    t0 = time.time()
    run.run_prepared(prep)
    dt = time.time() - t0
    st = run.stats()
    assert st["instances"] == ninst
    return {"value": ninst / dt, "unit": "kmer_instances/s", "cores": threads, "kind": "port",
            "sample": f"first {n} clusters of the workload ({ninst} instances, {st['patterns']} patterns) in {dt:.1f} s",
            "patterns_per_s": st["patterns"] / dt}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    dist = None
    # rehearsal mode for a 1-GPU box: every rank on cuda:0, gloo collectives on host tensors (control flow only)
    shared = os.environ.get("PANFEED_BENCH_SHARED_GPU") == "1"
    if shared:
        local = 0
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist
        if shared:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    dev = torch.device("cpu") if shared else torch.device("cuda", local)

    from panfeed_amd import devbatch, synth
    from panfeed_amd.distributed import merge_patterns
    from panfeed_amd.engine import Engine

    S, k = args.samples, args.k
    first = rank * args.clusters
    t_gen = time.time()
    eng = Engine(klength=k, max_strains=(S + 31) // 32 * 32, device=local, max_items=args.max_items,
                 pattern_capacity=1 << 25, dedup=not args.no_dedup)
    # generate + upload in slabs so the host never holds more than a slab of cluster objects
    slab = 50000
    dbs = []
    for s0 in range(0, args.clusters, slab):
        cl = synth.generate(min(slab, args.clusters - s0), S, first=first + s0, flank=args.flank, n_rate=0.0)
        dbs.append(devbatch.from_synth(eng, cl, k, first_ordinal=first + s0))
        del cl
    t_gen = time.time() - t_gen
    n_inst = sum(d.n_instances for d in dbs)
    packed_bytes = sum(d.packed_bytes for d in dbs)

    def step(eng=eng):
        from panfeed_amd import _lib
        _lib.check(eng.L.pf_reset_patterns(eng.ctx))
        tot = {"kept": 0, "new": 0, "scan_ms": 0.0, "rows_ms": 0.0, "emit_ms": 0.0, "total_ms": 0.0, "dedup_ms": 0.0,
               "patrows_ms": 0.0, "md5_ms": 0.0, "finish_ms": 0.0,
               "launches": 0, "items": 0, "retried": 0, "unique": 0, "dedup_clusters": 0, "scan_bytes": 0}
        for d in dbs:
            res = d.submit(eng)
            tm = eng.timing()
            tot["dedup_ms"] += tm["dedup_ms"]
            tot["patrows_ms"] += tm["patrows_ms"]
            tot["md5_ms"] += tm["md5_ms"]
            tot["finish_ms"] += tm["finish_ms"]
            tot["dedup_clusters"] += tm["n_dedup_clusters"]
            tot["scan_bytes"] += tm["scan_packed_bytes"]
            tot["kept"] += int(res.n_kept)
            tot["new"] += int(res.n_new_patterns)
            tot["unique"] += int(res.n_unique)
            for a, b in (("scan_ms", "scan_ms"), ("rows_ms", "rows_ms"), ("emit_ms", "emit_ms"), ("total_ms", "total_ms")):
                tot[a] += tm[b]
            tot["launches"] += tm["scan_launches"]
            tot["items"] += tm["n_items"]
            tot["retried"] += tm["n_retried"]
        n_global = tot["new"]
        if world > 1:
            n_global = merge_patterns(eng, dist, dev)
        tot["global_patterns"] = n_global
        return tot

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.time()
    last = None
    for _ in range(args.steps):
        last = step()
    fence()
    dt = time.time() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    agg = torch.tensor([float(n_inst), float(last["kept"]), float(packed_bytes)], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(agg, op=dist.ReduceOp.SUM)
    dt = float(tmax.item())
    tot_inst, tot_kept, tot_packed = (float(x) for x in agg.tolist())

    # transparency leg (untimed for `value`): the same pass with the identical-sequence shortcut off
    every = None
    if world == 1 and not args.no_dedup and not args.no_every_copy_leg:
        eng2 = Engine(klength=k, max_strains=(S + 31) // 32 * 32, device=local, max_items=min(args.max_items, 8192),
                      pattern_capacity=1 << 25, dedup=False)
        step(eng2)                       # untimed: first-use allocations of the second context
        torch.cuda.synchronize()
        t1 = time.time()
        e = step(eng2)
        torch.cuda.synchronize()
        dt1 = time.time() - t1
        assert (e["kept"], e["new"], e["unique"]) == (last["kept"], last["new"], last["unique"]), "dedup changed the result"
        every = {"value": n_inst / dt1, "unit": "kmer_instances/s", "ms_per_step": dt1 * 1e3, "scan_ms": e["scan_ms"],
                 "rows_ms": e["rows_ms"], "emit_ms": e["emit_ms"], "clusters_repartitioned": e["retried"],
                 "note": "PF_FLAG_NO_DEDUP: every copy of every sequence scanned; same outputs"}
        eng2.close()

    if rank == 0:
        ms_step = dt / args.steps * 1e3
        alg = algorithmic_bytes(packed_bytes, last["kept"], last["new"], S, k)   # this rank, one step
        kern_ms = {"cluster_dedup_kernel": last["dedup_ms"], "kmer_scan_kernel": last["scan_ms"],
                   "rows_kernel": last["rows_ms"], "emit_kernel": last["emit_ms"],
                   "pattern_rows_kernel": last["patrows_ms"], "md5_kernel": last["md5_ms"],
                   "finish_kernel": last["finish_ms"]}
        # The path is a chain of kernels over the same clusters (dedup -> scan -> finish -> md5), none of which moves
        # all of the algorithmic bytes on its own, so the roofline is taken over the chain: SURVEY 8d's algorithmic
        # bytes of the clusters one pf_submit processes / the device time of the chain, first kernel's start to last
        # kernel's end (HIP events on the library's stream; idle gaps between its kernels count against it).  Per
        # kernel: its own summed duration and the HBM bytes the PMC pass measured for it.
        chain_ms = last["total_ms"]            # first kernel's start to last kernel's end (HIP events), gaps included
        chain_s = chain_ms / 1e3
        achieved = alg / chain_s / 1e9 if chain_s > 0 else 0.0
        dom = max(kern_ms, key=kern_ms.get)
        # HBM traffic from the committed PMC summary of this same command (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
        # separate passes, gfx950 x2 correction on the read side); per step = per pf_submit chain
        traffic, traffic_src, per_kernel = None, None, {kn: {"ms": v} for kn, v in kern_ms.items() if v > 0}
        pmc_path = os.path.join(REPO, "profiles", "r01", "final_pmc_traffic.json")
        default_cmd = (args.clusters, S, k, args.flank, world, args.no_dedup) == (50000, 1000, 31, 100, 1, False)
        if default_cmd and os.path.exists(pmc_path):
            with open(pmc_path) as fh:
                pmc = json.load(fh)
            tot = 0.0
            for kn in per_kernel:
                hb = sum(v["hbm_bytes_per_step"] for name, v in pmc["kernels"].items() if name.startswith("pf::" + kn))
                if hb:
                    per_kernel[kn]["hbm_bytes"] = hb
                    per_kernel[kn]["hbm_GBps"] = hb / (per_kernel[kn]["ms"] / 1e3) / 1e9
                    tot += hb
            if tot:
                traffic = tot
                traffic_src = "profiles/r01/final_pmc_traffic.json"
        out = {
            "metric": "k-mer instances/s (+ unique patterns/s), k=31, 50k clusters x 1k samples",
            "value": tot_inst * args.steps / dt,
            "unit": "kmer_instances/s",
            "patterns_per_s": last["global_patterns"] * args.steps / dt,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": f"synthetic {args.clusters} clusters x {S} samples per GPU, k={k}, "
                                   f"+-{args.flank} bp flanks, canonical, maf 0.01 (BASELINE.json configs[2], pure-ACGT)",
                       "clusters_per_gpu": args.clusters, "samples": S, "k": k, "flank": args.flank,
                       "instances_per_gpu": n_inst, "packed_bytes_per_gpu": packed_bytes,
                       "unique_kmers": last["unique"], "kept_kmers": last["kept"], "patterns": last["global_patterns"],
                       "sharding": f"{world} x contiguous cluster ranges" + (", RCCL all-gather of pattern digests" if world > 1 else "")},
            "roofline": {"bound": "hbm", "kernel": "pf_submit kernel chain (dedup+scan+finish+md5)", "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": traffic_src, "algorithmic_bytes_per_step": alg,
                         "kernel_ms_per_step": chain_ms, "longest_kernel": dom, "per_kernel": per_kernel,
                         "kernels_ms_summed": sum(kern_ms.values()),
                         "note": "algorithmic bytes (SURVEY 8d) of the clusters one pf_submit processes / device time of "
                                 "its kernel chain (first start to last end, HIP events on the library's stream); "
                                 "traffic = HBM bytes of the same kernels per step (PMC)"},
            "device_ms_per_step": dict(kern_ms, submit_total=last["total_ms"]),
            "work_items": last["items"], "clusters_repartitioned": last["retried"],
            "clusters_deduplicated": last["dedup_clusters"], "scan_packed_bytes": last["scan_bytes"],
            "scan_every_copy": every,
            "setup_s": {"generate_and_upload": t_gen},
        }
        if world == 1 and not args.no_cpu_baseline:
            threads = min(len(os.sched_getaffinity(0)), 16)   # a 1-GPU box's CPU share
            out["cpu_baseline"] = cpu_baseline(args, threads)
        print(json.dumps(out), flush=True)
    for d in dbs:
        d.free()
    eng.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
