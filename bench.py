#!/usr/bin/env python3
"""bench.py -- throughput of the panfeed hot path (k-mer extraction + pattern hashing) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`--gpus N` with N > 1 and no WORLD_SIZE in the environment launches the N ranks itself (torch.distributed.run as a
child process; this parent never touches HIP) and fails loudly when the box has fewer than N devices.

One "step" = one pass of cluster_cutter + pattern_hasher (device path: pf_submit) over the rank's synthetic
pangenome, packed input already resident in HBM, results left in HBM, followed (N > 1) by the exchange of
{md5, first_seen} for the run-global pattern dedup (inside the timed region).

Workload: BASELINE.json configs[2] -- 50k clusters x 1k samples, k=31, +-100 bp flanks (SURVEY 8d generator,
pure-ACGT variant).  N = 1: that workload on the one GPU.  N > 1 (default): BASELINE configs[3] as written, the SAME
50k clusters sharded over the N GPUs with the RCCL pattern merge -- "strong" scaling, `value`; the weak figure (50k
clusters PER GPU) is measured too and reported under `weak_scaling`.  `--scaling weak` makes the weak figure `value`.
`--samples 5000 --k 21|51` is BASELINE configs[4]'s shape; `--targets-clusters M` adds its --targets second pass
(positional rows of kmers.tsv for M clusters with every sample a target strain), timed on its own.
`--sweep-alleles` runs a 5 000-cluster pass for several numbers of distinct sequences per cluster.

Prints ONE JSON line on rank 0.  `value` = k-mer instances/s over all ranks (trip count of
/root/reference/panfeed/panfeed.py:64); `patterns_per_s` = unique patterns/s (rows of hashes_to_patterns.tsv).
`roofline` prices the chain of kernels one pf_submit runs (dedup -> scan -> finish -> md5) against HBM with SURVEY
8(d)'s algorithmic bytes, and lists every kernel with the roof that actually bounds it; `cpu_baseline` is the CPU
oracle (a port, not the reference) on a bounded sample of the same workload, rank 0 at N=1 only; `parity_sample` is
that sample's kmers_to_hashes / hashes_to_patterns text, oracle against GPU, byte for byte; `with_N` is the same
workload with SURVEY 8d's share of 'N's (0.1 % of the sequences); `end_to_end` is files on disk -> the three files
(pipeline.run_files) and Seqinfo records -> the three files, with the stage split.
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

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8 TB/s HBM3E peak (6.3 TB/s measured achievable)
# vector issue: 256 CUs x 4 SIMDs, one wave64 VALU instruction per 2 cycles per SIMD, 2.4 GHz (the guide's figure; on
# this chip only the plain two-operand integer operations and v_bitop3 issue at that rate)
VALU_PEAK_WAVE_INSTS = 256 * 4 * 0.5 * 2.4e9
# what a SIMD sustains on the integer mix these kernels are made of (v_add3 / v_alignbit / v_lshl_add / v_cmp / v_mad / 64-bit
# shifts: 4 cycles; MD5's bitop3-add3-alignbit-add chain: 4.05 cycles per instruction with one to eight waves and one to
# four chains per wave) -- measured, tools/micro/valu_chain.hip + valu_ops.hip, profiles/r02/valu_*.txt
INT_MIX_PEAK_WAVE_INSTS = 256 * 4 * 2.4e9 / 4.05


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks (one per GPU); default 1, or WORLD_SIZE under a launcher")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--clusters", type=int, default=50000, help="gene clusters of the workload (per GPU when scaling is weak)")
    ap.add_argument("--scaling", default="auto", choices=["auto", "weak", "strong"],
                    help="auto: strong at N > 1 (BASELINE configs[3]: the same --clusters split over the GPUs, the weak "
                         "figure as a sub-key); weak: --clusters per GPU")
    ap.add_argument("--total-clusters", type=int, default=0, help="strong scaling over this many clusters (same as "
                                                                   "--scaling strong --clusters T)")
    ap.add_argument("--samples", type=int, default=1000)
    ap.add_argument("--flank", type=int, default=100)
    ap.add_argument("--k", type=int, default=31)
    ap.add_argument("--mean-alleles", type=float, default=7.0)
    ap.add_argument("--allele-decay", type=float, default=0.5, help="allele weights decay^i (1.0: uniform)")
    ap.add_argument("--allele-model", default="star", choices=["star", "tree"],
                    help="star: SURVEY 8d's alleles (default); tree: alleles that descend from one another (synth.py)")
    ap.add_argument("--n-rate", type=float, default=0.0, help="share of the sequences that carry one 'N' (SURVEY 8d: 0.001)")
    ap.add_argument("--max-items", type=int, default=65536)
    ap.add_argument("--pattern-capacity", type=int, default=0,
                    help="initial slots of the run-global pattern table (0 = by workload: 2^25 for the headline's 2.6 M patterns "
                         "per pass, 2^22 for small workloads; the table grows on demand either way)")
    ap.add_argument("--cpu-clusters", type=int, default=0, help="clusters in the CPU-baseline sample (0 = auto)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dedup", action="store_true", help="scan every copy of identical sequences (PF_FLAG_NO_DEDUP)")
    ap.add_argument("--no-unit-dedup", action="store_true", help="scan every unit of every distinct sequence (PF_FLAG_NO_UNIT_DEDUP)")
    ap.add_argument("--no-key-binning", action="store_true", help="key partitions walk the whole cluster (PF_FLAG_NO_KEY_BINNING)")
    ap.add_argument("--device-plan", action="store_true", help="the simple clusters' work items laid out on the device (PF_FLAG_DEVICE_PLAN)")
    ap.add_argument("--no-every-copy-leg", action="store_true", help="skip the extra scan-every-copy step")
    ap.add_argument("--no-n-leg", action="store_true", help="skip the leg with SURVEY 8d's share of 'N's")
    ap.add_argument("--no-e2e-leg", action="store_true", help="skip the end-to-end (files -> files) leg")
    ap.add_argument("--no-weak-leg", action="store_true", help="N > 1, strong scaling: skip the weak-scaling sub-leg")
    ap.add_argument("--e2e-clusters", type=int, default=1000, help="clusters of the end-to-end leg's on-disk pangenome")
    ap.add_argument("--e2e-long-clusters", type=int, default=8000, help="clusters of the long files -> files run (0: skip)")
    ap.add_argument("--e2e-records-clusters", type=int, default=400, help="clusters of the records -> files figure")
    ap.add_argument("--targets-clusters", type=int, default=0, help="clusters of the --targets second pass leg")
    ap.add_argument("--sweep-alleles", action="store_true", help="extra leg: throughput vs distinct sequences per cluster")
    ap.add_argument("--merge-method", default="owner", choices=["owner", "allgather"])
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------ launcher
def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: N fresh ranks through torch.distributed.run, as a child process.
    This parent has made no HIP call (torch.cuda.device_count() does not initialise the GPU on this image) and
    makes none: it passes the children's output through and exits with their code."""
    import socket
    import subprocess

    import torch
    shared = os.environ.get("PANFEED_BENCH_SHARED_GPU") == "1"
    ndev = torch.cuda.device_count()
    if ndev < n and not shared:
        sys.stderr.write(f"bench.py: {n} ranks requested, {ndev} device{'s' if ndev != 1 else ''} on this box -- one rank "
                         f"per GPU; nothing was run.  (PANFEED_BENCH_SHARED_GPU=1 rehearses the control flow with every "
                         f"rank on cuda:0 and gloo collectives; its numbers are not multi-GPU numbers.)\n")
        sys.exit(2)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.exit(subprocess.run(cmd, env=env).returncode)


def algorithmic_bytes(packed_bytes, n_kept, n_new_patterns, S, k):
    """sum ceil(len/4) + U*(Kb+16) + U*W8 + P_new*W8; also the part of it this design must move
    (the U*W8 bit-row term is never materialised: a k-mer carries a pattern id instead)."""
    kb = 8 if k <= 32 else 16
    w8 = (S + 7) // 8
    full = packed_bytes + n_kept * (kb + 16) + n_kept * w8 + n_new_patterns * w8
    return full, full - n_kept * w8


def synth_kw(args, n_rate=None):
    return dict(flank=args.flank, n_rate=args.n_rate if n_rate is None else n_rate, mean_alleles=args.mean_alleles,
                allele_decay=args.allele_decay, allele_model=args.allele_model)


def oracle_sample(args, local, threads, n, n_rate, timed):
    """The CPU oracle on the first `n` clusters of the workload (oracle/: the checker), and the same clusters through the
    HIP path: kmers_to_hashes.tsv and hashes_to_patterns.tsv bodies compared byte for byte.  Returns (cpu_baseline dict
    or None, parity dict)."""
    from oracle import oracle as po
    from panfeed_amd import devbatch, synth
    from panfeed_amd.engine import Engine
    S, k = args.samples, args.k
    cl = synth.generate(n, S, first=0, **synth_kw(args, n_rate))
    recs = [c.record() for c in cl]
    ninst = sum(c.n_instances(k) for c in cl)
    run = po.OracleRun(klength=k, want_kmers_tsv=False, threads=threads)
    prep = run.prepare(recs)
    t0 = time.time()
    run.run_prepared(prep)
    dt = time.time() - t0
    st = run.stats()
    assert st["instances"] == ninst
    del prep, recs
    base = None
    if timed:
        base = {"value": ninst / dt, "unit": "kmer_instances/s", "cores": threads, "kind": "port",
                "sample": f"first {n} clusters of the workload ({ninst} instances, {st['patterns']} patterns) in {dt:.1f} s",
                "patterns_per_s": st["patterns"] / dt}
    ekh, ehp = run.text(1).encode(), run.text(2).encode()
    run.close()
    eng = Engine(klength=k, max_strains=(S + 31) // 32 * 32, device=local, max_items=min(args.max_items, 8192),
                 dedup=not args.no_dedup)
    db = devbatch.from_synth(eng, cl, k)
    res = db.submit(eng)
    kh, hp = eng.render_device(db)
    eq_kh, eq_hp = bytes(kh) == ekh, bytes(hp) == ehp
    par = {"clusters": n, "equal": bool(eq_kh and eq_hp), "kmers_to_hashes_equal": bool(eq_kh),
           "hashes_to_patterns_equal": bool(eq_hp), "kmers_to_hashes_bytes": len(ekh), "hashes_to_patterns_bytes": len(ehp),
           "kept_kmers": int(res.n_kept), "patterns": int(res.n_new_patterns), "n_rate": n_rate,
           "note": "oracle (CPU restatement of panfeed.py:16-235) vs the HIP path on the first clusters of this workload: "
                   "the bodies of kmers_to_hashes.tsv and hashes_to_patterns.tsv, byte for byte"}
    db.free()
    eng.close()
    if not par["equal"]:
        raise AssertionError(f"GPU text differs from the oracle's on the first {n} clusters: {par}")
    return base, par


def kernel_source_hash():
    """sha256 (first 16 hex digits) of the kernel sources the library is built from: what a committed counter summary
    is compared with (the GPU box has no .git; tools/pmc_summary.py stamps the same hash into the summary)"""
    import hashlib
    h = hashlib.sha256()
    for f in ("pf_kernels.h", "pf_api.hip"):
        with open(os.path.join(REPO, "panfeed_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def load_pmc():
    """committed rocprofv3 --pmc summary of the default command (tools/profile_round.sh): HBM bytes and SQ counters.
    Only a summary whose kernel sources are THIS build's is used (kernel_source_sha16, stamped by tools/pmc_summary.py):
    counters of other kernels are not this run's traffic.  Returns (summary or None, its path, path of the newest
    summary that was passed over as stale or None)."""
    import glob
    paths = sorted(glob.glob(os.path.join(REPO, "profiles", "r[0-9]*", "final_pmc_summary.json")), reverse=True)
    want, stale = kernel_source_hash(), None
    for path in paths:
        with open(path) as fh:
            pmc = json.load(fh)
        if pmc.get("kernel_source_sha16") == want:
            return pmc, os.path.relpath(path, REPO), stale
        stale = stale or os.path.relpath(path, REPO)
    return None, None, stale


# which roof bounds which kernel, and why (DESIGN.md sections 4 and 6): the dedup pass streams the packed input (HBM);
# MD5 is a chain of dependent integer ops (vector issue); the scan waits on LDS round trips at the four waves per SIMD
# its whole-LDS table leaves it, the finish kernels on dependent global loads / atomics (latency)
KERNEL_BOUND = {"cluster_dedup_kernel": "hbm", "kmer_scan_kernel": "lds-latency", "md5_kernel": "valu",
                "finish_kernel": "latency", "rows_kernel": "latency", "emit_kernel": "latency",
                "pattern_rows_kernel": "latency"}


def make_step(eng, dbs, world, dist, dev, method):
    import torch
    from panfeed_amd import _lib
    from panfeed_amd.distributed import merge_patterns

    def step(eng=eng, checksum=False):
        _lib.check(eng.L.pf_reset_patterns(eng.ctx))
        sums = []
        tot = {"kept": 0, "new": 0, "scan_ms": 0.0, "rows_ms": 0.0, "emit_ms": 0.0, "total_ms": 0.0, "dedup_ms": 0.0,
               "patrows_ms": 0.0, "md5_ms": 0.0, "finish_ms": 0.0, "merge_ms": 0.0,
               "launches": 0, "items": 0, "retried": 0, "unique": 0, "dedup_clusters": 0, "scan_bytes": 0, "binned": 0}
        for d in dbs:
            res = d.submit(eng)
            tm = eng.timing()
            for key in ("dedup_ms", "patrows_ms", "md5_ms", "finish_ms", "scan_ms", "rows_ms", "emit_ms", "total_ms"):
                tot[key] += tm[key]
            tot["dedup_clusters"] += tm["n_dedup_clusters"]
            tot["scan_bytes"] += tm["scan_packed_bytes"]
            tot["kept"] += int(res.n_kept)
            tot["new"] += int(res.n_new_patterns)
            tot["unique"] += int(res.n_unique)
            tot["launches"] += tm["scan_launches"]
            tot["items"] += tm["n_items"]
            tot["retried"] += tm["n_retried"]
            tot["binned"] += tm["n_binned_clusters"]
            if checksum:                 # untimed legs only: what the files of this batch would hold, row for row
                sums.append(eng.result_checksum())
        tot["checksums"] = sums
        n_global = tot["new"]
        if world > 1:
            t0 = time.time()
            n_global = merge_patterns(eng, dist, dev, method=method)
            if dev.type == "cuda":
                torch.cuda.synchronize()
            tot["merge_ms"] = (time.time() - t0) * 1e3
        tot["global_patterns"] = n_global
        return tot
    return step


def targets_pass(args, local):
    """BASELINE configs[4]'s second pass: `--targets` with every sample a target strain on a few clusters: one row of
    kmers.tsv per k-mer instance (panfeed.py:90-107).  Host strings -> the kmers.tsv text, timed end to end and split."""
    from panfeed_amd import synth
    from panfeed_amd.engine import Engine
    from panfeed_amd.packing import build_batch_native
    S, k = args.samples, args.k
    cl = synth.generate(args.targets_clusters, S, first=10 ** 6, flank=args.flank, n_rate=0.0,
                        mean_alleles=args.mean_alleles, allele_decay=args.allele_decay, allele_model=args.allele_model)
    recs = [c.record() for c in cl]
    stroi = set(cl[0].names)
    eng = Engine(klength=k, max_strains=(S + 31) // 32 * 32, stroi=stroi, device=local)
    t0 = time.time()
    hb = build_batch_native(recs, k, True, eng.W, stroi=stroi, first_ordinal=0)
    t_pack = time.time() - t0
    t0 = time.time()
    eng.submit_host_batch(hb)
    t_submit = time.time() - t0
    # the rows: written by the GPU (kt_text_kernel), every block of the text through pinned host memory -- what a writer
    # thread would hand to its file
    t0 = time.time()
    dt = eng.render_targets_device(hb)
    nbytes = 0
    for blk in dt.chunks():
        nbytes += len(blk)
    t_render = time.time() - t0
    split = dict(eng.render_targets_timing)
    # untimed: the same text again for the row count and a CRC, against the host renderer's (pf_render_kmers_tsv)
    import zlib
    rows, crc_dev = 0, 0
    for blk in dt.chunks():
        arr = np.frombuffer(blk, dtype=np.uint8)
        rows += int(np.count_nonzero(arr == 10))
        crc_dev = zlib.crc32(blk, crc_dev)
    t0 = time.time()
    text = eng._render_targets(hb, hb.targets, owned=True)
    t_host = time.time() - t0
    host_split = dict(eng.render_targets_timing)
    crc_host = 0
    for a in range(0, len(text), 1 << 28):
        crc_host = zlib.crc32(text.view[a:a + (1 << 28)], crc_host)
    same = crc_host == crc_dev and len(text) == nbytes
    text.release()
    eng.close()
    if not same:
        raise SystemExit("targets pass: the device-written kmers.tsv differs from the host renderer's")
    return {"clusters": args.targets_clusters, "target_strains": S, "rows": rows, "bytes": nbytes,
            "pack_s": t_pack, "submit_s": t_submit, "marshal_s": split["marshal_s"],
            "device_render_and_copy_out_s": t_render - split["marshal_s"],
            "render_s": t_render,
            "rows_per_s_end_to_end": rows / (t_pack + t_submit + t_render),
            "GBps_text_out": nbytes / (t_render - split["marshal_s"]) / 1e9 if t_render > split["marshal_s"] else None,
            "equals_host_renderer": same,
            "host_renderer": {"render_s": t_host, "marshal_s": host_split["marshal_s"], "library_render_s": host_split["render_s"],
                              "rows_per_s_end_to_end": rows / (t_pack + t_submit + t_host)},
            "note": "kmers.tsv rows for every sample of the clusters (all strains are targets), host strings in, text out: "
                    "pack_s = records -> packed batch (pf_pack_records), submit_s = pf_submit (strand bits stay on the "
                    "device), render_s = marshalling of the per-sequence fields + kt_len / kt_text kernels + every block "
                    "of the text copied to pinned host memory (pf_device_text_chunk); equals_host_renderer = same length "
                    "and CRC-32 as pf_render_kmers_tsv's text (host_renderer: round 4's path, timed after)"}


def allele_sweep(args, local, allele_model="star"):
    """throughput against the number of distinct sequences per cluster (5 000 clusters x --samples, uniform allele
    weights): where the identical-sequence shortcut changes regime.  "star" = SURVEY 8d's alleles (each with its own
    random flanks and its own 1 % substitutions: the distinct k-mers of a cluster grow by ~480 per allele); "tree" =
    alleles descending from one another by two substitutions at a time (~62 new k-mers per allele).  `ms` is the best of
    three submits of one context (the second and third know the pangenome: the key-partition estimate is learned),
    `ms_first_submit` the first, which still has to find out (and pays first-use allocations)"""
    import torch
    from panfeed_amd import _lib, devbatch, synth
    from panfeed_amd.engine import Engine
    S, k = args.samples, args.k
    rows = []
    n = 5000
    for mean_alleles in (7, 30, 60, 70, 150, 500):
        eng = Engine(klength=k, max_strains=(S + 31) // 32 * 32, device=local, max_items=32768, pattern_capacity=1 << 24,
                     unit_dedup=not args.no_unit_dedup, key_binning=not args.no_key_binning,
                     device_plan=args.device_plan)
        cl = synth.generate(n, S, first=0, flank=args.flank, n_rate=0.0, mean_alleles=mean_alleles, allele_decay=1.0,
                            allele_model=allele_model)
        distinct = float(np.mean([len(np.unique(c.seq_allele)) for c in cl]))
        db = devbatch.from_synth(eng, cl, k)
        del cl
        best = first = None
        for rep in range(3):
            _lib.check(eng.L.pf_reset_patterns(eng.ctx))
            torch.cuda.synchronize()
            t0 = time.time()
            res = db.submit(eng)
            torch.cuda.synchronize()
            dt = time.time() - t0
            best = dt if best is None or dt < best else best
            first = dt if first is None else first
        tm = eng.timing()
        rows.append({"mean_alleles": mean_alleles, "distinct_per_cluster": distinct, "instances": db.n_instances,
                     "value": db.n_instances / best, "ms": best * 1e3, "ms_first_submit": first * 1e3, "clusters_mode1_or_2": tm["n_dedup_clusters"],
                     "clusters_mode0": n - tm["n_dedup_clusters"], "scan_ms": tm["scan_ms"], "dedup_ms": tm["dedup_ms"],
                     "finish_ms": tm["finish_ms"], "rows_ms": tm["rows_ms"], "emit_ms": tm["emit_ms"],
                     "patrows_ms": tm["patrows_ms"], "md5_ms": tm["md5_ms"], "kept": int(res.n_kept),
                     "patterns": int(res.n_new_patterns), "repartitioned": tm["n_retried"], "clusters_key_binned": tm["n_binned_clusters"]})
        db.free()
        eng.close()
    return rows


# ------------------------------------------------------------------------------------------------ end to end
def end_to_end(args, local):
    """What a user of the drop-in runs (the reference's whole run: __main__.py:299-356): files on disk -> the three
    files through pipeline.run_files (native reader -> genomes resident in HBM -> batches -> kernels -> text -> writer
    thread), with the text of the two big files rendered by the host threads ("plain") and written on the device
    ("device_text"); and Seqinfo records (the tuples input.py:468 yields, in host memory) -> the three files through
    Engine.run_stream.  The synthetic pangenome (SURVEY 8d's generator with its share of 'N's, +-flank) is written to a
    scratch directory first, untimed; its files are read back from the page cache."""
    import shutil
    import tempfile

    from panfeed_amd import synth
    from panfeed_amd.engine import (KMERS_TO_HASHES_HEADER, KMERS_TSV_HEADER, Engine, OwnedText,
                                    hashes_to_patterns_header)
    from panfeed_amd.pipeline import run_files
    S, k, up = args.samples, args.k, args.flank
    n = args.e2e_clusters
    out = {"shape": f"{n} clusters x {S} samples, k={k}, --upstream {up} --downstream {up}, SURVEY 8d generator with "
                    f"0.1 % of the sequences carrying an 'N' (the shape of BASELINE configs[2], a subset of its clusters)"}
    scratch = tempfile.mkdtemp(prefix="pf_bench_e2e_")
    try:
        t0 = time.time()
        cl = synth.generate(n, S, first=0, flank=up, n_rate=0.001, mean_alleles=args.mean_alleles,
                            allele_decay=args.allele_decay, allele_model=args.allele_model)
        ninst = sum(c.n_instances(k) for c in cl)
        csvp, gffs, _fas = synth.write_pangenome(scratch, cl, missing_gene_rate=0.0)
        in_bytes = sum(os.path.getsize(p) for p in gffs.values()) + os.path.getsize(csvp)
        out["setup_write_inputs_s"] = time.time() - t0
        out["input_bytes"] = in_bytes
        del cl
        for name, device_text in (("files_to_files_plain", False), ("files_to_files_device_text", True)):
            od = os.path.join(scratch, "out_" + name)
            t0 = time.time()
            st = run_files(csvp, os.path.join(scratch, "gffs"), od, klength=k, upstream=up, downstream=up,
                           batch_clusters=256, device_text=device_text, device=local)
            dt = time.time() - t0
            assert st["instances"] == ninst, (st["instances"], ninst)
            fbytes = sum(os.path.getsize(os.path.join(od, f)) for f in os.listdir(od))
            out[name] = {"seconds": dt, "inst_per_s": ninst / dt, "output_bytes": fbytes, "output_GBps": fbytes / dt / 1e9,
                         "input_GBps": in_bytes / dt / 1e9, "instances": ninst, "kept_kmers": st["kept_kmers"],
                         "patterns": st["patterns"], "stages_s": {a: round(b, 4) for a, b in st["stages"].items()}}
            shutil.rmtree(od)
        out["stages_note"] = ("open_parse_s: table + GFFs + FASTA read and indexed; context_s: pf_create; genome_upload_s: "
                              "contigs -> 2 bit/base in HBM; then per batch, overlapped: pack_busy_s (reader + packer "
                              "thread; pack_wait_s = what of it the GPU thread waited for), submit_s (coordinates + "
                              "literals H2D, gather, the kernel chain; device_ms of it on the device), text_s (device "
                              "text + D2H, or pf_fetch + host renderers), write_busy_s (writer thread); total_s: open to the last byte written; "
                              "close_reader_s: the reader handed to its teardown thread (its memory goes back in the "
                              "background, after the call), close_context_s: pf_destroy; seconds = the whole call")
    finally:
        shutil.rmtree(scratch, ignore_errors=True)
    # ---- records -> files: the boundary as the reference's callers use it (host strings in, three files out)
    n2 = args.e2e_records_clusters
    cl = synth.generate(n2, S, first=0, flank=up, n_rate=0.001, mean_alleles=args.mean_alleles,
                        allele_decay=args.allele_decay, allele_model=args.allele_model)
    names = cl[0].names
    recs = [c.record() for c in cl]
    ninst = sum(c.n_instances(k) for c in cl)
    del cl
    stroi = {names[10 % len(names)]}
    scratch = tempfile.mkdtemp(prefix="pf_bench_e2e_")
    try:
        for name, device_text in (("records_to_files_plain", False), ("records_to_files_device_text", True)):
            eng = Engine(klength=k, max_strains=(S + 31) // 32 * 32, stroi=stroi, device=local)
            t0 = time.time()
            nbytes = 0
            with open(os.path.join(scratch, "kmers.tsv"), "wb") as ks, open(os.path.join(scratch, "kmers_to_hashes.tsv"), "wb") as kh, \
                    open(os.path.join(scratch, "hashes_to_patterns.tsv"), "wb") as hp:
                ks.write(KMERS_TSV_HEADER.encode())
                kh.write(KMERS_TO_HASHES_HEADER.encode())
                hp.write(hashes_to_patterns_header(names).encode())
                for o in eng.run_stream(iter(recs), batch_clusters=128, device_text=device_text):
                    for fh, data in ((ks, o.kmers_tsv), (kh, o.kmers_to_hashes), (hp, o.hashes_to_patterns)):
                        if isinstance(data, OwnedText):
                            fh.write(data.view)
                            nbytes += len(data)
                            data.release()
                        else:
                            nbytes += fh.write(data.encode() if isinstance(data, str) else data)
            dt = time.time() - t0
            out[name] = {"clusters": n2, "seconds": dt, "inst_per_s": ninst / dt, "output_bytes": nbytes,
                         "output_GBps": nbytes / dt / 1e9, "instances": ninst,
                         "stages_s": {a: round(b, 4) for a, b in eng.stages.items()}}
            eng.close()
    finally:
        shutil.rmtree(scratch, ignore_errors=True)
    out["records_note"] = ("Seqinfo records (Python strings, one target strain: kmers.tsv rows included) -> pf_pack_records -> "
                           "H2D -> kernels -> text -> three files; the records are made before the clock starts")
    # ---- the same files -> files run, long enough for the start-up (opening 1 000 GFFs, the context, the genomes' upload) to be
    # a small share: what a whole pangenome runs at.  Fewer clusters when the scratch space does not hold the inputs.
    n_long = args.e2e_long_clusters
    while n_long >= 1000:
        scratch = tempfile.mkdtemp(prefix="pf_bench_e2e_long_")
        try:
            free = shutil.disk_usage(scratch).free
            if free < n_long * 1.3e6 * 1.6:        # 1.21 MB of GFF3 + FASTA per cluster in, 0.34 MB of TSV out, and room to spare
                out.setdefault("long_leg_fallbacks", []).append(f"{n_long} clusters need {n_long * 1.3e6 * 1.6 / 1e9:.1f} GB of scratch, "
                                                                f"{free / 1e9:.1f} GB free")
                n_long //= 2
                continue
            t0 = time.time()
            cl = synth.generate(n_long, S, first=0, flank=up, n_rate=0.001, mean_alleles=args.mean_alleles,
                                allele_decay=args.allele_decay, allele_model=args.allele_model)
            ninst = sum(c.n_instances(k) for c in cl)
            csvp, gffs, _fas = synth.write_pangenome(scratch, cl, missing_gene_rate=0.0, workers=min(16, os.cpu_count() or 1))
            in_bytes = sum(os.path.getsize(p) for p in gffs.values()) + os.path.getsize(csvp)
            setup = time.time() - t0
            del cl
            od = os.path.join(scratch, "out")
            # The run twice, both reported: the first one of a process pays for things a pangenome run pays once per
            # MACHINE rather than once per run (the input files' pages entering this process's NUMA neighbourhood, the first
            # use of the ingest kernels and of page-locked registration: 2.0 - 2.4 s against 0.9 - 1.2 s in every A/B of
            # round 5); `seconds` is the second run, `first_run_seconds` the first.  The reader of a run gives its memory
            # back on a thread of its own and the next open waits for it: joined (2 s) before the clock starts.
            dts = []
            for rep in range(2):
                if rep:
                    shutil.rmtree(od, ignore_errors=True)
                    time.sleep(2.0)
                t0 = time.time()
                st = run_files(csvp, os.path.join(scratch, "gffs"), od, klength=k, upstream=up, downstream=up, batch_clusters=256,
                               device_text=True, device=local)
                dts.append(time.time() - t0)
                if rep == 0:
                    first_stages = {a: round(b, 4) for a, b in st["stages"].items()}
            dt = dts[-1]
            assert st["instances"] == ninst, (st["instances"], ninst)
            fbytes = sum(os.path.getsize(os.path.join(od, f)) for f in os.listdir(od))
            stg = st["stages"]
            waits = {"the reader + packer thread (pack_wait_s)": stg.get("pack_wait_s", 0.0),
                     "the genomes' upload (first_submit_wait_s)": stg.get("first_submit_wait_s", 0.0),
                     "pf_submit itself (submit_s: coordinates up, gather, kernels)": stg.get("submit_s", 0.0),
                     "the text stage (text_s: device text + its copy to the host)": stg.get("text_s", 0.0)}
            out["files_to_files_device_text_long"] = {
                "clusters": n_long, "setup_write_inputs_s": setup, "input_bytes": in_bytes, "seconds": dt, "inst_per_s": ninst / dt,
                "first_run_seconds": dts[0], "first_run_inst_per_s": ninst / dts[0], "first_run_stages_s": first_stages,
                "output_bytes": fbytes, "output_GBps": fbytes / dt / 1e9, "input_GBps": in_bytes / dt / 1e9, "instances": ninst,
                "kept_kmers": st["kept_kmers"], "patterns": st["patterns"],
                "gpu_busy_share": stg.get("device_ms", 0.0) / 1e3 / dt,
                "gpu_thread_waits_longest_on": max(waits, key=waits.get),
                "start_up_share": (stg.get("open_parse_s", 0.0) + stg.get("context_wait_s", 0.0) + stg.get("first_submit_wait_s", 0.0)) / dt,
                "stages_s": {a: round(b, 4) for a, b in stg.items()}}
            break
        except OSError as e:                       # no room after all: half the clusters
            out.setdefault("long_leg_fallbacks", []).append(f"{n_long} clusters: {e}")
            n_long //= 2
        finally:
            shutil.rmtree(scratch, ignore_errors=True)
    return out


def start_heartbeat(period=60.0):
    """one line on stderr every minute while the run lasts: the oracle sample, the synthetic input of a 25 000 x 5 000
    workload and the CRC of a 37 GB text each take minutes in silence, and a silent run reads as a hung one"""
    import threading
    t_begin = time.time()

    def beat():
        while True:
            time.sleep(period)
            sys.stderr.write(f"bench.py: running, {time.time() - t_begin:.0f} s\n")
            sys.stderr.flush()
    threading.Thread(target=beat, name="bench-heartbeat", daemon=True).start()


def main():
    args = parse()
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and (args.gpus or 1) > 1:
        launch_ranks(args.gpus)                       # does not return
    rank = int(os.environ.get("RANK", "0"))
    world = int(env_world or "1")
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if rank == 0:
        start_heartbeat()
    if args.gpus is not None and args.gpus != world:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s); nothing was run\n")
        sys.exit(2)
    import torch
    dist = None
    # rehearsal mode for a 1-GPU box: every rank on cuda:0, gloo collectives on host tensors (control flow only)
    shared = os.environ.get("PANFEED_BENCH_SHARED_GPU") == "1"
    if shared:
        local = 0
    elif torch.cuda.device_count() < max(world, local + 1):
        sys.stderr.write(f"bench.py: {world} ranks requested, {torch.cuda.device_count()} device(s) on this box -- one rank "
                         f"per GPU; nothing was run\n")
        sys.exit(2)
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist
        if shared:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    dev = torch.device("cpu") if shared else torch.device("cuda", local)
    backend = dist.get_backend() if world > 1 else None
    n_ranks_seen = dist.get_world_size() if world > 1 else 1      # the ranks the collective backend actually has

    from panfeed_amd import devbatch, synth
    from panfeed_amd.distributed import shard_range
    from panfeed_amd.engine import Engine

    S, k = args.samples, args.k
    if args.total_clusters > 0:
        args.scaling, args.clusters = "strong", args.total_clusters
    strong = args.scaling == "strong" or (args.scaling == "auto" and world > 1)
    # (rehearsal with every rank on one GPU: the ranks share its memory, a work item's scratch slice is ~1.9 MB)
    max_items = args.max_items if not shared else max(2048, args.max_items // (4 * world))
    if strong:
        first, stop = shard_range(args.clusters, rank, world)
        n_mine = stop - first
    else:
        first, n_mine = rank * args.clusters, args.clusters
    if not args.pattern_capacity:
        # (every step starts from an empty pattern set -- pf_reset_patterns, inside the timed region: clearing a table sized
        # for the headline costs 0.16 ms, a tenth of a 5 000 x 200 pass)
        args.pattern_capacity = 1 << 25 if n_mine * max(S, 1) >= 20_000_000 else 1 << 22
    eng = Engine(klength=k, max_strains=(S + 31) // 32 * 32, device=local, max_items=max_items,
                 pattern_capacity=args.pattern_capacity, dedup=not args.no_dedup, unit_dedup=not args.no_unit_dedup, key_binning=not args.no_key_binning,
                 device_plan=args.device_plan)

    def build(first, n_mine, n_rate=None):
        # generate + upload in slabs so the host never holds more than a slab of cluster objects
        slab = max(1, 50000 * 1000 // max(S, 1))
        out = []
        for s0 in range(0, n_mine, slab):
            cl = synth.generate(min(slab, n_mine - s0), S, first=first + s0, **synth_kw(args, n_rate))
            out.append(devbatch.from_synth(eng, cl, k, first_ordinal=first + s0))
            del cl
        return out

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(step, steps, warmup):
        for _ in range(warmup):
            step()
        fence()
        t0 = time.time()
        last = None
        for _ in range(steps):
            last = step()
        fence()
        return time.time() - t0, last

    def reduce_max_sum(maxes, sums):
        tmax = torch.tensor(maxes, dtype=torch.float64, device=dev)
        agg = torch.tensor(sums, dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dist.all_reduce(agg, op=dist.ReduceOp.SUM)
        return [float(x) for x in tmax.tolist()], [float(x) for x in agg.tolist()]

    t_gen = time.time()
    dbs = build(first, n_mine)
    t_gen = time.time() - t_gen
    n_inst = sum(d.n_instances for d in dbs)
    packed_bytes = sum(d.packed_bytes for d in dbs)
    step = make_step(eng, dbs, world, dist, dev, args.merge_method)
    dt, last = timed(step, args.steps, args.warmup)
    (dt, max_kernel_ms, max_merge_ms), (tot_inst, tot_kept, tot_packed) = reduce_max_sum(
        [dt, last["total_ms"], last["merge_ms"]], [float(n_inst), float(last["kept"]), float(packed_bytes)])

    # N > 1, strong scaling: the weak figure beside it (--clusters per GPU, rank r owns [r * clusters, (r + 1) * clusters))
    weak = None
    if world > 1 and strong and not args.no_weak_leg:
        for d in dbs:
            d.free()
        dbs = []
        wdbs = build(rank * args.clusters, args.clusters)
        wstep = make_step(eng, wdbs, world, dist, dev, args.merge_method)
        wdt, wlast = timed(wstep, args.steps, args.warmup)
        (wdt, wk, wm), (winst,) = reduce_max_sum([wdt, wlast["total_ms"], wlast["merge_ms"]],
                                                  [float(sum(d.n_instances for d in wdbs))])
        weak = {"value": winst * args.steps / wdt, "unit": "kmer_instances/s", "scaling": "weak",
                "ms_per_step": wdt / args.steps * 1e3, "clusters_per_gpu": args.clusters,
                "patterns_per_s": wlast["global_patterns"] * args.steps / wdt,
                "max_kernel_chain_ms": wk, "max_merge_ms": wm}
        for d in wdbs:
            d.free()

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
        # full-size parity property: both paths must write the same rows in the same places -- a device-side checksum over
        # every (cluster, position, k-mer key, pattern digest) and every cluster row (pf_result_checksum), batch by batch
        sum_on, sum_off = step(eng, checksum=True)["checksums"], step(eng2, checksum=True)["checksums"]
        assert sum_on == sum_off and sum(c[2] for c in sum_on) == last["kept"], "dedup changed the rows"
        every = {"value": n_inst / dt1, "unit": "kmer_instances/s", "ms_per_step": dt1 * 1e3, "scan_ms": e["scan_ms"],
                 "rows_ms": e["rows_ms"], "emit_ms": e["emit_ms"], "clusters_repartitioned": e["retried"],
                 "rows_checksum": [f"{x:016x}" for x in (sum(c[0] for c in sum_on) & (2 ** 64 - 1), sum(c[1] for c in sum_on) & (2 ** 64 - 1))],
                 "note": "PF_FLAG_NO_DEDUP: every copy of every sequence scanned; same outputs (counts and the device-side "
                         "checksum over every k-mer row and cluster row, pf_result_checksum, equal to the timed path's)"}
        eng2.close()

    default_shape = (args.clusters, S, k, args.flank, args.no_dedup, args.mean_alleles, args.allele_decay,
                     args.allele_model) == (50000, 1000, 31, 100, False, 7.0, 0.5, "star")
    # SURVEY 8d's generator proper: 0.1 % of the sequences carry one 'N' (the value above is its pure-ACGT variant)
    with_n = None
    if world == 1 and default_shape and args.n_rate == 0.0 and not args.no_n_leg:
        for d in dbs:
            d.free()
        dbs = []
        ndbs = build(first, n_mine, n_rate=0.001)
        nstep = make_step(eng, ndbs, world, dist, dev, args.merge_method)
        ndt, nlast = timed(nstep, args.steps, args.warmup)
        ninst_n = sum(d.n_instances for d in ndbs)
        with_n = {"value": ninst_n * args.steps / ndt, "unit": "kmer_instances/s", "ms_per_step": ndt / args.steps * 1e3,
                  "n_rate": 0.001, "instances": ninst_n, "slow_path_rows": sum(d.n_extra for d in ndbs),
                  "kept_kmers": nlast["kept"], "patterns": nlast["global_patterns"],
                  "patterns_per_s": nlast["global_patterns"] * args.steps / ndt,
                  "device_ms_per_step": {"cluster_dedup_kernel": nlast["dedup_ms"], "kmer_scan_kernel": nlast["scan_ms"],
                                         "finish_kernel": nlast["finish_ms"], "md5_kernel": nlast["md5_ms"],
                                         "rows_kernel": nlast["rows_ms"], "emit_kernel": nlast["emit_ms"],
                                         "submit_total": nlast["total_ms"]},
                  "note": "the same workload from SURVEY 8d's generator as written: 0.1 % of the sequences carry one 'N' "
                          "(their windows around it are slow-path rows, panfeed.py:65-79 string semantics)"}
        for d in ndbs:
            d.free()

    if rank == 0:
        ms_step = dt / args.steps * 1e3
        alg, alg_must = algorithmic_bytes(packed_bytes, last["kept"], last["new"], S, k)   # this rank, one step
        kern_ms = {"cluster_dedup_kernel": last["dedup_ms"], "kmer_scan_kernel": last["scan_ms"],
                   "rows_kernel": last["rows_ms"], "emit_kernel": last["emit_ms"],
                   "pattern_rows_kernel": last["patrows_ms"], "md5_kernel": last["md5_ms"],
                   "finish_kernel": last["finish_ms"]}
        # The path is a chain of kernels over the same clusters (dedup -> scan -> finish -> md5), none of which moves
        # all of the algorithmic bytes on its own, so the roofline is taken over the chain: SURVEY 8d's algorithmic
        # bytes of the clusters one pf_submit processes / the device time of the chain, first kernel's start to last
        # kernel's end (HIP events on the library's stream; idle gaps between its kernels count against it).
        chain_ms = last["total_ms"]
        chain_s = chain_ms / 1e3
        achieved = alg / chain_s / 1e9 if chain_s > 0 else 0.0
        dom = max(kern_ms, key=kern_ms.get)
        # Per kernel: its own time, the roof that bounds it, and -- for the default command -- what the committed PMC
        # passes measured (HBM bytes; SQ instruction / wait counters): tools/profile_round.sh, tools/pmc_summary.py
        traffic, traffic_src = None, None
        per_kernel = {kn: {"ms": v, "bound": KERNEL_BOUND.get(kn, "latency")} for kn, v in kern_ms.items() if v > 0}
        if "cluster_dedup_kernel" in per_kernel:
            pk = per_kernel["cluster_dedup_kernel"]
            pk["algorithmic_GBps"] = packed_bytes / (pk["ms"] / 1e3) / 1e9     # it has to read the packed input once
            pk["frac_of_hbm_peak"] = pk["algorithmic_GBps"] / HBM_PEAK_GBS
        default_cmd = default_shape and world == 1 and not strong and args.n_rate == 0.0
        pmc, pmc_rel, pmc_stale = load_pmc() if default_cmd else (None, None, None)
        if pmc:
            tot = 0.0
            for name, v in pmc["kernels"].items():
                if "synth_expand_kernel" in name:                # builds the synthetic input once, before the timed steps
                    continue
                tot += v.get("hbm_bytes_per_step", 0.0)          # every pf:: kernel of a step, fills included
            for kn, pk in per_kernel.items():
                sel = [v for name, v in pmc["kernels"].items() if name.startswith("pf::" + kn)]
                hb = sum(v.get("hbm_bytes_per_step", 0.0) for v in sel)
                if hb:
                    pk["hbm_bytes"] = hb
                    pk["hbm_GBps"] = hb / (pk["ms"] / 1e3) / 1e9
                    pk["hbm_frac_of_peak"] = pk["hbm_GBps"] / HBM_PEAK_GBS
                valu = sum(v.get("SQ_INSTS_VALU_per_step", 0.0) for v in sel)
                if valu:
                    pk["valu_wave_insts"] = valu
                    pk["valu_frac_of_issue_peak"] = valu / (pk["ms"] / 1e3) / VALU_PEAK_WAVE_INSTS
                    pk["valu_frac_of_measured_integer_issue_rate"] = valu / (pk["ms"] / 1e3) / INT_MIX_PEAK_WAVE_INSTS
                wc = sum(v.get("SQ_WAVE_CYCLES_per_step", 0.0) for v in sel)
                if wc:
                    pk["wave_cycles_waiting_frac"] = sum(v.get("SQ_WAIT_ANY_per_step", 0.0) for v in sel) / wc
                    pk["wave_cycles_issue_stalled_frac"] = sum(v.get("SQ_WAIT_INST_ANY_per_step", 0.0) for v in sel) / wc
            if tot:
                traffic, traffic_src = tot, pmc_rel
        workload = (f"synthetic {n_mine} clusters x {S} samples on this GPU"
                    + (f" ({args.clusters} over {world} GPUs)" if strong and world > 1 else (" per GPU" if world > 1 else ""))
                    + f", k={k}, +-{args.flank} bp flanks, canonical, maf 0.01")
        if args.n_rate:
            workload += f", {args.n_rate:g} of the sequences with one 'N'"
        if default_shape and world == 1 and args.n_rate == 0.0:
            workload += " (BASELINE.json configs[2], pure-ACGT)"
        elif default_shape and world == 1 and args.n_rate == 0.001:
            workload += " (BASELINE.json configs[2] with SURVEY 8d's share of 'N's)"
        elif default_shape and strong:
            workload += " (BASELINE.json configs[3]: configs[2] sharded)"
        elif S == 5000:
            workload += " (BASELINE.json configs[4] shape)"
        out = {
            "metric": "k-mer instances/s (+ unique patterns/s), k=31, 50k clusters x 1k samples",
            "value": tot_inst * args.steps / dt,
            "unit": "kmer_instances/s",
            "patterns_per_s": last["global_patterns"] * args.steps / dt,
            "n_gpus": n_ranks_seen, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step, "higher_is_better": True,
            "scaling": "strong" if strong and world > 1 else "weak",
            "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": workload,
                       "clusters_this_gpu": n_mine, "samples": S, "k": k, "flank": args.flank,
                       "mean_alleles": args.mean_alleles, "allele_decay": args.allele_decay, "allele_model": args.allele_model,
                       "instances_this_gpu": n_inst, "packed_bytes_this_gpu": packed_bytes,
                       "unique_kmers": last["unique"], "kept_kmers": last["kept"], "patterns": last["global_patterns"],
                       "sharding": f"{world} x contiguous cluster ranges" + (
                           f", pattern digests merged by {'RCCL' if not shared else 'gloo'} "
                           + ("all-to-all to the digest's owner rank and back" if args.merge_method == "owner" else "all-gather")
                           if world > 1 else "")},
            "roofline": {"bound": "hbm", "kernel": "pf_submit kernel chain (dedup+scan+finish+md5)", "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": traffic_src,
                         # the counters are a committed pass, not this run's: which commit's kernels it profiled, and whether
                         # the kernel sources of THIS build are those (false: re-collect with tools/profile_round.sh)
                         "traffic_commit": (pmc or {}).get("commit"),
                         "traffic_matches_this_build": True if pmc else None,
                         # a committed summary of OTHER kernel sources is not used: traffic stays null until
                         # tools/profile_round.sh has been run on this build
                         "traffic_stale_summary_ignored": pmc_stale,
                         "algorithmic_bytes_per_step": alg,
                         "must_move_bytes_per_step": alg_must,
                         "must_move_GBps": alg_must / chain_s / 1e9 if chain_s > 0 else 0.0,
                         "must_move_frac": alg_must / chain_s / 1e9 / HBM_PEAK_GBS if chain_s > 0 else 0.0,
                         "kernel_ms_per_step": chain_ms, "longest_kernel": dom, "per_kernel": per_kernel,
                         "kernels_ms_summed": sum(kern_ms.values()),
                         "note": "achieved = SURVEY 8d algorithmic bytes of the clusters one pf_submit processes / device "
                                 "time of its kernel chain (first start to last end, HIP events on the library's stream, "
                                 "measured live in this run). "
                                 "U*ceil(S/8) of those bytes (one bit row per kept k-mer) are never materialised by this "
                                 "design -- a k-mer carries a pattern id -- so must_move_* prices only the bytes the path "
                                 "has to move (packed input, key + pattern id per kept k-mer, one row per new pattern). "
                                 "Only cluster_dedup_kernel is HBM-bound; per_kernel gives each kernel its own roof: hbm = "
                                 "bytes it must read / time vs 8 TB/s (algorithmic_GBps, frac_of_hbm_peak), valu = "
                                 "SQ_INSTS_VALU / time vs the chip's vector issue rate, 256 CU x 4 SIMD x 1 "
                                 "wave-instruction / 2 cycles x 2.4 GHz (valu_frac_of_issue_peak), latency / lds-latency "
                                 "= share of wave cycles spent waiting, SQ_WAIT_ANY / SQ_WAVE_CYCLES "
                                 "(wave_cycles_waiting_frac).  traffic (and the per-kernel counter figures) are NOT "
                                 "measured by this run: they are the HBM bytes of every pf:: kernel of a step (FETCH_SIZE "
                                 "x 2 on the 16-byte-per-lane reads + WRITE_SIZE, fill and count kernels included) from the "
                                 "committed rocprofv3 --pmc passes of this same command (traffic_source)."},
            "device_ms_per_step": dict(kern_ms, submit_total=last["total_ms"]),
            "work_items": last["items"], "clusters_repartitioned": last["retried"],
            "clusters_deduplicated": last["dedup_clusters"], "clusters_key_binned": last["binned"],
            "scan_packed_bytes": last["scan_bytes"],
            "scan_every_copy": every,
            "setup_s": {"generate_and_upload": t_gen},
        }
        if with_n is not None:
            out["with_N"] = with_n
        if world > 1:
            out["multi_gpu"] = {"max_kernel_chain_ms": max_kernel_ms, "max_merge_ms": max_merge_ms,
                                "merge_method": args.merge_method, "backend": backend,
                                "ranks_in_process_group": n_ranks_seen,
                                "note": "slowest rank's device chain and host-timed digest exchange of the last step"}
            if weak is not None:
                out["weak_scaling"] = weak
        if args.targets_clusters:
            out["targets_second_pass"] = targets_pass(args, local)
        if args.sweep_alleles:
            out["allele_sweep"] = allele_sweep(args, local)
            out["allele_sweep_tree"] = allele_sweep(args, local, "tree")
        if world == 1 and not args.no_cpu_baseline:
            threads = min(len(os.sched_getaffinity(0)), 16)   # a 1-GPU box's CPU share
            per_cluster = max(1, S // 1000)
            n = args.cpu_clusters or max(8, min(args.clusters, 80 * threads // per_cluster))   # ~15-25 s at ~5e6 instances/s/thread
            out["cpu_baseline"], out["parity_sample"] = oracle_sample(args, local, threads, n, args.n_rate, True)
            if with_n is not None:
                _, with_n["parity_sample"] = oracle_sample(args, local, threads, max(8, n // 4), 0.001, False)
        if world == 1 and default_shape and not args.no_e2e_leg:
            try:
                out["end_to_end"] = end_to_end(args, local)
            except OSError as e:             # no room for the 1.2 GB scratch pangenome: the line is still worth printing
                out["end_to_end"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)
    for d in dbs:
        d.free()
    eng.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
