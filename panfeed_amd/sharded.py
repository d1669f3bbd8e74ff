"""A whole run on N GPUs (one process per GPU) whose files equal the single-GPU / `--cores 1` files byte for byte.

What the reference does with one writer process (/root/reference/panfeed/__main__.py:67-81: `pattern_hasher` owns the
run-global `patterns` set and the three file handles, panfeed.py:146-150, 179-187, 210-226) is split like this:

* rank r processes the contiguous range `shard_range(...)` of the processing order with the GLOBAL cluster ordinals
  (`Engine.next_ordinal` = start of its range), so `first_seen = ordinal << 32 | rank inside the cluster` is unique
  over the run and monotone in the `--cores 1` order;
* `kmers_to_hashes.tsv` / `kmers.tsv` bodies: per rank, as the batches finish -> `<file>.part<r>`;
* `hashes_to_patterns.tsv`: nothing is written while the shard runs (`defer_patterns`); at the end the ranks exchange
  {md5, first_seen} (`distributed.merge_pattern_tensors`: RCCL on the GPUs, gloo in rehearsals), every rank keeps the
  patterns whose `first_seen` is the run-wide minimum of their digest, sorts them by `first_seen` and renders those rows
  (`pf_render_pattern_rows`) -> `hashes_to_patterns.tsv.part<r>`.  All `first_seen` of rank r precede all of rank
  r + 1, so the parts concatenated in rank order are the reference's first-seen order;
* rank 0 writes header + parts in rank order into the final files (`assemble`).  Under `compress` every part is a
  sequence of gzip members and so is the concatenation.
* `--multiple-files`: the pattern set restarts in every cluster (panfeed.py:165): no exchange at all, every rank writes
  its clusters' directories.

`PatternSource` is the little a rank needs from its engine after the shard has run; `Engine` provides it, and the CPU
tests drive the same merge / assembly code with a stand-in.
"""
import os
import shutil

import numpy as np

from .engine import KMERS_TO_HASHES_HEADER, KMERS_TSV_HEADER, OwnedText, hashes_to_patterns_header

FILES = ("kmers.tsv", "kmers_to_hashes.tsv", "hashes_to_patterns.tsv")


def _part_path(output, name, rank, compress):
    return os.path.join(output, ".parts", f"{name}{'.gz' if compress else ''}.part{rank:04d}")


class _PlainPart:
    def __init__(self, path):
        self.fh = open(path, "wb")

    def write(self, data):
        if isinstance(data, str):
            data = data.encode()
        if len(data):
            self.fh.write(data)

    def close(self):
        self.fh.close()


class ShardWriter:
    """The part files of one rank (bodies only; headers are rank 0's business in `assemble`)."""

    def __init__(self, output, rank, compress=False):
        self.output, self.rank, self.compress = output, rank, compress
        os.makedirs(os.path.join(output, ".parts"), exist_ok=True)
        self.handles = {}
        for name in FILES:
            path = _part_path(output, name, rank, compress)
            if compress:
                from .output import ParallelGzipWriter
                h = ParallelGzipWriter(path, compresslevel=9)
                h.wrote = True            # an empty part adds nothing (no empty gzip member in the middle of the file)
                self.handles[name] = h
            else:
                self.handles[name] = _PlainPart(path)
        self.bytes = 0

    def write_batch(self, kmers_tsv, kmers_to_hashes):
        if isinstance(kmers_tsv, OwnedText):     # engine.OwnedText: the library's block, written where it lies
            try:
                self.handles["kmers.tsv"].write(kmers_tsv.view)
            finally:
                n_kt = len(kmers_tsv)
                kmers_tsv.release()
        else:
            self.handles["kmers.tsv"].write(kmers_tsv)
            n_kt = len(kmers_tsv)
        self.handles["kmers_to_hashes.tsv"].write(kmers_to_hashes)
        self.bytes += n_kt + len(kmers_to_hashes)

    def write_patterns(self, blocks):
        for b in blocks:
            self.handles["hashes_to_patterns.tsv"].write(b)
            self.bytes += len(b)

    def close(self):
        for h in self.handles.values():
            h.close()


def kept_pattern_ids(md5, first_seen, dist=None, engine=None, device=None, method="owner"):
    """ids (into this rank's pool) of the patterns this rank has to write, in first-seen order, and the number of
    run-global patterns.  md5: uint8 [n,16], first_seen: uint64 [n] (numpy)."""
    import torch

    from .distributed import merge_pattern_tensors
    t_md5 = torch.from_numpy(np.ascontiguousarray(md5))
    t_fs = torch.from_numpy(np.ascontiguousarray(first_seen).view(np.int64))
    if device is not None and device.type == "cuda":
        t_md5, t_fs = t_md5.to(device), t_fs.to(device)
    keep, n_global = merge_pattern_tensors(t_md5, t_fs, dist, engine=engine if t_md5.is_cuda else None, method=method)
    keep = keep.cpu().numpy().astype(bool)
    ids = np.flatnonzero(keep)
    ids = ids[np.argsort(np.asarray(first_seen)[ids], kind="stable")]
    return ids.astype(np.uint32), int(n_global)


def check_first_seen_disjoint(first_seen, dist, device=None):
    """Ranks must not share cluster ordinals (every rank running with ordinals from 0 would make ties that keep a
    digest twice): the ordinal ranges [min, max] of the ranks have to be disjoint and ascending with the rank.
    `device`: this rank's GPU -- under RCCL the gathered tensors have to live on it (every rank on cuda:0 is a
    duplicate-GPU communicator error or a hang)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return
    import torch
    fs = np.asarray(first_seen, dtype=np.uint64)
    lo = int(fs.min() >> np.uint64(32)) if len(fs) else -1
    hi = int(fs.max() >> np.uint64(32)) if len(fs) else -1
    mine = torch.tensor([lo, hi], dtype=torch.int64)
    allr = [torch.zeros(2, dtype=torch.int64) for _ in range(dist.get_world_size())]
    backend = dist.get_backend()
    if backend == "nccl":
        dev = device if device is not None and device.type == "cuda" else torch.device("cuda", torch.cuda.current_device())
        allr = [t.to(dev) for t in allr]
        mine = mine.to(dev)
    dist.all_gather(allr, mine)
    prev = -1
    for r, t in enumerate(allr):
        a, b = (int(x) for x in t.tolist())
        if a < 0:
            continue
        if a <= prev:
            raise RuntimeError(f"rank {r} holds cluster ordinals {a}..{b} overlapping an earlier rank's (<= {prev}): "
                               "every rank of a sharded run must number its clusters with the run's global ordinals")
        prev = b


def finish_shard(source, writer, dist=None, device=None, method="owner"):
    """After the shard's batches: exchange digests, write this rank's pattern rows.  `source`: export_patterns() and
    render_pattern_rows(ids) (an Engine).  Returns (rows written by this rank, run-global pattern count)."""
    md5, fs = source.export_patterns()
    check_first_seen_disjoint(fs, dist, device)
    eng = source if hasattr(source, "ctx") else None
    ids, n_global = kept_pattern_ids(md5, fs, dist, engine=eng, device=device, method=method)
    writer.write_patterns(source.render_pattern_rows(ids))
    return len(ids), n_global


def assemble(output, world, strains, compress=False, keep_parts=False):
    """Rank 0, after every rank has closed its parts: header + parts in rank order -> the three files."""
    headers = {"kmers.tsv": KMERS_TSV_HEADER, "kmers_to_hashes.tsv": KMERS_TO_HASHES_HEADER,
               "hashes_to_patterns.tsv": hashes_to_patterns_header(strains)}
    for name in FILES:
        final = os.path.join(output, name + (".gz" if compress else ""))
        with open(final, "wb") as out:
            if compress:
                import gzip
                out.write(gzip.compress(headers[name].encode(), compresslevel=9))
            else:
                out.write(headers[name].encode())
            for r in range(world):
                with open(_part_path(output, name, r, compress), "rb") as part:
                    shutil.copyfileobj(part, out, 16 << 20)
    if not keep_parts:
        shutil.rmtree(os.path.join(output, ".parts"), ignore_errors=True)


def _barrier(dist, device=None):
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        if device is not None and device.type == "cuda" and dist.get_backend() == "nccl":
            dist.barrier(device_ids=[device.index if device.index is not None else 0])
        else:
            dist.barrier()


def _bind_device(device):
    """one process per GPU: collectives of this process run on `device` (RCCL picks torch's current device)"""
    if device is not None and device.type == "cuda":
        import torch
        torch.cuda.set_device(device)


def run_records_sharded(records, output, strains, rank, world, dist=None, device=None, klength=31, canon=True,
                        consider_missing=False, patfilt=True, maf=0.01, targets=(), compress=False,
                        multiple_files=False, batch_clusters=256, weights=None, method="owner", gpu=0,
                        max_items=0, pattern_capacity=0, dedup=True):
    """One rank of a sharded run over in-memory reference-shaped records (the tuples input.py:468 yields; every rank
    is given the same list).  Returns a dict of counters; the files are complete once rank 0 returns."""
    from .distributed import shard_range
    from .engine import Engine
    _bind_device(device)
    records = list(records)
    start, stop = shard_range(len(records), rank, world, weights)
    max_strains = max([len(strains)] + [max(len(r[0]), len(r[2])) for r in records] + [1])
    eng = Engine(klength=klength, canon=canon, consider_missing=consider_missing, patfilt=patfilt, maf=maf,
                 multiple_files=multiple_files, max_strains=(max_strains + 31) // 32 * 32, stroi=set(targets) if targets else (),
                 device=gpu, max_items=max_items, pattern_capacity=pattern_capacity, dedup=dedup)
    try:
        eng.next_ordinal = start                          # global ordinals: first_seen is unique over the run
        batches = eng.run_stream(records[start:stop], batch_clusters=batch_clusters, defer_patterns=not multiple_files)
        return _drive(eng, batches, output, strains, rank, world, dist, device, compress, multiple_files, method,
                      (start, stop))
    finally:
        eng.close()


def run_files_sharded(presence_absence, gffdir, output, rank, world, dist=None, device=None, fastadir=None, klength=31,
                      canon=True, consider_missing=False, patfilt=True, maf=0.01, upstream=0, downstream=0,
                      downstream_start_codon=False, targets=(), genes=None, compress=False, multiple_files=False,
                      batch_clusters=256, resident=True, device_text=True, method="owner", gpu=0, max_items=0,
                      pattern_capacity=0):
    """`pipeline.run_files` on N GPUs: every rank opens the pangenome, takes its range of the processing order
    (balanced by the number of gene entries per table row) and writes its parts; rank 0 assembles.  Option names as
    in the reference's CLI (`__main__.py:86-186`)."""
    from .distributed import shard_range
    from .engine import Engine
    from .native_input import Pangenome
    _bind_device(device)
    targets = tuple(targets or ())
    exists = os.path.isdir(output)
    _barrier(dist, device)                                        # every rank has looked before rank 0 creates it
    if exists:                                            # input.py:213-216
        raise FileExistsError(f"Output directory {output} already exists; remove it or change --output")
    if rank == 0:
        os.makedirs(output)
    _barrier(dist, device)
    def make_engine(n_strains):
        return Engine(klength=klength, canon=canon, consider_missing=consider_missing, patfilt=patfilt, maf=maf,
                      multiple_files=multiple_files, max_strains=max(32, (n_strains + 31) // 32 * 32),
                      stroi=set(targets), device=gpu,
                      # the same rule as pipeline.run_files (a cluster that needs more makes the library re-make its scratch)
                      max_items=max_items or max(512, 2 * int(batch_clusters)), pattern_capacity=pattern_capacity)
    # every rank reads every genome (a cluster's sequences come from all of them): with resident genomes the files go to
    # the rank's GPU as they are read (pf_pangenome_open_device), as in pipeline.run_files; the context is made first, from
    # the table's header line
    from .pipeline import _peek_n_strains
    eng = pg = None
    n_peek = _peek_n_strains(presence_absence) if resident else 0
    if n_peek:
        eng = make_engine(n_peek)
        try:
            pg = Pangenome(presence_absence, gffdir, fastadir, upstream, downstream, downstream_start_codon, targets=targets,
                           genes=genes, engine=eng)
        except BaseException:
            eng.close()
            raise
        if eng.max_strains < pg.n_strains:               # (the header was not what the reader made of it)
            pg.close()
            eng.close()
            eng = pg = None
    if pg is None:
        pg = Pangenome(presence_absence, gffdir, fastadir, upstream, downstream, downstream_start_codon, targets=targets,
                       genes=genes)
    try:
        w = pg.weights()
        start, stop = shard_range(len(w), rank, world, w)
        pg.set_range(start, stop - start)
        if eng is None:
            eng = make_engine(pg.n_strains)
        if resident and not pg.resident:
            pg.make_resident(eng)
        eng.next_ordinal = start
        batches = eng.run_pangenome(pg, batch_clusters=batch_clusters, device_text=device_text,
                                    defer_patterns=not multiple_files)
        stats = _drive(eng, batches, output, pg.strains, rank, world, dist, device, compress, multiple_files, method,
                       (start, stop))
        stats["log"] = pg.take_log()
        return stats
    finally:
        pg.close()
        if eng is not None:
            eng.close()


def _drive(eng, batches, output, strains, rank, world, dist, device, compress, multiple_files, method, rng):
    from .output import create_hash_files, create_kmer_stroi, write_headers
    stats = {"rank": rank, "clusters": 0, "instances": 0, "kept_kmers": 0, "range": list(rng), "device_ms": 0.0}

    class _Cols:
        columns = list(strains)

    if multiple_files:
        for o in batches:
            for idx, kt, kh, hp in o.per_cluster:
                path = os.path.join(output, idx)                      # panfeed.py:38-43, 159-167
                os.makedirs(path, exist_ok=True)
                ks = create_kmer_stroi(path, compress)
                ks.write(kt)
                ks.close()
                f_hp, f_kh = create_hash_files(path, compress)
                write_headers(f_hp, f_kh, _Cols)
                f_hp.write(hp)
                f_kh.write(kh)
                f_hp.close()
                f_kh.close()
            stats["clusters"] += o.stats.get("clusters", 0)
            stats["instances"] += o.stats.get("instances", 0)
        _barrier(dist, device)
        return stats
    writer = ShardWriter(output, rank, compress)
    try:
        for o in batches:
            writer.write_batch(o.kmers_tsv, o.kmers_to_hashes)
            stats["clusters"] += o.stats.get("clusters", 0)
            stats["instances"] += o.stats.get("instances", 0)
            stats["kept_kmers"] += o.stats.get("kept_kmers", 0)
            stats["device_ms"] += o.timing.get("total_ms", 0.0)
        stats["local_patterns"] = eng.pattern_count()
        stats["pattern_rows"], stats["patterns"] = finish_shard(eng, writer, dist, device, method)
    finally:
        writer.close()
    _barrier(dist, device)
    if rank == 0:
        assemble(output, world, strains, compress)
    _barrier(dist, device)
    stats["bytes"] = writer.bytes
    return stats
