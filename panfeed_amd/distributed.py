"""Multi-GPU: gene clusters shard across ranks (one process per GPU, contiguous ranges of the
processing order); the only cross-rank state of the path is the run-global `patterns` set and its
first-seen rule (/root/reference/panfeed/panfeed.py:149-150, 179-180, 210-212).

Every rank dedups its own clusters on its GPU; at the end (or per super-batch) the ranks exchange
{md5 digest (16 B), first_seen (8 B)} of their patterns over RCCL and each rank keeps a pattern row iff its
first_seen is the minimum for that digest.  first_seen = (global cluster ordinal << 32 | rank inside the
cluster) is monotone in the reference's --cores 1 order, so the surviving rows, sorted by first_seen, are
exactly hashes_to_patterns.tsv.

Two forms of the exchange (same result):
  "owner"     (default) every digest has an owner rank (a few of its bits mod world); rows go to their owners with
              one all-to-all, the owner marks the first of every digest (device hash table), the marks go back
              with a second all-to-all of one byte per row.  A rank sends and receives its own volume once: on the
              point-to-point xGMI links that is 1/world of what an all-gather moves, and the merge kernel sees
              1/world of the rows.
  "allgather" every rank gathers every row and marks its own.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib


def shard_range(n_clusters, rank, world, weights=None):
    """contiguous [start, stop) of the processing order for `rank`; balanced by `weights`
    (k-mer instances per cluster) when given, else by count."""
    if weights is None:
        base, rem = divmod(n_clusters, world)
        start = rank * base + min(rank, rem)
        return start, start + base + (1 if rank < rem else 0)
    w = np.asarray(weights, dtype=np.float64)
    cum = np.concatenate(([0.0], np.cumsum(w)))
    total = cum[-1]
    cuts = [int(np.searchsorted(cum, total * r / world, side="left")) for r in range(world + 1)]
    cuts[0], cuts[-1] = 0, n_clusters
    for i in range(1, world + 1):
        cuts[i] = max(cuts[i], cuts[i - 1])
    return cuts[rank], cuts[rank + 1]


def _pack_rows(md5, first_seen):
    n = md5.shape[0]
    pay = torch.empty((n, 3), dtype=torch.int64, device=md5.device)
    if n:
        pay[:, :2] = md5.contiguous().view(torch.int64).view(n, 2)
        pay[:, 2] = first_seen
    return pay


def _first_per_digest(rows, engine=None):
    """rows: int64 [m,3] {md5 lo, md5 hi, first_seen}.  (keep bool [m]: the row holds the minimum first_seen of its
    digest, number of distinct digests).  On the GPU by the library's hash-table kernels, else a torch sort."""
    m = rows.shape[0]
    dev = rows.device
    if m == 0:
        return torch.zeros(0, dtype=torch.bool, device=dev), 0
    if engine is not None and rows.is_cuda:
        return _merge_on_device(engine, rows, 0, m)
    # lexicographic sort by (digest hi, digest lo): two stable passes
    order = torch.argsort(rows[:, 1], stable=True)
    order = order[torch.argsort(rows[order, 0], stable=True)]
    sp = rows[order]
    new_group = torch.ones(m, dtype=torch.bool, device=dev)
    new_group[1:] = (sp[1:, 0] != sp[:-1, 0]) | (sp[1:, 1] != sp[:-1, 1])
    gid = torch.cumsum(new_group.to(torch.int64), 0) - 1
    n_distinct = int(gid[-1].item()) + 1
    gmin = torch.full((n_distinct,), torch.iinfo(torch.int64).max, dtype=torch.int64, device=dev)
    gmin.scatter_reduce_(0, gid, sp[:, 2], reduce="amin")
    keep = torch.empty(m, dtype=torch.bool, device=dev)
    keep[order] = sp[:, 2] == gmin[gid]
    return keep, n_distinct


def _check_collective_device(dist, *tensors):
    """RCCL moves device memory: under the nccl backend every tensor handed to a collective has to live on THIS rank's
    GPU (a host tensor, or one on another rank's device, fails deep inside the collective -- or hangs the group)."""
    if dist is None or not dist.is_initialized() or dist.get_backend() != "nccl":
        return
    cur = torch.cuda.current_device()
    for t in tensors:
        if not t.is_cuda or t.device.index != cur:
            raise ValueError(f"tensor on {t.device} handed to an RCCL collective of the rank whose device is cuda:{cur}")


def _merge_owner(md5, first_seen, dist, engine):
    world = dist.get_world_size()
    dev = md5.device
    _check_collective_device(dist, md5, first_seen)
    n = md5.shape[0]
    pay = _pack_rows(md5, first_seen)
    owner = (((pay[:, 0] >> 17) & 0x7FFFFFFF) % world).to(torch.int16)    # any fixed function of the digest
    order = torch.argsort(owner, stable=True)                    # 16-bit keys: a two-pass radix sort
    send = pay.index_select(0, order)
    edges = torch.searchsorted(owner.index_select(0, order), torch.arange(world + 1, device=dev, dtype=torch.int16))
    send_counts = (edges[1:] - edges[:-1]).to(torch.int64)
    recv_counts = torch.empty(world, dtype=torch.int64, device=dev)
    dist.all_to_all_single(recv_counts, send_counts)
    sc, rc = send_counts.tolist(), recv_counts.tolist()
    recv = torch.empty((sum(rc), 3), dtype=torch.int64, device=dev)
    dist.all_to_all_single(recv, send, output_split_sizes=rc, input_split_sizes=sc)
    if recv.is_cuda:
        torch.cuda.synchronize(dev)                             # the library runs on its own stream
    keep_recv, n_owned = _first_per_digest(recv, engine)
    keep_back = torch.empty(n, dtype=torch.uint8, device=dev)
    dist.all_to_all_single(keep_back, keep_recv.to(torch.uint8).contiguous(), output_split_sizes=sc, input_split_sizes=rc)
    keep = torch.empty(n, dtype=torch.bool, device=dev)
    keep[order] = keep_back.bool()
    tot = torch.tensor([n_owned], dtype=torch.int64, device=dev)
    dist.all_reduce(tot)
    return keep, int(tot.item())


def merge_pattern_tensors(md5, first_seen, dist=None, engine=None, method="owner"):
    """md5: uint8 [n,16], first_seen: int64 [n] (this rank's patterns, any device).
    Returns (keep_mask [n] bool: this rank's row is the global first for its digest,
             n_global: number of distinct digests over all ranks).
    With CUDA tensors and an `engine`, the marking runs in the library's hash-table kernels (pf_merge_patterns,
    O(rows) atomics); otherwise (CPU / gloo tests) by a lexicographic sort in torch."""
    dev = md5.device
    n = md5.shape[0]
    use_kernel = engine is not None and md5.is_cuda
    multi = dist is not None and dist.is_initialized() and dist.get_world_size() > 1
    if multi and method == "owner":
        return _merge_owner(md5, first_seen, dist, engine)
    if multi:
        world, rank = dist.get_world_size(), dist.get_rank()
        _check_collective_device(dist, md5, first_seen)
        counts = torch.zeros(world, dtype=torch.int64, device=dev)
        mine = torch.tensor([n], dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(counts, mine)
        cl = counts.tolist()
        nmax = max(max(cl), 1)
        pay = torch.zeros((nmax, 3), dtype=torch.int64, device=dev)
        if n:
            pay[:n] = _pack_rows(md5, first_seen)
        allpay = torch.empty((world * nmax, 3), dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(allpay, pay)
        if use_kernel:
            # straight from the padded gather buffer: no compaction, no sort
            keep = torch.zeros(max(n, 1), dtype=torch.uint8, device=dev)
            torch.cuda.synchronize(dev)
            n_global = C.c_uint64()
            _lib.check(engine.L.pf_merge_patterns_padded(engine.ctx, C.c_void_p(allpay.data_ptr()), world, nmax,
                                                         C.c_void_p(counts.data_ptr()), rank, n,
                                                         C.c_void_p(keep.data_ptr()), C.byref(n_global)))
            return keep[:n].bool(), int(n_global.value)
        valid = (torch.arange(nmax, device=dev)[None, :] < counts[:, None]).reshape(-1)
        allpay = allpay[valid]
        my_first = sum(cl[:rank])
    else:
        allpay = _pack_rows(md5, first_seen)
        my_first = 0
    keep_all, n_global = _first_per_digest(allpay, engine if use_kernel else None)
    return keep_all[my_first:my_first + n], n_global


def _merge_on_device(engine, allpay, my_first, my_count):
    """allpay: int64 [rows,3] on the GPU = {md5 lo, md5 hi, first_seen} of every rank, this rank's rows at
    [my_first, my_first+my_count)."""
    allpay = allpay.contiguous()
    keep = torch.zeros(max(my_count, 1), dtype=torch.uint8, device=allpay.device)
    torch.cuda.synchronize(allpay.device)        # the library runs on its own stream
    n_global = C.c_uint64()
    _lib.check(engine.L.pf_merge_patterns(engine.ctx, C.c_void_p(allpay.data_ptr()), allpay.shape[0], int(my_first),
                                          int(my_count), C.c_void_p(keep.data_ptr()), C.byref(n_global)))
    return keep[:my_count].bool(), int(n_global.value)


def export_patterns(engine, device):
    """(md5 uint8 [n,16], first_seen int64 [n]) of the engine's pattern pool as torch tensors."""
    L = engine.L
    if device.type == "cuda":
        n = int(engine_n_patterns(engine))
        md5 = torch.empty((max(n, 1), 16), dtype=torch.uint8, device=device)
        fs = torch.empty(max(n, 1), dtype=torch.int64, device=device)
        got = C.c_uint64()
        _lib.check(L.pf_export_patterns_dev(engine.ctx, n, C.c_void_p(md5.data_ptr()), C.c_void_p(fs.data_ptr()),
                                            C.byref(got)))
        return md5[:n], fs[:n]
    n = C.c_uint64()
    p_md5 = C.POINTER(C.c_uint8)()
    p_fs = C.POINTER(C.c_uint64)()
    _lib.check(L.pf_export_patterns(engine.ctx, C.byref(n), C.byref(p_md5), C.byref(p_fs)))
    n = n.value
    if n == 0:
        return torch.zeros((0, 16), dtype=torch.uint8), torch.zeros(0, dtype=torch.int64)
    md5 = torch.from_numpy(np.ctypeslib.as_array(p_md5, shape=(n * 16,)).reshape(n, 16).copy())
    fs = torch.from_numpy(np.ctypeslib.as_array(p_fs, shape=(n,)).astype(np.int64))
    return md5, fs


def engine_n_patterns(engine):
    n = C.c_uint64()
    _lib.check(engine.L.pf_pattern_count(engine.ctx, C.byref(n)))
    return n.value


def merge_patterns(engine, dist, device, method="owner"):
    """Exchange the engine's pattern digests and return the number of run-global unique patterns."""
    md5, fs = export_patterns(engine, device)
    _keep, n_global = merge_pattern_tensors(md5, fs, dist, engine=engine, method=method)
    return n_global
