"""N>1 path on CPU: world_size-2 gloo run of the pattern merge (the only collective of the path)
and the contiguous shard split.  No GPU, no HIP calls."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from panfeed_amd.distributed import merge_pattern_tensors, shard_range


def _make(rank, world, seed=3):
    """digests with overlaps across ranks; first_seen follows contiguous cluster ranges"""
    rng = np.random.default_rng(seed)
    pool = rng.integers(0, 256, size=(60, 16), dtype=np.uint8)
    per = []
    for r in range(world):
        pick = rng.choice(60, size=35, replace=False)
        fs = (np.int64(r * 1000) + rng.integers(0, 1000, size=35).astype(np.int64)) << 32 | rng.integers(0, 99, size=35)
        per.append((pool[pick], fs))
    return per


def _worker(rank, world, port, q, method="owner"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    per = _make(rank, world)
    md5 = torch.from_numpy(per[rank][0].copy())
    fs = torch.from_numpy(per[rank][1].copy())
    keep, n_global = merge_pattern_tensors(md5, fs, dist, method=method)
    q.put((rank, keep.numpy().tolist(), n_global))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("method", ["owner", "allgather"])
@pytest.mark.parametrize("world", [2, 3, 8])
def test_merge_patterns_gloo(world, method):
    """both forms of the exchange (all-to-all to the digest's owner and back / all-gather) mark the same rows"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + world + (10 if method == "owner" else 0)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, method)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        r, keep, n = q.get(timeout=120)
        got[r] = (keep, n)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    per = _make(0, world)
    best = {}
    for r in range(world):
        for d, f in zip(per[r][0], per[r][1]):
            key = d.tobytes()
            if key not in best or f < best[key][0]:
                best[key] = (int(f), r)
    for r in range(world):
        keep, n = got[r]
        assert n == len(best)
        exp = [best[d.tobytes()] == (int(f), r) for d, f in zip(per[r][0], per[r][1])]
        assert keep == exp


def test_merge_single_process_and_empty():
    md5 = torch.zeros((0, 16), dtype=torch.uint8)
    keep, n = merge_pattern_tensors(md5, torch.zeros(0, dtype=torch.int64))
    assert n == 0 and keep.numel() == 0
    d = torch.tensor(np.random.default_rng(0).integers(0, 256, (5, 16), dtype=np.uint8))
    d[3] = d[1]
    fs = torch.tensor([5, 9, 7, 2, 1], dtype=torch.int64)
    keep, n = merge_pattern_tensors(d, fs)
    assert n == 4 and keep.tolist() == [True, False, True, True, True]


def test_shard_range_contiguous_and_balanced():
    for n, w in ((10, 3), (50000, 8), (7, 8), (0, 2)):
        cuts = [shard_range(n, r, w) for r in range(w)]
        assert cuts[0][0] == 0 and cuts[-1][1] == n
        for a, b in zip(cuts, cuts[1:]):
            assert a[1] == b[0]
        assert max(e - s for s, e in cuts) - min(e - s for s, e in cuts) <= 1
    wts = np.array([1, 1, 1, 1, 100, 1, 1, 1])
    cuts = [shard_range(8, r, 2, wts) for r in range(2)]
    assert cuts[0][1] == cuts[1][0] and cuts[0][0] == 0 and cuts[1][1] == 8
