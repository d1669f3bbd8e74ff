"""The RCCL calls of the multi-GPU path on whatever GPUs this process group has (one rank is enough to check that the
collectives are called the way RCCL accepts: split sizes, dtypes, device tensors): the owner and the all-gather form of
the pattern merge through a real `torch.distributed` NCCL group, against the single-process marks.

    python tools/nccl_selftest.py                      (world 1)
    torchrun --nproc-per-node N tools/nccl_selftest.py (one rank per GPU)"""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, ".")
from panfeed_amd import distributed  # noqa: E402
from panfeed_amd.engine import Engine  # noqa: E402

rank = int(os.environ.get("RANK", "0"))
world = int(os.environ.get("WORLD_SIZE", "1"))
local = int(os.environ.get("LOCAL_RANK", "0"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
eng = Engine(klength=31, max_strains=64, device=local)
g = torch.Generator(device="cpu").manual_seed(7)                 # every rank draws the same pool of digests
pool = torch.randint(0, 256, (5000, 16), generator=g, dtype=torch.uint8)
pick = torch.randperm(5000, generator=torch.Generator(device="cpu").manual_seed(100 + rank))[:3000]
md5 = pool[pick].to(dev)
fs = ((torch.arange(3000, dtype=torch.int64) + rank * 100000) << 32).to(dev)
ok = True
for method in ("owner", "allgather"):
    if world > 1:
        keep, n = distributed.merge_pattern_tensors(md5, fs, dist, engine=eng, method=method)
    elif method == "owner":
        keep, n = distributed._merge_owner(md5, fs, dist, eng)    # one rank: every row goes to itself, through RCCL
    else:
        continue
    # expectation: a digest is kept by the lowest rank that drew it
    seen = set()
    for r in range(rank):
        pr = torch.randperm(5000, generator=torch.Generator(device="cpu").manual_seed(100 + r))[:3000]
        seen.update(pr.tolist())
    exp = torch.tensor([int(i) not in seen for i in pick.tolist()])
    all_ids = set()
    for r in range(world):
        all_ids.update(torch.randperm(5000, generator=torch.Generator(device="cpu").manual_seed(100 + r))[:3000].tolist())
    good = bool(torch.equal(keep.cpu(), exp)) and n == len(all_ids)
    ok = ok and good
    print(f"rank {rank}/{world} {method}: n_global {n} (expected {len(all_ids)}), keep marks {'ok' if good else 'WRONG'}", flush=True)
# the end of a shard (sharded.finish_shard) through RCCL on THIS rank's device: the ordinal-range check's all-gather and
# the id selection
import numpy as np  # noqa: E402
from panfeed_amd import sharded  # noqa: E402
fs_np = fs.cpu().numpy().view(np.uint64)
sharded.check_first_seen_disjoint(fs_np, dist, dev)
ids, n_glob = sharded.kept_pattern_ids(md5.cpu().numpy(), fs_np, dist if world > 1 else None, engine=eng, device=dev)
good = n_glob == (len(all_ids) if world > 1 else 3000) and len(ids) == int(exp.sum() if world > 1 else 3000)
ok = ok and good
print(f"rank {rank}/{world} finish_shard path on {dev}: {len(ids)} rows kept of {n_glob} run-global patterns "
      f"{'ok' if good else 'WRONG'}", flush=True)
dist.barrier(device_ids=[local])
dist.destroy_process_group()
eng.close()
sys.exit(0 if ok else 1)
