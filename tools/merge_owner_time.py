"""Local cost of the owner form of the multi-GPU pattern merge (distributed._merge_owner) for one rank of an 8-rank
run, on ONE GPU: the collectives are replaced by local copies of the same sizes (what a rank receives is taken to be what
it sends), so what is timed is everything but the xGMI transfers: packing, partition by owner, the merge kernels, the
scatter of the marks.  usage: python tools/merge_owner_time.py [patterns per rank] [world]"""
import sys
import time

import torch

sys.path.insert(0, ".")
from panfeed_amd import distributed  # noqa: E402
from panfeed_amd.engine import Engine  # noqa: E402


class FakeDist:
    def __init__(self, world):
        self.world = world

    def is_initialized(self):
        return True

    def get_world_size(self):
        return self.world

    def get_rank(self):
        return 0

    def all_to_all_single(self, out, inp, output_split_sizes=None, input_split_sizes=None):
        out.copy_(inp.reshape(out.shape) if out.numel() == inp.numel() else inp[:out.shape[0]])

    def all_reduce(self, t, op=None):
        t.mul_(self.world)

    def all_gather_into_tensor(self, out, inp):
        out.view(self.world, -1).copy_(inp.reshape(1, -1).expand(self.world, -1))


P = int(sys.argv[1]) if len(sys.argv) > 1 else 2_640_000
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
eng = Engine(klength=31, max_strains=64)
dev = torch.device("cuda", 0)
g = torch.Generator(device="cpu").manual_seed(0)
md5 = torch.randint(0, 256, (P, 16), generator=g, dtype=torch.uint8).to(dev)
fs = torch.arange(P, dtype=torch.int64, device=dev)
for method in ("owner", "allgather"):
    for rep in range(3):
        torch.cuda.synchronize()
        t = time.time()
        keep, n = distributed.merge_pattern_tensors(md5, fs, FakeDist(world), engine=eng, method=method)
        torch.cuda.synchronize()
        print(f"{method}: {1e3 * (time.time() - t):.2f} ms  ({P} rows per rank, world {world}; n_global {n}, kept {int(keep.sum())})")
