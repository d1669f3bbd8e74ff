"""What the multi-GPU pattern merge costs on ONE GPU besides the wire: a world of `W` ranks is played by this rank alone
(its rows are split by owner as for an all-to-all, every "received" row is its own), so the torch glue, the sort by
owner, the library's marking kernels and the way back are all timed -- everything except RCCL itself.
usage: python tools/merge_time.py [clusters] [world]"""
import sys
import time

import torch

sys.path.insert(0, ".")
from panfeed_amd import devbatch, distributed as D, synth  # noqa: E402
from panfeed_amd.engine import Engine  # noqa: E402


class FakeDist:
    """all-to-all with oneself: what is sent to rank r comes back as what rank r sent"""
    def __init__(self, world):
        self.world = world

    def get_world_size(self):
        return self.world

    def get_rank(self):
        return 0

    def is_initialized(self):
        return True

    def all_to_all_single(self, out, inp, output_split_sizes=None, input_split_sizes=None):
        out.copy_(inp)

    def all_reduce(self, t):
        return t


n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
eng = Engine(klength=31, max_strains=1024, pattern_capacity=1 << 25)
db = devbatch.from_synth(eng, synth.generate(n, 1000, flank=100, n_rate=0.0), 31)
db.submit(eng)
dev = torch.device("cuda", 0)
fd = FakeDist(world)
for rep in range(4):
    torch.cuda.synchronize()
    t0 = time.time()
    md5, fs = D.export_patterns(eng, dev)
    torch.cuda.synchronize()
    t1 = time.time()
    keep, n_global = D.merge_pattern_tensors(md5, fs, fd, engine=eng, method="owner")
    torch.cuda.synchronize()
    t2 = time.time()
    print(f"patterns {md5.shape[0]}  export {1e3 * (t1 - t0):.3f} ms  owner merge without the wire {1e3 * (t2 - t1):.3f} ms  "
          f"kept {int(keep.sum())} global {n_global}")
