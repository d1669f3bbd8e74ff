import sys, time, torch
sys.path.insert(0, ".")
from panfeed_amd.engine import Engine
from panfeed_amd.distributed import _merge_on_device, merge_pattern_tensors
eng = Engine(klength=31, max_strains=64)
dev = torch.device("cuda", 0)
P, R = 2_640_000, 8
g = torch.Generator(device="cpu").manual_seed(0)
base = torch.randint(-2**62, 2**62, (P * R, 3), generator=g).to(dev)
# 10 % of the digests are shared between ranks
base[P:P + P // 10, :2] = base[:P // 10, :2]
for _ in range(2):
    torch.cuda.synchronize(); t = time.time()
    keep, n = _merge_on_device(eng, base, 0, P)
    torch.cuda.synchronize(); print("device merge: %.1f ms, n_global %d, kept %d" % ((time.time() - t) * 1e3, n, int(keep.sum())))
md5 = base[:, :2].contiguous().view(torch.uint8).view(-1, 16)
torch.cuda.synchronize(); t = time.time()
k2, n2 = merge_pattern_tensors(md5, base[:, 2].contiguous())
torch.cuda.synchronize(); print("torch sort merge on GPU: %.1f ms n %d" % ((time.time() - t) * 1e3, n2))
