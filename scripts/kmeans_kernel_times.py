import sys, json, time
sys.path.insert(0, ".")
import torch
from opengaussian_amd import _lib
from opengaussian_amd.kmeans import lloyd
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
N, d, k, iters = 2_000_000, 9, 64, 5
feat = torch.cat([torch.rand(N, 6, generator=g), torch.randn(N, 3, generator=g)], dim=1).to(dev)
cent = feat[torch.randperm(N, generator=g)[:k].to(dev)].clone()
for _ in range(3): lloyd(feat, cent.clone(), iters=iters, nchunks=N // 10000 + 1)
torch.cuda.synchronize()
_lib.prof_enable(1)
for _ in range(5): lloyd(feat, cent.clone(), iters=iters, nchunks=N // 10000 + 1)
torch.cuda.synchronize()
p = _lib.prof_collect(); _lib.prof_enable(0)
print({k_: round(v["total_ms"] / v["calls"] * 1e3, 1) for k_, v in p.items()}, {k_: v["calls"] for k_, v in p.items()})
