"""Per-kernel timing of the k-means path on the GPU (dev helper): python scripts/kmeans_prof.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from opengaussian_amd import _lib
from opengaussian_amd.kmeans import lloyd

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for (N, d, k) in [(2_000_000, 9, 64), (2_000_000, 6, 10), (10_000, 9, 64)]:
    feat = torch.cat([torch.rand(N, min(d, 6), generator=g), torch.randn(N, max(d - 6, 0), generator=g)], dim=1).to(dev)
    cent = feat[torch.randperm(N, generator=g)[:k].to(dev)].clone()
    for _ in range(2):
        lloyd(feat, cent.clone(), iters=5, nchunks=N // 10000 + 1)
    torch.cuda.synchronize()
    _lib.prof_enable(True)
    t0 = time.time()
    reps = 5
    for _ in range(reps):
        lloyd(feat, cent.clone(), iters=5, nchunks=N // 10000 + 1)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / reps
    prof = _lib.prof_collect()
    _lib.prof_enable(False)
    print(f"N={N} d={d} k={k}: {dt*1e3:.3f} ms/call, {5/dt:.0f} it/s")
    for name, v in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"]):
        print(f"   {name:45s} calls {v['calls']:4d}  avg {v['total_ms']/v['calls']*1e3:9.1f} us")
