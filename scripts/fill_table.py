"""dev helper: the extra rows of BASELINE.md section 4 (C2, RGB-only, k-means at N = 10 k)."""
import json, subprocess, sys, time
import torch
sys.path.insert(0, ".")
rows = {}
for name, args in (("S1M_rgb", ["--rgb-only"]), ("C2_rgb", ["--workload", "C2-100k-800", "--rgb-only"]),
                   ("C2_fused", ["--workload", "C2-100k-800"])):
    out = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--no-kmeans", *args], capture_output=True, text=True)
    d = json.loads(out.stdout.strip().splitlines()[-1])
    k = d["kernels_ms_per_step"]
    bwd = sum(v for n, v in k.items() if "backward" in n)
    rows[name] = dict(mpix=d["value"], ms=d["ms_per_step"], ms_fwd=sum(k.values()) - bwd, ms_bwd=bwd, D=d["scene"]["D_num_rendered"],
                      step_GBps=d["step_algorithmic_GBps"], roofline=d["roofline"])
    print(name, json.dumps(rows[name]), flush=True)
from opengaussian_amd import kmeans as km
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for N in (10_000, 2_000_000):
    feat = torch.cat([torch.rand(N, 6, generator=g), torch.randn(N, 3, generator=g)], 1).to(dev)
    init = feat[:64].clone()
    for _ in range(3):
        c = init.clone(); km.lloyd(feat, c, 5, N // 10000 + 1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    K = 50
    for _ in range(K):
        c = init.clone(); km.lloyd(feat, c, 5, N // 10000 + 1)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
    print(f"kmeans N={N} d=9 k=64: {5 / dt:.0f} it/s ({dt * 1e3:.3f} ms per 5-iteration call incl. final assign)", flush=True)
# CPU oracle k-means (port) at N = 10k
import numpy as np
from oracle import kmeans_oracle as ko
feat = torch.cat([torch.rand(10000, 6, generator=g), torch.randn(10000, 3, generator=g)], 1).numpy()
t0 = time.perf_counter(); ko.lloyd(feat, feat[:64].copy(), 5, 2); dt = time.perf_counter() - t0
print(f"kmeans CPU oracle N=10000: {5 / dt:.1f} it/s", flush=True)
