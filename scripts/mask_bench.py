"""dev/bench helper: the stage-1 loss (train.py:450-456) on a 1080p feature map with 96 SAM-like masks:
mask_feature_mean + cohesion_loss + separation_loss forward and backward through the HIP segmented reductions."""
import sys, time
import torch
sys.path.insert(0, ".")
from opengaussian_amd import mask_ops as mk, _lib
dev = torch.device("cuda:0")
H, W, N, C = 1080, 1920, 96, 6
g = torch.Generator().manual_seed(0)
feat = torch.rand(C, H, W, generator=g).to(dev).requires_grad_(True)
coarse = torch.randint(0, N + 1, ((H + 31) // 32, (W + 31) // 32), generator=g)
labels = coarse.repeat_interleave(32, 0).repeat_interleave(32, 1)[:H, :W].to(dev)
masks = torch.stack([labels == (n + 1) for n in range(N)])            # [N,H,W] bool
sil = torch.rand(1, H, W, generator=g).to(dev)


def step():
    feat.grad = None
    mean = mk.mask_feature_mean(feat, masks, image_mask=sil)
    loss = mk.separation_loss(mean, 1000) + 0.1 * mk.cohesion_loss(feat, masks, mean)
    loss.backward()


for _ in range(5):
    step()
torch.cuda.synchronize()
K = 50
t0 = time.perf_counter()
for _ in range(K):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
print(f"stage-1 loss fwd+bwd, {W}x{H}, {N} masks: {dt * 1e3:.3f} ms/step", flush=True)
_lib.prof_enable(1)
for _ in range(10):
    step()
torch.cuda.synchronize()
prof = _lib.prof_collect(); _lib.prof_enable(0)
stack = N * H * W; fmap = C * H * W * 4; wmap = H * W * 4
alg = {"mask_feature_sums_kernel": fmap + wmap + stack, "mask_feature_sums_backward_kernel": wmap + stack + fmap,
       "mask_cohesion_kernel": fmap + stack, "mask_cohesion_backward_kernel": 2 * fmap + stack}
for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"]):
    ms = v["total_ms"] / v["calls"]
    base = k.strip("()").split("<")[0]
    extra = f"  {alg[base] / ms / 1e6:7.0f} GB/s algorithmic" if base in alg else ""
    print(f"  {k:50s} {ms * 1e3:8.1f} us{extra}", flush=True)
