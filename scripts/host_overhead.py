"""Dev helper: is the bench step host-bound?  Prints enqueue time vs GPU time per step."""
import os, sys, time, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from opengaussian_amd.rasterizer import GaussianRasterizationSettings, rasterize_fused
from opengaussian_amd.synthetic import make_scene, make_camera
dev = torch.device("cuda:0")
P, W, H, f = 1_000_000, 1920, 1080, 1000.0
sc = make_scene(P, W, H, f, f).to(dev); cam = make_camera(W, H, f, f).to(dev)
rs = GaussianRasterizationSettings(H, W, math.tan(cam.FoVx/2), math.tan(cam.FoVy/2), torch.zeros(3, device=dev), 1.0,
                                   cam.world_view_transform, cam.full_proj_transform, 3, cam.camera_center, False, False)
leaves = {k: getattr(sc, k).requires_grad_(True) for k in ("means3D", "scales", "rotations", "opacities", "shs", "ins_feat")}
g = torch.Generator().manual_seed(0)
gCF = torch.randn(9, H, W, generator=g).to(dev); gA = torch.randn(1, H, W, generator=g).to(dev)
def step():
    for v in leaves.values(): v.grad = None
    m2 = torch.zeros(P, 3, device=dev, requires_grad=True)
    t0 = time.perf_counter()
    c, r, d, a = rasterize_fused(leaves["means3D"], m2, leaves["opacities"], leaves["shs"], leaves["ins_feat"], rs,
                                 scales=leaves["scales"], rotations=leaves["rotations"])
    t1 = time.perf_counter()
    torch.autograd.backward([c, a], [gCF, gA])
    t2 = time.perf_counter()
    return t1 - t0, t2 - t1
for _ in range(3): step()
torch.cuda.synchronize()
K = 20
t0 = time.perf_counter(); tf = tb = 0
for _ in range(K):
    a, b = step(); tf += a; tb += b
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"per step: host enqueue {t_enq/K*1e3:.3f} ms (forward call {tf/K*1e3:.3f}, backward call {tb/K*1e3:.3f}), total {t_all/K*1e3:.3f} ms")
