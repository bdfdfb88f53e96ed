"""dev helper: large-scene stress (8 M Gaussians at 1080p, D in the tens of millions): no overflow, finite
outputs, invariants, timing."""
import math, sys, time
import torch
sys.path.insert(0, ".")
from opengaussian_amd.rasterizer import GaussianRasterizationSettings, rasterize_fused
from opengaussian_amd.synthetic import make_scene, make_camera
dev = torch.device("cuda:0")
P, W, H, f = 8_000_000, 1920, 1080, 1000.0
sc = make_scene(P, W, H, f, f, seed=3, log_scale_mean=-4.0).to(dev)
cam = make_camera(W, H, f, f).to(dev)
rs = GaussianRasterizationSettings(H, W, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), torch.zeros(3, device=dev), 1.0,
                                   cam.world_view_transform, cam.full_proj_transform, 3, cam.camera_center, False, False)
leaves = {k: getattr(sc, k).requires_grad_(True) for k in ("means3D", "scales", "rotations", "opacities", "shs", "ins_feat")}
for it in range(3):
    for v in leaves.values():
        v.grad = None
    m2 = torch.zeros(P, 3, device=dev, requires_grad=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    c, r, d, a = rasterize_fused(leaves["means3D"], m2, leaves["opacities"], leaves["shs"], leaves["ins_feat"], rs,
                                 scales=leaves["scales"], rotations=leaves["rotations"])
    (c.sum() + a.sum()).backward()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    D = c.grad_fn.num_rendered if c.grad_fn is not None else -1
    print(f"iter {it}: {dt * 1e3:.2f} ms, visible {int((r > 0).sum())}, alpha range [{float(a.min()):.4f}, {float(a.max()):.6f}]", flush=True)
assert torch.isfinite(c).all() and torch.isfinite(d).all() and float(a.max()) <= 1.0
for k, v in leaves.items():
    assert torch.isfinite(v.grad).all(), k
print("max memory allocated %.1f GB" % (torch.cuda.max_memory_allocated() / 1e9))
print("stress ok")
