"""dev helper: per-call latency of small passes (the regime of the per-cluster calls and of the SAM refiner's
single-Gaussian renders): host-side cost of one GaussianRasterizer call vs GPU time."""
import math, sys, time
import torch
sys.path.insert(0, ".")
from opengaussian_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
from opengaussian_amd.synthetic import make_scene, make_camera
dev = torch.device("cuda:0")
W, H, f = 648, 484, 500.0
cam = make_camera(W, H, f, f).to(dev)
rs = GaussianRasterizationSettings(H, W, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), torch.zeros(3, device=dev), 1.0,
                                   cam.world_view_transform, cam.full_proj_transform, 3, cam.camera_center, False, False)
rast = GaussianRasterizer(rs)
for P in (1, 1000, 10000, 100000):
    sc = make_scene(P, W, H, f, f, seed=1).to(dev)
    m2 = torch.zeros(P, 3, device=dev)
    feat = sc.ins_feat.contiguous()
    def call(grad):
        if grad:
            f_ = feat.clone().requires_grad_(True)
            c, r, d, a = rast(means3D=sc.means3D, means2D=m2, opacities=sc.opacities, colors_precomp=f_, scales=sc.scales, rotations=sc.rotations)
            c.sum().backward()
        else:
            with torch.no_grad():
                rast(means3D=sc.means3D, means2D=m2, opacities=sc.opacities, colors_precomp=feat, scales=sc.scales, rotations=sc.rotations)
    for grad in (False, True):
        for _ in range(5): call(grad)
        torch.cuda.synchronize()
        K = 50
        t0 = time.perf_counter()
        for _ in range(K): call(grad)
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        print(f"P={P:7d} 6ch {'fwd+bwd' if grad else 'fwd    '}: {t_all / K * 1e3:7.3f} ms/call (host enqueue {t_host / K * 1e3:7.3f})", flush=True)
