"""A-B of the heaviest-first workgroup order (OGS_TILE_ORDER=0/1, read once per process) on a scene whose screen-space
density is NOT uniform: the S1M-1080p scene of bench.py with the image-plane coordinates pulled towards the centre
(u -> sign(u) |u|^skew).  Prints one JSON line: ms per fused forward+backward step and per blend kernel.
usage: OGS_TILE_ORDER=0|1 python scripts/tile_order_ab.py [--skew 2.0] [--steps 40]"""
import argparse
import json
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opengaussian_amd import _lib  # noqa: E402
from opengaussian_amd.rasterizer import GaussianRasterizationSettings, rasterize_fused  # noqa: E402
from opengaussian_amd.synthetic import make_scene, orbit_camera  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--skew", type=float, default=2.0)
    ap.add_argument("--steps", type=int, default=40)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    P, W, H, f = 1_000_000, 1920, 1080, 1000.0
    sc = make_scene(P, W, H, f, f, seed=0)
    z = sc.means3D[:, 2:3]
    uv = sc.means3D[:, :2] / z.abs().clamp_min(1e-3)
    tan = torch.tensor([W / (2 * f), H / (2 * f)])
    n = (uv / tan / 1.1).clamp(-1, 1)
    n = n.sign() * n.abs() ** a.skew
    means = torch.cat([n * 1.1 * tan * z.abs().clamp_min(1e-3), z], dim=1)
    cam = orbit_camera(W, H, f, f, view_index=0, num_views=8).to(dev)
    rs = GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5),
        bg=torch.zeros(3, device=dev), scale_modifier=1.0, viewmatrix=cam.world_view_transform,
        projmatrix=cam.full_proj_transform, sh_degree=3, campos=cam.camera_center, prefiltered=False, debug=False)
    leaves = {k: v.to(dev).requires_grad_(True) for k, v in dict(
        means3D=means.contiguous(), opacities=sc.opacities, shs=sc.shs, ins_feat=sc.ins_feat, scales=sc.scales,
        rotations=sc.rotations).items()}
    g = torch.Generator(device="cpu").manual_seed(1)
    gCF = torch.randn(9, H, W, generator=g).to(dev)
    gA = torch.randn(1, H, W, generator=g).to(dev)

    def step():
        for v in leaves.values():
            v.grad = None
        m2 = torch.zeros(P, 3, device=dev, requires_grad=True)
        color, radii, depth, alpha = rasterize_fused(leaves["means3D"], m2, leaves["opacities"], leaves["shs"],
                                                     leaves["ins_feat"], rs, scales=leaves["scales"],
                                                     rotations=leaves["rotations"])
        D = color.grad_fn.num_rendered
        torch.autograd.backward([color, alpha], [gCF, gA])
        return D

    for _ in range(5):
        D = step()
    torch.cuda.synchronize()
    _lib.prof_enable(2)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    prof = _lib.prof_collect()
    _lib.prof_enable(0)
    print(json.dumps({"tile_order": os.environ.get("OGS_TILE_ORDER", "1"), "skew": a.skew, "num_rendered": int(D),
                      "ms_per_step": el / a.steps * 1e3,
                      "kernels_ms": {k: v["total_ms"] / max(v["calls"], 1) for k, v in prof.items()}}))


if __name__ == "__main__":
    main()
