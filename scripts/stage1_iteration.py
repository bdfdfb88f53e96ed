"""One whole stage-1 training iteration as train.py:352-358,425-456,594-611 wires it, at the bench size (1 M Gaussians,
1920x1080, 96 SAM-like masks, 8 cameras cycled): render() (fused RGB + 6-D ins_feat pass, geometry detached) ->
mask_feature_mean weighted by the silhouette -> separation + 0.1 * cohesion -> backward through the rasterizer's
features-only path -> FusedAdam on the instance features.  Prints one JSON line: ms per iteration and the split between
the render (forward + backward kernels) and everything else."""
import json
import math
import os
import sys
import time
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opengaussian_amd import _lib, mask_ops as mk  # noqa: E402
from opengaussian_amd.optim import FusedAdam  # noqa: E402
from opengaussian_amd.renderer import render  # noqa: E402
from opengaussian_amd.synthetic import make_scene, orbit_camera  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    P, W, H, f, V, NM = 1_000_000, 1920, 1080, 1000.0, 8, 96
    sc = make_scene(P, W, H, f, f, seed=0).to(dev)
    cams = [orbit_camera(W, H, f, f, v, V).to(dev) for v in range(V)]
    ins = torch.nn.Parameter((sc.ins_feat * 2 - 1).clone())
    geo = types.SimpleNamespace(
        get_xyz=sc.means3D, get_scaling=sc.scales, get_rotation=sc.rotations, get_opacity=sc.opacities,
        get_features=sc.shs, get_ins_feat=lambda origin=False: torch.nn.functional.normalize(ins, dim=1),
        active_sh_degree=3, max_sh_degree=3)
    pipe = types.SimpleNamespace(debug=False, compute_cov3D_python=False, convert_SHs_python=False)
    bg = torch.zeros(3, device=dev)
    # SAM-like label image: 12 x 8 blocky regions, label 0 = invalid
    lab = (torch.arange(H, device=dev)[:, None] // 135) * 12 + (torch.arange(W, device=dev)[None, :] // 160) + 1
    lab[:, :16] = 0
    masks = torch.stack([lab == (n + 1) for n in range(NM)])
    opt = FusedAdam([{"params": [ins], "lr": 1e-3, "name": "ins_feat"}], lr=0.0, eps=1e-15)

    def iteration(it):
        out = render(cams[it % V], geo, pipe, bg, iteration=it, rescale=False)
        feat, sil = out["ins_feat"], out["silhouette"]
        mean = mk.mask_feature_mean(feat, masks, image_mask=sil)
        loss = mk.separation_loss(mean, it) + 0.1 * mk.cohesion_loss(feat, masks, mean)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        return loss

    for it in range(10):
        iteration(it)
    torch.cuda.synchronize()
    K = 100
    t0 = time.perf_counter()
    for it in range(K):
        iteration(10 + it)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / K * 1e3
    _lib.prof_enable(1)
    for it in range(16):
        iteration(200 + it)
    torch.cuda.synchronize()
    pr = _lib.prof_collect()
    _lib.prof_enable(0)
    kern = {k: v["total_ms"] / 16 for k, v in pr.items()}
    raster = sum(v for k, v in kern.items() if not k.startswith(("mask_", "separation", "adam")))
    print(json.dumps({"stage1_iteration_ms": ms, "masks": NM, "hip_kernel_ms_per_iteration": sum(kern.values()),
                      "rasterizer_kernels_ms": raster,
                      "loss_and_optimizer_kernels_ms": {k: round(v, 4) for k, v in kern.items() if k.startswith(("mask_", "separation", "adam"))}}))


if __name__ == "__main__":
    main()
