"""How many atomic gradient records would the backward blend send if one wave owned TWO quadrants of a tile?

The backward adds one 128-byte record per (entry, 8x8 quadrant) that received a contribution (the kernel is
bound by the L2's atomic path, charged per 64-byte segment).  From the quadrant streams of a default pass -- per tile and quadrant
the compact indices of the entries that can reach the quadrant, cut at the quadrant's deepest last contributor -- this counts

    sum over tiles of |S0| + |S1| + |S2| + |S3|                 (today: one record per (entry, quadrant))
    |S0 u S1| + |S2 u S3|,  |S0 u S2| + |S1 u S3|               (a wave owns a horizontal / a vertical pair of quadrants)
    |S0 u S1 u S2 u S3|                                         (one record per (entry, tile))

usage: python scripts/pair_merge_stats.py [workload]"""
import json
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import WORKLOADS  # noqa: E402
from opengaussian_amd.rasterizer import GaussianRasterizationSettings, rasterize_fused  # noqa: E402
from opengaussian_amd.synthetic import make_scene, orbit_camera  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "S1M-1080p"
    wl = WORKLOADS[name]
    P, W, H, f = wl["P"], wl["W"], wl["H"], wl["f"]
    dev = torch.device("cuda:0")
    sc = make_scene(P, W, H, f, f, seed=0).to(dev)
    cam = orbit_camera(W, H, f, f, view_index=0, num_views=8).to(dev)
    rs = GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5),
        bg=torch.zeros(3, device=dev), scale_modifier=1.0, viewmatrix=cam.world_view_transform,
        projmatrix=cam.full_proj_transform, sh_degree=3, campos=cam.camera_center, prefiltered=False, debug=False)
    m2 = torch.zeros(P, 3, device=dev, requires_grad=True)
    color, radii, depth, alpha = rasterize_fused(sc.means3D, m2, sc.opacities, sc.shs, sc.ins_feat, rs, scales=sc.scales,
                                                 rotations=sc.rotations)
    saved = color.grad_fn.saved_tensors
    image, quad_list = saved[14], saved[17]
    gx, gy = (W + 15) // 16, (H + 15) // 16
    tiles = gx * gy
    up = lambda n: (n + 255) // 256 * 256
    ranges = image[: tiles * 8].view(torch.int32).view(tiles, 2).cpu().numpy().astype(np.int64)
    o_nc = up(tiles * 8)
    nc = image[o_nc:o_nc + W * H * 4].view(torch.int32).view(H, W).cpu().numpy().astype(np.int64)
    o_qc = o_nc + up(W * H * 4)
    qc = image[o_qc:o_qc + tiles * 20].view(torch.int32).view(tiles, 5).cpu().numpy().astype(np.int64)
    ql = quad_list.view(torch.int32).cpu().numpy()[16:]          # kQuadPad u32 in front
    ncp = np.zeros((gy * 16, gx * 16), np.int64)
    ncp[:H, :W] = nc
    # deepest last contributor per (tile, quadrant): position in the quadrant stream (1-based)
    hi = ncp.reshape(gy, 2, 8, gx, 2, 8).max(axis=(2, 5)).transpose(0, 2, 1, 3).reshape(tiles, 4)   # q = (qy << 1) | qx
    tot = {"per_quadrant": 0, "horizontal_pairs": 0, "vertical_pairs": 0, "per_tile": 0}
    for t in range(tiles):
        s, n = ranges[t, 0], ranges[t, 1] - ranges[t, 0]
        if n == 0:
            continue
        S = [ql[5 * s + q * n: 5 * s + q * n + min(hi[t, q], qc[t, q])] for q in range(4)]
        tot["per_quadrant"] += sum(len(x) for x in S)
        u = lambda a, b: len(np.union1d(a, b))
        tot["horizontal_pairs"] += u(S[0], S[1]) + u(S[2], S[3])
        tot["vertical_pairs"] += u(S[0], S[2]) + u(S[1], S[3])
        tot["per_tile"] += len(np.unique(np.concatenate(S)))
    out = {"workload": name, **tot, "relative_to_per_quadrant": {k: v / tot["per_quadrant"] for k, v in tot.items()}}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
