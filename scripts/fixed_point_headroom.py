"""Would a ONE-SEGMENT gradient record in int32 fixed point be viable?  (Round 4 analysis, no kernel behind it.)

Round 4's probes say the backward blend is bound by the L2's atomic path and that this path charges per
64-byte segment touched: a record of sixteen 4-byte slots would put `blend_backward_kernel<9>` at 0.51 ms instead of 0.62.  fp32
slots bring back the run-to-run spread; int32 fixed-point slots are order-independent (two's-complement adds commute, and even
intermediate wrap-around is harmless) but need, per Gaussian and slot class, a scale 2^e that the FINAL sum provably never
exceeds.  This script measures how much of the 31 bits such a provable bound costs on the headline scene:

    feature slot c of Gaussian g:   |sum_p w_p g_c(p)|              <=  gmax_c * W_g
    moment slot of degree k:        |sum_p q_p d_p^k|  (centred)    <=  Qmax * o_g * W_g' * (r_g + 1)^k
      with W_g  = min(pixels of the tile rect, o_g * (2 pi sqrt(det Sigma2D) + margin))   (sum of the weights <= sum of alpha)
           Qmax = max over pixels of  2 * cmax * sum_{c < GC} |g_c| + |g_alpha|           (|c_i - R_c| <= 2 cmax, T <= 1, G <= 1)

against the sums the HIP backward actually produced (the fp64 record of one fused 9-channel pass, moments re-centred), and
reports (i) the distribution of log2(bound / |actual|) -- the bits a per-Gaussian power-of-two scale would waste -- and (ii) the
absolute resolution bound_g * 2^-31 of the WORST Gaussian relative to the family's largest actual sum, which is what the 2e-4
parity bar sees.

usage (GPU): python scripts/fixed_point_headroom.py [workload]
"""
import json
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import WORKLOADS  # noqa: E402
from opengaussian_amd import rasterizer as R  # noqa: E402
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
    leaves = {k: getattr(sc, k).detach().clone().requires_grad_(True) for k in ("means3D", "scales", "rotations", "opacities", "shs", "ins_feat")}
    m2 = torch.zeros(P, 3, device=dev, requires_grad=True)
    color, radii, depth, alpha = rasterize_fused(leaves["means3D"], m2, leaves["opacities"], leaves["shs"], leaves["ins_feat"], rs,
                                                 scales=leaves["scales"], rotations=leaves["rotations"])
    gen = torch.Generator().manual_seed(100)
    gC = torch.randn(3, H, W, generator=gen).to(dev)
    gA = torch.randn(1, H, W, generator=gen).to(dev)
    gF = torch.randn(6, H, W, generator=gen).to(dev)
    geom = color.grad_fn.saved_tensors[13]
    R._DEBUG_KEEP_BWD_TMP = []
    torch.autograd.backward([color, alpha], [torch.cat([gC, gF]), gA])
    torch.cuda.synchronize()
    rec = R._DEBUG_KEEP_BWD_TMP[0].view(torch.float64)[: P * 16].view(P, 16).cpu().numpy()
    R._DEBUG_KEEP_BWD_TMP = None
    recv4 = 2 + (9 + 3) // 4
    g = geom[: P * recv4 * 16].view(torch.float32).view(P, recv4 * 4).cpu().numpy()
    px, py, radius = g[:, 0].astype(np.float64), g[:, 1].astype(np.float64), g[:, 3].view(np.int32).astype(np.float64)
    A, B, Cc, opac = (g[:, 4].astype(np.float64), g[:, 5].astype(np.float64), g[:, 6].astype(np.float64), g[:, 7].astype(np.float64))
    vis = radius > 0
    # centred moments from the image-origin sums (what preprocess_bwd does)
    M0, MX, MY, MXX, MXY, MYY = (rec[:, 10 + k] for k in range(6))
    S = np.stack([M0, px * M0 - MX, py * M0 - MY, px * (px * M0 - 2 * MX) + MXX, px * (py * M0 - MY) - py * MX + MXY,
                  py * (py * M0 - 2 * MY) + MYY], 1)
    feats = rec[:, :9]
    # bounds
    det_conic = np.maximum(A * Cc - B * B, 1e-30)
    area_gauss = opac * (2 * math.pi / np.sqrt(det_conic) + 4 * (2 * radius + 1) + 4)       # sum of alpha over the pixel grid, with a perimeter margin
    rect = np.minimum((2 * radius + 1) ** 2, float(W * H))
    Wg = np.minimum(rect, area_gauss)
    gmax = np.concatenate([gC.abs().amax((1, 2)).cpu().numpy(), gF.abs().amax((1, 2)).cpu().numpy()]).astype(np.float64)
    cmax = float(np.abs(g[vis, 8:11]).max())
    Qmax = float((2 * cmax * gC.abs().sum(0) + gA.abs()[0]).max())
    b_feat = gmax[None, :] * Wg[:, None]
    deg = np.array([0, 1, 1, 2, 2, 2], np.float64)
    b_mom = Qmax * opac[:, None] * Wg[:, None] * (radius[:, None] + 1.0) ** deg[None, :]

    def summarize(actual, bound, label):
        a, b = np.abs(actual[vis]), bound[vis]
        nz = a > 0
        waste = np.log2(b[nz] / a[nz])
        fam_max = a.max()
        return {"slot_class": label, "gaussians_with_a_sum": int(nz.sum()),
                "bits_wasted_log2_bound_over_actual": {p: float(np.percentile(waste, q)) for p, q in
                                                       (("p1", 1), ("p10", 10), ("median", 50), ("p90", 90), ("p99", 99), ("max", 100))},
                "bound_violations": int((a > b * (1 + 1e-9)).sum()),
                "worst_absolute_resolution_over_family_max": float((b.max() * 2.0 ** -31) / fam_max),
                "p99_absolute_resolution_over_family_max": float((np.percentile(b, 99) * 2.0 ** -31) / fam_max)}

    out = {"workload": name, "P_visible": int(vis.sum()), "Qmax": Qmax, "cmax": cmax,
           "features": summarize(feats, b_feat, "9 feature slots"),
           "moments_deg0": summarize(S[:, :1], b_mom[:, :1], "S0"),
           "moments_deg1": summarize(S[:, 1:3], b_mom[:, 1:3], "Sx, Sy"),
           "moments_deg2": summarize(S[:, 3:], b_mom[:, 3:], "Sxx, Sxy, Syy"),
           "reading": "a per-Gaussian power-of-two scale 2^31 / bound keeps 31 - bits_wasted bits of the actual sum; the parity bar "
                      "(2e-4 of the family maximum) needs worst_absolute_resolution_over_family_max * sqrt(#contributions) well below 2e-4"}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
