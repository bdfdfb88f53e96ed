"""How deep into its sorted list does a tile really blend?  (measure before restructuring the forward.)

The reference's forward fetches a tile's list 256 entries at a time and stops the whole block once every pixel is done
(SURVEY.md section 2.1 `renderCUDA`); the fused pack + blend kernel of round 3 packed the tile's ENTIRE (culled) list before the
first pixel was blended.  For every tile of a bench workload this script reports, from the n_contrib export and the tile ranges
of a default (culled-binning) pass:

    used(t)   = max over the tile's pixels of n_contrib  (1-based position of the last contributor in the tile's culled list)
    len(t)    = length of the tile's culled list
    packed(t) = records the chunked forward really packed before its workgroup-wide exit (qcount[t][4]) -- equals the number of
                records the tile's list keeps when the tile ran to the end of its list

and the work-weighted fractions  sum used / sum len  and  sum packed(t) / sum kept-if-fully-packed(t)  (the second from two passes:
OGS_PACK_FUSED=2 packs every list to the end).

usage: python scripts/list_depth_stats.py [workload ...]      (default: S1M-1080p C3-500k-988 C4-2M-648 C2-100k-800)
"""
import ctypes as C
import json
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import WORKLOADS  # noqa: E402
from opengaussian_amd import _lib  # noqa: E402
from opengaussian_amd.rasterizer import GaussianRasterizationSettings, rasterize_fused  # noqa: E402
from opengaussian_amd.synthetic import make_scene, orbit_camera  # noqa: E402
from tests import helpers  # noqa: E402


def stats(workload, dev):
    wl = WORKLOADS[workload]
    P, W, H, f = wl["P"], wl["W"], wl["H"], wl["f"]
    sc = make_scene(P, W, H, f, f, seed=0).to(dev)
    cam = orbit_camera(W, H, f, f, view_index=0, num_views=8).to(dev)
    rs = GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5),
        bg=torch.zeros(3, device=dev), scale_modifier=1.0, viewmatrix=cam.world_view_transform,
        projmatrix=cam.full_proj_transform, sh_degree=3, campos=cam.camera_center, prefiltered=False, debug=False)
    m2 = torch.zeros(P, 3, device=dev, requires_grad=True)
    color, radii, depth, alpha = rasterize_fused(sc.means3D, m2, sc.opacities, sc.shs, sc.ins_feat, rs, scales=sc.scales,
                                                 rotations=sc.rotations)
    ctx = color.grad_fn
    (_m3, _shs, _cols, _op, _scl, _rot, _cov, _bg, _v, _p, _cp, _radii, _alpha, geom, image, point_list, sorted_rec,
     quad_list) = ctx.saved_tensors
    a = _lib.OgsRasterFwdArgs()
    a.P, a.W, a.H, a.C = ctx.P, W, H, ctx.Cn
    a.geom_buffer, a.image_buffer, a.point_list = _lib.ptr(geom), _lib.ptr(image), _lib.ptr(point_list)
    a.sorted_rec, a.quad_list = _lib.ptr(sorted_rec), _lib.ptr(quad_list)
    keys, ranges, ncontrib, raw = helpers._export_binning_of(a, ctx.num_rendered, point_list, W, H, dev)
    gx, gy = (W + 15) // 16, (H + 15) // 16
    lens = (ranges[:, 1].astype(np.int64) - ranges[:, 0].astype(np.int64))
    tiles = gx * gy
    n_pix = W * H
    off = lambda nbytes: (nbytes + 255) // 256 * 256          # ImageState layout: ranges | n_contrib | qcount | final_T | tile_order
    o_nc = off(tiles * 8)
    o_qc = o_nc + off(n_pix * 4)
    qc = image[o_qc:o_qc + tiles * 5 * 4].view(torch.int32).view(tiles, 5).cpu().numpy().astype(np.int64)
    packed = qc[:, 4]
    pad_h, pad_w = gy * 16, gx * 16
    nc = np.zeros((pad_h, pad_w), np.int64); nc[:H, :W] = ncontrib
    nc_t = nc.reshape(gy, 16, gx, 16).transpose(0, 2, 1, 3).reshape(tiles, 256)
    used = nc_t.max(axis=1)
    nz = lens > 0
    frac = used[nz] / lens[nz]
    out = {
        "workload": workload, "tiles": int(tiles), "tiles_nonempty": int(nz.sum()),
        "num_rendered_full": int(ctx.num_rendered), "culled_list_entries": int(lens.sum()),
        "mean_culled_list": float(lens[nz].mean()), "max_culled_list": int(lens.max()),
        "records_packed_before_the_exit": int(packed.sum()),
        "used_over_len": {"median": float(np.median(frac)), "mean": float(frac.mean()),
                          "p10": float(np.percentile(frac, 10)), "p90": float(np.percentile(frac, 90))},
        "work_weighted": {"sum_used_over_sum_len": float(used[nz].sum() / lens[nz].sum())},
    }
    return out


def main():
    dev = torch.device("cuda:0")
    names = sys.argv[1:] or ["S1M-1080p", "C3-500k-988", "C4-2M-648", "C2-100k-800"]
    res = [stats(n, dev) for n in names]
    mode = os.environ.get("OGS_PACK_FUSED", "1")
    for r in res:
        r["pack_mode"] = {"1": "chunked with workgroup-wide exit (default)", "2": "whole list packed (round 3)",
                          "0": "two launches, whole list packed"}.get(mode, mode)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
