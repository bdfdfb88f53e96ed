"""How deep into its sorted list does a tile really blend?  (measure before restructuring the forward.)

The reference's forward fetches a tile's list 256 entries at a time and stops the whole block once every pixel is done
(SURVEY.md section 2.1 `renderCUDA`); the fused pack + blend kernel of round 3 packed the tile's ENTIRE (culled) list before the
first pixel was blended.  For every tile of a bench workload this script reports, from the n_contrib export and the tile ranges
of a default (culled-binning) pass:

    used(t) = max over the tile's pixels of n_contrib  (1-based position of the last contributor in the tile's culled list)
    len(t)  = length of the tile's culled list
    chunked(t) = min(len(t), ceil(used_exit(t) / 256) * 256)   what a 256-entry-chunked pack would touch, where used_exit is the
                 list position at which the LAST pixel of the tile saturates (T < 1e-4) or the list ends -- a tile with one
                 unsaturated pixel has to walk its whole list, whatever its other pixels do

and the work-weighted fractions  sum used / sum len,  sum chunked / sum len.

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
    # per tile: deepest last contributor, and whether every pixel of the tile saturated (final T < 1e-4 <=> the walk stopped early)
    tiles = gx * gy
    n_pix = W * H
    final_T = image.view(torch.float32)          # ImageState layout: ranges | n_contrib | qcount | final_T | tile_order (256-B aligned)
    off = lambda nbytes: (nbytes + 255) // 256 * 256
    o_nc = off(tiles * 8)
    o_qc = o_nc + off(n_pix * 4)
    o_ft = o_qc + off(tiles * 5 * 4)
    fT = image[o_ft:o_ft + n_pix * 4].view(torch.float32).view(H, W).cpu().numpy()
    pad_h, pad_w = gy * 16, gx * 16
    nc = np.zeros((pad_h, pad_w), np.int64); nc[:H, :W] = ncontrib
    sat = np.ones((pad_h, pad_w), bool); sat[:H, :W] = fT < 1e-4       # pixels outside the image never hold a tile back
    nc_t = nc.reshape(gy, 16, gx, 16).transpose(0, 2, 1, 3).reshape(tiles, 256)
    sat_t = sat.reshape(gy, 16, gx, 16).transpose(0, 2, 1, 3).reshape(tiles, 256)
    used = nc_t.max(axis=1)
    all_sat = sat_t.all(axis=1)
    # a tile exits early only when ALL its pixels are saturated; the exit position is then a little past the deepest
    # contributor (the entry that saturates a pixel is not applied): used + 1 is a lower bound, used itself is reported
    exit_pos = np.where(all_sat, np.minimum(used + 1, lens), lens)
    chunked = np.minimum(lens, (exit_pos + 255) // 256 * 256)
    nz = lens > 0
    frac = used[nz] / lens[nz]
    out = {
        "workload": workload, "tiles": int(tiles), "tiles_nonempty": int(nz.sum()),
        "num_rendered_full": int(ctx.num_rendered), "culled_list_entries": int(lens.sum()),
        "mean_culled_list": float(lens[nz].mean()), "max_culled_list": int(lens.max()),
        "tiles_fully_saturated_frac": float(all_sat[nz].mean()),
        "used_over_len": {"median": float(np.median(frac)), "mean": float(frac.mean()),
                          "p10": float(np.percentile(frac, 10)), "p90": float(np.percentile(frac, 90))},
        "work_weighted": {"sum_used_over_sum_len": float(used[nz].sum() / lens[nz].sum()),
                          "sum_exit_over_sum_len": float(exit_pos[nz].sum() / lens[nz].sum()),
                          "sum_chunked256_over_sum_len": float(chunked[nz].sum() / lens[nz].sum())},
    }
    return out


def main():
    dev = torch.device("cuda:0")
    names = sys.argv[1:] or ["S1M-1080p", "C3-500k-988", "C4-2M-648", "C2-100k-800"]
    res = [stats(n, dev) for n in names]
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
