"""Generate tests/golden/ref_fragments.npz by RUNNING the reference's own restatements of rasterizer
sub-formulas on the CPU (build container only; the reference never travels, only these vectors do).

The rasterizer itself is absent from /root/reference, but these in-reference fragments pin pieces of the
oracle:
  * utils/sh_utils.py:57-112 eval_sh, with the +0.5 / clamp_min(0) of gaussian_renderer/__init__.py:96-97
  * utils/general_utils.py:64-110 + scene/gaussian_model.py:41-45: R(q), L = R S, Sigma = L L^T, packing
  * utils/graphics_utils.py:54-74 getProjectionMatrix, :38-52 getWorld2View2, :22-29 geom_transform_points
  * scene/cameras.py:71-78: world_view_transform / full_proj_transform / camera_center conventions
Modules are loaded by file path with device="cuda" mapped to the CPU.
"""
import importlib.util
import math
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def _load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


def _cpu_shim():
    torch.Tensor.cuda = lambda self, *a, **k: self
    for fn in ("zeros", "ones", "tensor", "empty", "rand", "full"):
        orig = getattr(torch, fn)

        def wrap(*a, __orig=orig, **k):
            if str(k.get("device", "")).startswith("cuda"):
                k["device"] = "cpu"
            return __orig(*a, **k)
        setattr(torch, fn, wrap)
    real_device = torch.device

    class _Dev:
        def __new__(cls, *a, **k):
            if a and str(a[0]).startswith("cuda"):
                return real_device("cpu")
            return real_device(*a, **k)
    torch.device = _Dev


def main():
    if not os.path.isdir(REF):
        sys.exit("reference not mounted: goldens can only be (re)generated in the build container")
    _cpu_shim()
    pkg = types.ModuleType("utils"); pkg.__path__ = [os.path.join(REF, "utils")]
    sys.modules["utils"] = pkg
    sh = _load("utils.sh_utils", "utils/sh_utils.py")
    gu = _load("utils.graphics_utils", "utils/graphics_utils.py")
    ge = _load("utils.general_utils", "utils/general_utils.py")
    cams = _load("ref_cameras", "scene/cameras.py")

    g = torch.Generator().manual_seed(7)
    P = 257
    out = {}
    # --- SH ---
    shs = torch.randn(P, 16, 3, generator=g) * 0.4
    pos = torch.randn(P, 3, generator=g) * 3
    campos = torch.tensor([0.3, -0.2, 0.5])
    d = pos - campos
    d = d / d.norm(dim=1, keepdim=True)
    for deg in range(4):
        rgb = torch.clamp_min(sh.eval_sh(deg, shs.transpose(1, 2), d) + 0.5, 0.0)
        out[f"sh_rgb_deg{deg}"] = rgb.numpy()
    out.update(sh_shs=shs.numpy(), sh_pos=pos.numpy(), sh_campos=campos.numpy())
    # --- covariance ---
    scales = torch.exp(torch.randn(P, 3, generator=g) - 2)
    q = torch.randn(P, 4, generator=g)
    qn = q / q.norm(dim=1, keepdim=True)
    for mod in (1.0, 0.7):
        L = ge.build_scaling_rotation(mod * scales, qn)
        cov = ge.strip_symmetric(L @ L.transpose(1, 2))
        out[f"cov3d_mod{mod}"] = cov.numpy()
    out["rot_matrix"] = ge.build_rotation(qn).numpy()
    out.update(cov_scales=scales.numpy(), cov_quat=qn.numpy())
    # --- camera conventions ---
    ang = 0.3
    R = np.array([[math.cos(ang), 0, math.sin(ang)], [0, 1, 0], [-math.sin(ang), 0, math.cos(ang)]])
    T = np.array([0.2, -0.1, 1.5])
    fovx, fovy = 1.1, 0.7
    H, W = 60, 100
    cam = cams.Camera(colmap_id=0, R=R, T=T, FoVx=fovx, FoVy=fovy, cx=W / 2, cy=H / 2, image=torch.zeros(3, H, W),
                      depth=None, gt_alpha_mask=None, gt_sam_mask=None, gt_mask_feat=None, image_name="x", uid=0,
                      data_device="cpu")
    out.update(cam_R=R, cam_T=T, cam_fov=np.array([fovx, fovy]), cam_wh=np.array([W, H]),
               cam_world_view=cam.world_view_transform.numpy(), cam_full_proj=cam.full_proj_transform.numpy(),
               cam_center=cam.camera_center.numpy(), cam_proj=cam.projection_matrix.numpy())
    out["proj_matrix"] = gu.getProjectionMatrix(0.01, 100.0, fovx, fovy).numpy()
    # --- homogeneous divide ---
    pts = torch.randn(P, 3, generator=g)
    out["geom_pts"] = pts.numpy()
    out["geom_ndc"] = gu.geom_transform_points(pts, cam.full_proj_transform).numpy()
    path = os.path.join(HERE, "ref_fragments.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
