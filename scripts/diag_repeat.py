"""Diagnose the run-to-run spread of the S1M backward (GPUTEST_r01 failure: rotations, 1.05e-4 of max with the fp32 gradient
record of round 1; profiles/r02_grad_accum_f32_vs_f64.json).  The record is fp64 only since round 4 (its moment slots are sums
about the image origin); the script still reports the spread of the record and of every gradient family over repeated runs."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opengaussian_amd import _lib, rasterizer as R
from opengaussian_amd.synthetic import make_camera, make_scene
from tests import helpers

dev = torch.device("cuda:0")
mode = "f64"
P, W, H, f = 1_000_000, 1920, 1080, 1000.0
sc = make_scene(P, W, H, f, f, seed=0).to(dev)
cam = make_camera(W, H, f, f).to(dev)
rs = helpers.settings_for(cam, (0.0, 0.0, 0.0), 3, dev)
g = torch.Generator().manual_seed(3)
g1 = torch.randn(3, H, W, generator=g).to(dev)
names = ("means3D", "scales", "rotations", "opacities", "shs")


def run():
    leaves = {k: getattr(sc, k).detach().clone().requires_grad_(True) for k in names}
    m2 = torch.zeros(P, 3, device=dev, requires_grad=True)
    color, radii, depth, alpha = R.GaussianRasterizer(rs)(means3D=leaves["means3D"], means2D=m2, opacities=leaves["opacities"],
                                                          shs=leaves["shs"], scales=leaves["scales"], rotations=leaves["rotations"])
    R._DEBUG_KEEP_BWD_TMP = []
    torch.autograd.backward([color], [g1])
    torch.cuda.synchronize()
    tmp = R._DEBUG_KEEP_BWD_TMP[0]
    R._DEBUG_KEEP_BWD_TMP = None
    rec = tmp.view(torch.float64 if mode == "f64" else torch.float32)[: P * 16].view(P, 16).double().cpu().numpy()
    return {k: v.grad.cpu().numpy() for k, v in leaves.items()} | {"means2D": m2.grad.cpu().numpy()}, rec, radii.cpu().numpy()


a, ra, radii = run()
b, rb, _ = run()
out = {"mode": mode}
for k in a:
    scale = np.abs(a[k]).max()
    d = np.abs(a[k] - b[k])
    out[k] = {"max": float(scale), "max_diff": float(d.max()), "rel": float(d.max() / scale)}
drec = np.abs(ra - rb)
out["rec_slot_maxdiff"] = drec.max(0).tolist()
out["rec_slot_max"] = np.abs(ra).max(0).tolist()
i = int(np.abs(a["rotations"] - b["rotations"]).max(1).argmax())
out["worst_rot_gaussian"] = {"idx": i, "radius": int(radii[i]), "rot_a": a["rotations"][i].tolist(), "rot_b": b["rotations"][i].tolist(),
                             "rec_a": ra[i].tolist(), "rec_b": rb[i].tolist()}
# timing of the two backward kernels
_lib.prof_enable(1)
for _ in range(10):
    run_leaves = {k: getattr(sc, k).detach().clone().requires_grad_(True) for k in names}
    m2 = torch.zeros(P, 3, device=dev, requires_grad=True)
    color, *_ = R.GaussianRasterizer(rs)(means3D=run_leaves["means3D"], means2D=m2, opacities=run_leaves["opacities"],
                                         shs=run_leaves["shs"], scales=run_leaves["scales"], rotations=run_leaves["rotations"])
    torch.autograd.backward([color], [g1])
torch.cuda.synchronize()
prof = _lib.prof_collect()
_lib.prof_enable(0)
out["kernels_ms"] = {k: v["total_ms"] / v["calls"] for k, v in prof.items() if "backward" in k}
print(json.dumps(out, indent=1))
