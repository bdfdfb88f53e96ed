"""Per-call latency of small passes (the SAM refiner's single-Gaussian footprint renders,
utils/sam_refinement_utils.py:330-403, and the small subset renders of stages 2.2 / 3): wall time per call, host
enqueue time and GPU time (HIP events), tiny path vs streaming path.  Writes one JSON object to stdout."""
import json
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opengaussian_amd import rasterizer as R
from opengaussian_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
from opengaussian_amd.synthetic import make_camera, make_scene

dev = torch.device("cuda:0")
out = {}
for (W, H, f) in ((1920, 1080, 1000.0), (648, 484, 500.0)):
    cam = make_camera(W, H, f, f).to(dev)
    rs = GaussianRasterizationSettings(H, W, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), torch.zeros(3, device=dev), 1.0,
                                       cam.world_view_transform, cam.full_proj_transform, 3, cam.camera_center, False, False)
    rast = GaussianRasterizer(rs)
    for P in (10000, 1, 64, 256, 1000):       # largest first: measured last in the process, the 10 k case is bimodal (150 / 400 us)
        sc = make_scene(max(P, 2), W, H, f, f, seed=1).to(dev)
        sl = slice(0, P)
        kw = dict(means3D=sc.means3D[sl], opacities=sc.opacities[sl], shs=sc.shs[sl], scales=sc.scales[sl], rotations=sc.rotations[sl])
        for path in (("tiny", "streaming") if P <= 256 else ("streaming",)):
            R.TINY_MAX_P = 256 if path == "tiny" else 0

            def call():
                # exactly the refiner's call: a fresh means2D that requires grad, never back-propagated
                m2 = torch.zeros(P, 3, device=dev, requires_grad=True)
                return rast(means2D=m2, **kw)
            for _ in range(10):
                call()
            torch.cuda.synchronize()
            K = 200
            st0 = dict(R.PASS_STATS)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record()
            for _ in range(K):
                call()
            e1.record()
            t_host = time.perf_counter() - t0
            torch.cuda.synchronize()
            t_all = time.perf_counter() - t0
            out[f"{W}x{H} P={P} {path}"] = {"wall_us_per_call": t_all / K * 1e6, "host_enqueue_us_per_call": t_host / K * 1e6,
                                           "gpu_us_per_call": e0.elapsed_time(e1) / K * 1e3,
                                           "render_phase_sizing": {k: R.PASS_STATS[k] - st0[k] for k in ("blocking", "deferred", "overflow", "tiny")}}
R.TINY_MAX_P = 256
print(json.dumps(out, indent=1))
