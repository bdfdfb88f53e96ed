"""Dev helper: is a fused fwd+bwd step host-bound?  Prints, per workload, the host time to enqueue a step (forward call,
backward call, the wait on the geometry phase inside the forward) against the GPU time, and a cProfile of the hottest host
functions.  usage: python scripts/host_overhead.py [C2-100k-800 C3-500k-988 ...] > profiles/r03_c2_c3_host.json"""
import cProfile, io, json, math, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from opengaussian_amd import rasterizer as R
from opengaussian_amd.rasterizer import GaussianRasterizationSettings, rasterize_fused
from opengaussian_amd.synthetic import make_scene, make_camera
WL = {"S1M-1080p": (1_000_000, 1920, 1080, 1000.0), "C2-100k-800": (100_000, 800, 800, 700.0),
      "C3-500k-988": (500_000, 988, 731, 800.0), "C4-2M-648": (2_000_000, 648, 484, 500.0)}
dev = torch.device("cuda:0")
out = {}
for name in (sys.argv[1:] or ["C2-100k-800", "C3-500k-988"]):
    P, W, H, f = WL[name]
    sc = make_scene(P, W, H, f, f).to(dev); cam = make_camera(W, H, f, f).to(dev)
    rs = GaussianRasterizationSettings(H, W, math.tan(cam.FoVx/2), math.tan(cam.FoVy/2), torch.zeros(3, device=dev), 1.0,
                                       cam.world_view_transform, cam.full_proj_transform, 3, cam.camera_center, False, False)
    leaves = {k: getattr(sc, k).requires_grad_(True) for k in ("means3D", "scales", "rotations", "opacities", "shs", "ins_feat")}
    g = torch.Generator().manual_seed(0)
    gCF = torch.randn(9, H, W, generator=g).to(dev); gA = torch.randn(1, H, W, generator=g).to(dev)
    def step():
        for v in leaves.values(): v.grad = None
        m2 = torch.zeros(P, 3, device=dev, requires_grad=True)
        t0 = time.perf_counter()
        c, r, d, a = rasterize_fused(leaves["means3D"], m2, leaves["opacities"], leaves["shs"], leaves["ins_feat"], rs,
                                     scales=leaves["scales"], rotations=leaves["rotations"])
        t1 = time.perf_counter()
        torch.autograd.backward([c, a], [gCF, gA])
        t2 = time.perf_counter()
        return t1 - t0, t2 - t1
    for _ in range(5): step()
    torch.cuda.synchronize()
    K = 200
    t0 = time.perf_counter(); tf = tb = 0
    for _ in range(K):
        a, b = step(); tf += a; tb += b
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    # GPU-only time of the same steps: events around a batch of steps, host far ahead is impossible (the forward waits for
    # its geometry phase), so take the kernel sum from the library's per-launch events instead
    from opengaussian_amd import _lib
    _lib.prof_enable(1)
    for _ in range(20): step()
    torch.cuda.synchronize()
    prof = _lib.prof_collect(); _lib.prof_enable(0)
    ksum = sum(v["total_ms"] for v in prof.values()) / 20
    launches = sum(v["calls"] for v in prof.values()) / 20
    pr = cProfile.Profile(); pr.enable()
    for _ in range(100): step()
    pr.disable(); torch.cuda.synchronize()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14)
    top = [ln.strip() for ln in s.getvalue().splitlines() if ln.strip() and (ln.strip()[0].isdigit())][:14]
    out[name] = {"P": P, "W": W, "H": H, "ms_per_step_wall": t_all / K * 1e3, "host_enqueue_ms_per_step": t_enq / K * 1e3,
                 "forward_call_ms": tf / K * 1e3, "backward_call_ms": tb / K * 1e3, "kernel_sum_ms": ksum,
                 "kernel_launches_per_step": launches, "pass_stats": dict(R.PASS_STATS), "cprofile_top_tottime_100_steps": top}
print(json.dumps(out, indent=1))
