#!/usr/bin/env python
"""bench.py -- fwd+bwd Mpix/s of the MI355X rasterizer hot path on the BASELINE.json workload.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json `metric`): synthetic S1M-1080p scene -- 1,000,000 Gaussians,
1920x1080, seed 0 -- one "step" = one VIEW rendered forward AND backward with everything the metric names:
  (A) RGB through SH degree 3 (+depth, alpha), backward to means3D / scales / rotations / opacity / SH /
      means2D  (stage-0 style, all gradient families),
  (B) the 6-D ins_feat map (the reference needs two 3-channel passes, gaussian_renderer/__init__.py:129-151),
      backward to ins_feat only (stage-1 style: geometry detached, train.py:431-436).
Default: (A) and (B) are rendered by ONE fused 9-channel rasterizer pass (rasterize_fused; the backward kernel
keeps the feature loss out of the geometry gradients, so the gradients equal those of two separate passes --
tests/test_10_raster_gpu.py::test_fused_pass_equals_separate_passes).  --separate-passes runs them as two passes
(3-channel SH + 6-channel), --rgb-only times (A) alone.
Inputs are resident in HBM before the timed region.  value = views*W*H / time, summed over all ranks
(weak scaling: rank r renders its own view of the same replicated scene; for N > 1 every step exchanges the
per-Gaussian gradients over RCCL -- one flat SUM all-reduce + an all-gather of the rank-1 factor of the SH
gradient + a MAX all-reduce of the radii, opengaussian_amd/dp.py -- by default overlapped with the next view's
render (--exchange pipelined; every exchange still completes inside the timed region), --exchange sync waits for
each step's own exchange).

One JSON line on rank 0, with `roofline` (dominant kernel, HIP events on the launch stream, algorithmic
bytes of SURVEY.md section 8(d)) and `cpu_baseline` (the CPU oracle = pure-PyTorch per-tile alpha blend,
bounded sample, rank 0 at N = 1 only).
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec (guides/MI355X_MICROARCH.md); ~6300 achievable
# VALU issue peak, MEASURED on this part (scripts/ubench/valu_issue2.hip -> profiles/r03_valu_issue_price_list.json):
# a SIMD issues one wave64 vector instruction per 2 cycles when >= 4 waves are resident AND every operand is a VGPR /
# inline constant / literal (v_fma, v_mul, v_add, v_sub, v_mov, v_add_u32, v_and); per 4 cycles for ANY form with an SGPR
# operand, DPP, v_cmp*, v_cndmask (SGPR-pair mask), v_max / v_min, shifts, v_cvt, v_pk_*, f64, v_readlane / v_writelane;
# per 8 for v_exp / v_rcp.  `peak` below is the 2-cycle rate; `roofline_valu.issue_model` prices the dominant kernel's
# own static instruction mix (scripts/isa_issue_mix.py -> profiles/r04_issue_mix.json).
VALU_PEAK_GINST = 256 * 4 * 2.4 / 2
MFMA_BF16_PEAK_TFLOPS = 2500.0      # dense bf16 (guide); k-means scores + one-hot accumulate run there


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--views", type=int, default=8,
                    help="cameras of the orbit the steps cycle through (multi-view training: num_rendered changes from "
                         "step to step, so the sync-free render phase's capacity hint is exercised); 1 = the same view "
                         "every step")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the untimed extra measurements (stage-1 features-only pass, per-kernel breakdown pass)")
    ap.add_argument("--workload", default="S1M-1080p", choices=list(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-tile-stride", type=int, default=8, help="CPU baseline renders this many full tile rows")
    ap.add_argument("--no-kmeans", action="store_true")
    ap.add_argument("--no-extra-workloads", action="store_true",
                    help="skip the C2 / C3 / C4 lines (`extras` in the JSON: the other single-GPU configs of BASELINE.json, each "
                         "measured by a child run of this script after the headline)")
    ap.add_argument("--rgb-only", action="store_true", help="time pass A only (BASELINE.md row 'RGB')")
    ap.add_argument("--exchange", default="pipelined", choices=["pipelined", "sync"],
                    help="N > 1: 'pipelined' = the gradient exchange of view i overlaps the render of view i+1 on the "
                         "same rank (double-buffered buckets; every exchange still completes inside the timed "
                         "region); 'sync' = each step waits for its own exchange before the next render starts")
    ap.add_argument("--dense-sh-allreduce", action="store_true",
                    help="N > 1: all-reduce the dense [P,16,3] SH gradient instead of all-gathering its rank-1 factor")
    ap.add_argument("--stage1", action="store_true",
                    help="time the STAGE-1 training step instead of the all-gradient step: everything but ins_feat detached "
                         "(train.py:431-436) -> fused 9-channel forward + features-only backward; for N > 1 the only exchange "
                         "is the SUM all-reduce of dL/d ins_feat (24 B per Gaussian).  Without this flag an N > 1 run still "
                         "reports that step as the untimed extra `dist.stage1` (both curves from the driver's one command)")
    ap.add_argument("--separate-passes", action="store_true",
                    help="render RGB and the 6-ch ins_feat map as two rasterizer passes (reference structure) "
                         "instead of the single fused 9-channel pass")
    return ap.parse_args()


WORKLOADS = {
    "S1M-1080p": dict(P=1_000_000, W=1920, H=1080, f=1000.0),
    "C2-100k-800": dict(P=100_000, W=800, H=800, f=700.0),
    "C3-500k-988": dict(P=500_000, W=988, H=731, f=800.0),      # LeRF-teatime-class image and point count
    "C4-2M-648": dict(P=2_000_000, W=648, H=484, f=500.0),      # ScanNet-class: many Gaussians, small image
    # the same model seen from INSIDE: three quarters of it behind the camera (a room-scale scan: a view sees a fraction of the
    # model; the synthetic scenes above put every Gaussian in front of the camera).  Not a BASELINE.json config: an `extras`
    # line for the per-Gaussian kernels and the depth sort, whose work should follow what is visible
    "C4-2M-648-inside": dict(P=2_000_000, W=648, H=484, f=500.0, behind=0.75),
}


def algorithmic_bytes(P, D, npx, C, K_in, S=6):
    """SURVEY.md section 8(d).  Returns dict of per-kernel and whole-pass algorithmic bytes."""
    b = {}
    b["preprocess_kernel"] = P * (44 + 4 * K_in + 52)
    b["binning"] = D * (12 + 24 * S + 8)
    b["blend_forward_kernel"] = D * (28 + 4 * C + 4) + npx * (4 * C + 12)
    # the fused pack + blend kernel does what the reference's forward blend does per list entry (gather the Gaussian's
    # data, blend it): same algorithmic bytes; the stand-alone per-block blend kernel likewise
    b["pack_blend_chunked_kernel"] = b["pack_blend_forward_kernel"] = b["blend_forward_rows_kernel"] = b["blend_forward_kernel"]
    b["blend_backward_kernel"] = npx * (4 * C + 16) + D * (28 + 4 * C + 4) + D * 2 * (4 * C + 28)
    # features-only backward (stage >= 1): F0 = 3 channels come from the (detached) SH colours in a fused pass
    F0 = 3 if C > 3 else 0
    b["blend_backward_feat_lds_kernel"] = b["blend_backward_feat_kernel"] = (
        npx * (4 * (C - F0) + 4) + D * (32 + 4 * C) + 2 * D * 4 * (C - F0))
    b["preprocess_backward_kernel"] = P * (44 + 4 * K_in + 4 + 4 * C + 28) + P * (40 + 4 * K_in)
    b["fwd"] = b["preprocess_kernel"] + b["binning"] + b["blend_forward_kernel"]
    b["bwd"] = b["blend_backward_kernel"] + b["preprocess_backward_kernel"]
    return b


def cpu_baseline(scene, cam, W, H, f, stride, rgb_only):
    """Pure-PyTorch CPU alpha blend (the oracle) fwd + autograd bwd on every `stride`-th tile of the same
    scene; preprocess + binning (NumPy) are run in full and charged pro rata."""
    import dataclasses
    import numpy as np
    from oracle import raster_oracle as ro
    # a one-GPU box owns a 16-core share of a much larger host: never oversubscribe past it
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    ncores = max(1, min(16, avail))
    torch.set_num_threads(ncores)
    torch.set_flush_denormal(True)        # exp() underflow denormals otherwise dominate the CPU time
    tanx, tany = W / (2 * f), H / (2 * f)
    t0 = time.time()
    g = ro.preprocess(scene.means3D.numpy(), scene.opacities.numpy(), cam.world_view_transform.numpy(),
                      cam.full_proj_transform.numpy(), cam.camera_center.numpy(), W, H, tanx, tany,
                      scales=scene.scales.numpy(), rotations=scene.rotations.numpy(), shs=scene.shs.numpy(), sh_degree=3)
    t_pre = time.time() - t0
    # bounded sample: a band of `rows` full tile rows through the middle of the image, rendered as its own
    # (W x rows*16) image over the Gaussians whose tile rect reaches the band
    gx, gy = (W + 15) // 16, (H + 15) // 16
    rows = max(1, min(gy, stride))
    y0 = (gy - rows) // 2
    sel = np.nonzero((g.radii > 0) & (g.rect_min[:, 1] < y0 + rows) & (g.rect_max[:, 1] > y0))[0]
    rmin = g.rect_min[sel].copy(); rmax = g.rect_max[sel].copy()
    rmin[:, 1] = np.clip(rmin[:, 1] - y0, 0, rows); rmax[:, 1] = np.clip(rmax[:, 1] - y0, 0, rows)
    xy = g.xy[sel].copy(); xy[:, 1] -= y0 * 16
    sub = dataclasses.replace(
        g, depth=g.depth[sel], radii=g.radii[sel], xy=xy, conic=g.conic[sel], opacity=g.opacity[sel], rgb=g.rgb[sel],
        clamped=g.clamped[sel], rect_min=rmin, rect_max=rmax, cov3D=g.cov3D[sel],
        tiles_touched=((rmax[:, 0] - rmin[:, 0]) * (rmax[:, 1] - rmin[:, 1])).astype(np.uint32))
    Hb = rows * 16
    t0 = time.time()
    b = ro.bin_tiles(sub, W, Hb)
    t_bin = time.time() - t0
    gen = torch.Generator().manual_seed(1)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    t0 = time.time()
    passes = [(t(sub.rgb), 3)] + ([] if rgb_only else [(scene.ins_feat[torch.from_numpy(sel)].clone(), 6)])
    for feats, C in passes:
        log(f"cpu baseline: {C}-channel blend fwd+bwd over {gx * rows} tiles on {ncores} threads")
        leaves = [t(sub.xy).requires_grad_(True), t(sub.conic).requires_grad_(True), t(sub.opacity).requires_grad_(True),
                  feats.requires_grad_(True), t(sub.depth).requires_grad_(True)]
        color, depth, alpha, _ = ro.blend(*leaves, b.ranges, b.point_list, W, Hb, torch.zeros(C))
        gC = torch.randn(color.shape, generator=gen)
        gA = torch.randn(alpha.shape, generator=gen)
        torch.autograd.backward([color, alpha], [gC, gA])
    t_blend = time.time() - t0
    frac = rows / gy
    px = W * Hb
    total = t_blend + t_bin + t_pre * frac
    return {"value": px / total / 1e6, "unit": "Mpix/s", "cores": ncores, "kind": "port",
            "sample": (f"oracle/raster_oracle.py (pure-PyTorch per-tile alpha blend fwd + autograd bwd, fp32, flush-denormal) on a "
                       f"band of {rows} full tile rows ({W}x{Hb} px, {len(sel)} Gaussians, D={b.num_rendered}) of the same scene: "
                       f"blend {t_blend:.1f} s + NumPy binning of the band {t_bin:.1f} s + NumPy preprocess of all "
                       f"{scene.means3D.shape[0]} Gaussians charged pro rata ({t_pre:.1f} s x {frac:.3f}); "
                       f"per-Gaussian preprocess backward not included")}


def kmeans_bench(device):
    """k-means it/s (second half of BASELINE.json's metric): N=2M, d=9 (6 feat + 3 xyz), k=64, 5 Lloyd
    iterations + re-assignment per call (ScanNet config C4, scene/kmeans_quantize.py:146-241)."""
    from opengaussian_amd.kmeans import lloyd
    g = torch.Generator().manual_seed(0)
    N, d, k, iters = 2_000_000, 9, 64, 5
    feat = torch.cat([torch.rand(N, 6, generator=g), torch.randn(N, 3, generator=g)], dim=1).to(device)
    cent = feat[torch.randperm(N, generator=g)[:k].to(device)].clone()
    for _ in range(2):
        lloyd(feat, cent.clone(), iters=iters, nchunks=N // 10000 + 1)
    torch.cuda.synchronize()
    reps = 10
    t0 = time.time()
    for _ in range(reps):
        lloyd(feat, cent.clone(), iters=iters, nchunks=N // 10000 + 1)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / reps
    bytes_iter = N * (4 * d + 8) + 8 * k * d
    out = {"it_per_s": iters / dt, "ms_per_call": dt * 1e3, "N": N, "d": d, "k": k, "iters_per_call": iters,
           "algorithmic_GBps": bytes_iter * iters / dt / 1e9}
    # per-kernel times (HIP events around every launch, separate run) and the two rooflines of the dominant pass
    from opengaussian_amd import _lib
    _lib.prof_filter("kmeans_")
    _lib.prof_enable(1)
    for _ in range(5):
        lloyd(feat, cent.clone(), iters=iters, nchunks=N // 10000 + 1)
    torch.cuda.synchronize()
    prof = _lib.prof_collect()
    _lib.prof_enable(0)
    _lib.prof_filter("blend_")
    kern = {kname: v["total_ms"] / v["calls"] * 1e3 for kname, v in prof.items()}
    out["kernels_us_per_launch"] = kern
    acc = next((v for kname, v in kern.items() if "accum" in kname), None)
    if acc:
        # accumulate pass: reads N*4*d bytes; matrix work = scores (2 * 64 K-slots per point and centre) + one-hot accumulate
        # (3 bf16 terms x 2 * 16 columns per point and centre)
        flops = N * k * (2 * 64 + 3 * 2 * 16)
        out["roofline"] = {"kernel": "kmeans accumulate pass (scores + one-hot accumulate on the bf16 matrix path)",
                           "bound": "hbm", "achieved": N * 4 * d / (acc * 1e-6) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                           "frac": N * 4 * d / (acc * 1e-6) / 1e9 / HBM_PEAK_GBPS, "avg_launch_us": acc,
                           "matrix_pipe": {"achieved_TFLOPs": flops / (acc * 1e-6) / 1e12, "peak_TFLOPs": MFMA_BF16_PEAK_TFLOPS,
                                           "frac": flops / (acc * 1e-6) / 1e12 / MFMA_BF16_PEAK_TFLOPS},
                           "limiter": "vector issue of the operand packing / argmax around the MFMAs (DESIGN.md section 4)"}
    return out


def kmeans_leaf_bench(device):
    """Fine level of the two-level codebook (BASELINE config 4: "coarse+fine k-means"): after one coarse assignment of the
    N = 2 M / k1 = 64 workload, ALL 64 leaf assignments -- Quantize_kMeans.cluster_assign(mode="leaf", selected_leaf=c) for
    c = 0..63, k2 = 5 (scripts/train_scannet.sh:42-43), d = 6, 5 Lloyd iterations each on the ~31 k points of coarse cluster
    c (scene/kmeans_quantize.py:195-206,232-240; train.py:321-330 calls one of them every 50 iterations of stage 2.2).
    Reported: Lloyd it/s over the 64 calls, ms per leaf assign, and what the reference's per-call rebuild of
    equalize_cluster_size's index table (scene/kmeans_quantize.py:89-144) would add (the table is lazy here)."""
    from opengaussian_amd.kmeans import Quantize_kMeans
    g = torch.Generator().manual_seed(0)
    N, k1, k2, iters = 2_000_000, 64, 5, 5
    feat6 = torch.rand(N, 6, generator=g).to(device)
    xyz = torch.randn(N, 3, generator=g).to(device)
    q = Quantize_kMeans(num_clusters=k1, num_leaf_clusters=k2, num_iters=iters, dim=9, dim_leaf=6)
    q.cluster_assign(torch.cat([feat6, xyz], dim=1), mode="root")
    q.iLeafSubNum = torch.full((k1,), k2, dtype=torch.int64, device=device)

    def sweep(eager=False):
        for c in range(k1):
            q.cluster_assign(feat6, mode="leaf", selected_leaf=c)
            if eager:
                q.cluster_ids                 # forces the padded index table, as the reference rebuilds it on every assign
    sweep()                                   # warm-up: leaf centres initialised, allocator warm
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sweep()
    torch.cuda.synchronize()
    t_lazy = time.perf_counter() - t0
    sweep(eager=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sweep(eager=True)
    torch.cuda.synchronize()
    t_eager = time.perf_counter() - t0
    pts = [int((q.cls_ids == c).sum()) for c in (0, k1 // 2, k1 - 1)]
    return {"leaf_assigns": k1, "k2": k2, "d": 6, "iters_per_assign": iters, "points_per_coarse_cluster_sample": pts,
            "ms_per_leaf_assign": t_lazy / k1 * 1e3, "it_per_s": k1 * iters / t_lazy,
            "ms_per_leaf_assign_with_eager_index_table": t_eager / k1 * 1e3,
            "equalize_cluster_size_share_when_eager": max(0.0, 1.0 - t_lazy / t_eager),
            "note": "equalize_cluster_size's padded index table (scene/kmeans_quantize.py:89-144: rebuilt over all N points on "
                    "every assign, read by nothing in the training loop) is built lazily on first access; the eager figure forces "
                    "it after every assign.  A leaf assign is ~10 launches on ~31 k rows plus the boolean-index gather / scatter "
                    "of the coarse cluster's points (as the reference does): host-paced"}


def kmeans_cpu_baseline():
    """The k-means oracle (NumPy restatement of scene/kmeans_quantize.py:146-241, pinned by the reference goldens) on
    a bounded sample: N = 200k rows of the same distribution, d = 9, k = 64, 5 Lloyd iterations."""
    import numpy as np
    from oracle import kmeans_oracle as ko
    g = torch.Generator().manual_seed(0)
    N, k, iters = 200_000, 64, 5
    feat = torch.cat([torch.rand(N, 6, generator=g), torch.randn(N, 3, generator=g)], dim=1).numpy()
    cent = feat[:k].copy()
    t0 = time.time()
    ko.lloyd(feat, cent, iters=iters, nchunks=N // 10000 + 1)
    dt = time.time() - t0
    return {"value": iters / dt, "unit": "Lloyd it/s", "cores": 1, "kind": "port",
            "sample": f"oracle/kmeans_oracle.py lloyd, N={N} (1/10 of the GPU workload's rows), d=9, k={k}, {iters} iterations "
                      f"+ re-assignment in {dt:.2f} s; per-row cost is size independent, so the 2M-row rate is ~1/10 of this"}


def stage1_bench(leaves, all_settings, gF, P, device, reps=40):
    """Stage >= 1 training step of the reference (train.py:431-436: everything but ins_feat detached): RGB + 6-D
    ins_feat + silhouette in ONE fused 9-channel forward, features-only backward (dL/d ins_feat alone).  Untimed
    extra; reported beside the headline (which is the all-gradient stage-0 style step)."""
    from opengaussian_amd import _lib
    from opengaussian_amd.rasterizer import rasterize_fused
    det = {k: v.detach() for k, v in leaves.items()}
    feat = leaves["ins_feat"].detach().clone().requires_grad_(True)
    gC9 = torch.cat([torch.zeros(3, *gF.shape[1:], device=device), gF])

    V = len(all_settings)

    def one(i, kept=False):
        m2 = torch.zeros(P, 3, device=device)                     # no grad: nothing consumes dL/dmeans2D after stage 0
        color, radii, depth, alpha = rasterize_fused(det["means3D"], m2, det["opacities"], det["shs"], feat,
                                                     all_settings[i % V], scales=det["scales"], rotations=det["rotations"],
                                                     frozen_key=(("bench", i % V), "frozen", None) if kept else None)
        feat.grad = None
        color.backward(gC9)

    def timed(kept):
        for i in range(max(5, V)):                                # kept: one full pass per camera fills the cache
            one(i, kept)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(reps):
            one(i, kept)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        _lib.prof_enable(1)
        for i in range(V + 2):
            one(i, kept)
        torch.cuda.synchronize()
        prof = _lib.prof_collect()
        _lib.prof_enable(0)
        return dt, prof

    dt, prof = timed(False)
    W, H = gF.shape[2], gF.shape[1]
    out = {"ms_per_step": dt * 1e3, "Mpix_per_s": W * H / dt / 1e6,
           "kernels_ms": {k: v["total_ms"] / v["calls"] for k, v in prof.items() if "backward" in k or "blend" in k}}
    # the same step with the camera's pass KEPT (rasterizer.KEPT_PASSES: frozen geometry, stage-1 calls draw no rescale,
    # train.py:346-350): forward = ONE blend launch over the kept streams, feature channels read from the current tensor
    try:
        from opengaussian_amd import rasterizer as R
        saved, R.KEPT_PASSES = R.KEPT_PASSES, R.KeptPasses(budget_bytes=24 << 30)
        try:
            n0 = R.PASS_STATS["reblend"]
            dtk, profk = timed(True)
            kp = R.KEPT_PASSES
            out["kept_pass"] = {"ms_per_step": dtk * 1e3, "Mpix_per_s": W * H / dtk / 1e6,
                                "kernels_ms_per_launch": {k: v["total_ms"] / v["calls"] for k, v in profk.items()},
                                "views_kept": len(kp.slots), "kept_bytes_per_view": kp.nbytes // max(len(kp.slots), 1),
                                "reblends": R.PASS_STATS["reblend"] - n0, "cache": dict(kp.stats),
                                "note": "every timed step is a hit: 8 cameras cycled, all kept after the warm-up sweep"}
        finally:
            R.KEPT_PASSES = saved
    except Exception as e:  # noqa: BLE001
        out["kept_pass"] = {"error": repr(e)}
    return out


def frozen_rgb_bench(scene, cams, W, H, device, reps=32):
    """A call in which nothing trains (every stage-2.2 iteration, train.py:339-341; the RGB pass of a rescaled stage-2.1 call):
    `renderer.render(..., render_feat_map=False)` over a model shaped like scene/gaussian_model.py:GaussianModel with every
    parameter detached -- first through the full pipeline (cache off), then with rasterizer.KeptImages (the drop-in default)."""
    import types
    from opengaussian_amd import rasterizer as R
    from opengaussian_amd.renderer import render
    shs = scene.shs.detach()
    pc = types.SimpleNamespace(
        _xyz=scene.means3D.detach(), _scaling=torch.log(scene.scales.detach()), _rotation=scene.rotations.detach(),
        _opacity=torch.logit(scene.opacities.detach().clamp(1e-6, 1 - 1e-6)), _features_dc=shs[:, :1].contiguous(),
        _features_rest=shs[:, 1:].contiguous(), _ins_feat=scene.ins_feat.detach(), active_sh_degree=3, max_sh_degree=3)
    pc.get_xyz = pc._xyz
    pc.get_scaling, pc.get_rotation, pc.get_opacity, pc.get_features = scene.scales.detach(), pc._rotation, scene.opacities.detach(), shs
    pc.get_ins_feat = lambda origin=False: pc._ins_feat
    pipe = types.SimpleNamespace(debug=False, compute_cov3D_python=False, convert_SHs_python=False)
    bg = torch.zeros(3, device=device)
    cams = [c.to(device) for c in cams]

    def timed():
        for i in range(len(cams)):
            render(cams[i], pc, pipe, bg, 60000, rescale=False, render_feat_map=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(reps):
            render(cams[i % len(cams)], pc, pipe, bg, 60000, rescale=False, render_feat_map=False)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    saved = R.KEPT_IMAGES
    try:
        R.KEPT_IMAGES = R.KeptImages(budget_bytes=0)
        full = timed()
        R.KEPT_IMAGES = R.KeptImages(budget_bytes=4 << 30)
        kept = timed()
        st, nbytes = dict(R.KEPT_IMAGES.stats), R.KEPT_IMAGES.nbytes // max(len(R.KEPT_IMAGES.slots), 1)
    finally:
        R.KEPT_IMAGES = saved
    return {"ms_per_call_full_pass": full, "ms_per_call_kept_outputs": kept, "kept_bytes_per_view": nbytes, "cache": st,
            "what": "render(render_feat_map=False) of a fully detached model, 3-channel SH pass, forward only; kept: the "
                    "call returns clones of the camera's kept image / depth / alpha / radii, no pass is launched"}


def extra_workloads(args):
    """The other single-GPU configurations of BASELINE.json (C2 100 k / 800x800, C3-class 500 k / 988x731, C4-class 2 M /
    648x484), same fused 9-channel fwd+bwd step: each is a CHILD run of this script (fresh process, this process keeps its
    scene resident but launches nothing meanwhile); reported: ms/step, Mpix/s, the kernel sum of the per-launch breakdown
    (HIP events around every launch inflate sub-10-us kernels) and the HBM roofline of its dominant kernel."""
    import subprocess
    res = {}
    for wl, key in (("C2-100k-800", "C2"), ("C3-500k-988", "C3"), ("C4-2M-648", "C4"), ("C4-2M-648-inside", "C4_inside")):
        log(f"extra workload {wl}")
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--workload", wl, "--steps", "100", "--warmup", "10",
               "--views", str(args.views), "--no-cpu-baseline", "--no-kmeans", "--no-extra-workloads"]
        try:
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=600)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
            d = json.loads(line)
            res[key] = {"workload": d["config"]["workload"], "ms_per_step": d["ms_per_step"], "Mpix_per_s": d["value"],
                        "kernel_sum_ms": sum(d["kernels_ms_per_step"].values()),
                        "D_num_rendered": d["scene"]["D_num_rendered"], "mean_tile_list": d["scene"]["mean_tile_list"],
                        "roofline": d["roofline"], "step_algorithmic_GBps": d["step_algorithmic_GBps"],
                        "kernels_ms_per_step": d["kernels_ms_per_step"],
                        "stage1_ms_per_step": (d.get("stage1_pass") or {}).get("ms_per_step"),
                        "stage1_kept_pass_ms_per_step": ((d.get("stage1_pass") or {}).get("kept_pass") or {}).get("ms_per_step"),
                        "kept_bytes_per_view": ((d.get("stage1_pass") or {}).get("kept_pass") or {}).get("kept_bytes_per_view"),
                        "render_phase_sizing_timed": d["render_phase_sizing_timed"]}
        except Exception as e:          # an extra line never takes the headline down
            res[key] = {"error": repr(e)}
    return res


_T0 = time.time()


def log(msg):
    """progress on stderr (the JSON line is the only thing on stdout)"""
    sys.stderr.write(f"[bench +{time.time() - _T0:7.1f}s] {msg}\n")
    sys.stderr.flush()


def self_launch(args) -> int:
    """`python3 bench.py --gpus N` with N > 1 and no launcher around it (WORLD_SIZE unset): start the N ranks ourselves.

    The parent never touches the GPU (nothing below initialises HIP; `import torch` alone does not): it runs
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free>
    bench.py <same arguments>` as a CHILD process (no exec), relays rank 0's JSON line on stdout and returns the
    children's return code.  Under the driver's own torchrun command WORLD_SIZE is set and this is skipped."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on this pool (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    log("self-launch: " + " ".join(cmd))
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)       # stderr is inherited (progress lines)
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    js = [ln for ln in lines if ln.lstrip().startswith("{")]
    for ln in (js[-1:] if js else lines):
        print(ln)
    sys.stdout.flush()
    if r.returncode == 0 and not js:
        log("self-launch: the ranks exited 0 but printed no JSON line")
        return 1
    return r.returncode


def main():
    args = parse()
    log("start")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))
    from opengaussian_amd import _lib, dp
    from opengaussian_amd import rasterizer as R
    # backward on the calling thread: one process drives one GPU, so autograd's per-device worker thread only adds a thread hop per
    # backward() -- invisible at S1M (device-bound), but a 100 k-Gaussian step is paced by the host: 0.35-0.47 -> 0.295-0.30 ms on
    # one box (INTEGRATION.md recommends the same line to training scripts).  OGS_BENCH_AUTOGRAD_MT=1 restores torch's default.
    autograd_mt = os.environ.get("OGS_BENCH_AUTOGRAD_MT", "0") == "1"
    torch.autograd.set_multithreading_enabled(autograd_mt)
    from opengaussian_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer, rasterize_fused
    from opengaussian_amd.synthetic import make_scene, orbit_camera

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    rank, world, local = dp.init_from_env("cuda")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    device = torch.device("cuda", local % max(torch.cuda.device_count(), 1))
    torch.cuda.set_device(device)
    import torch.distributed as dist

    wl = WORKLOADS[args.workload]
    P, W, H, f = wl["P"], wl["W"], wl["H"], wl["f"]
    scene_cpu = make_scene(P, W, H, f, f, seed=0)
    if wl.get("behind"):
        gone = torch.rand(P, generator=torch.Generator().manual_seed(1234)) < wl["behind"]
        scene_cpu.means3D[gone, 2] = -scene_cpu.means3D[gone, 2].abs() - 1.0
    log(f"scene built: P={P} {W}x{H}")
    # step i of rank r renders view (i * world + r) mod V of a fan of V cameras about the scene centre (view 0 = the
    # identity camera of SURVEY.md section 8(d)): every rank a different view every step, as in multi-view training
    V = max(args.views, world, 1)
    cams_cpu = [orbit_camera(W, H, f, f, view_index=v, num_views=V) for v in range(V)]
    cam_cpu = cams_cpu[0]
    scene = scene_cpu.to(device)
    bg0 = torch.zeros(3, device=device)

    def settings_of(c):
        c = c.to(device)
        return GaussianRasterizationSettings(
            image_height=H, image_width=W, tanfovx=math.tan(c.FoVx * 0.5), tanfovy=math.tan(c.FoVy * 0.5), bg=bg0,
            scale_modifier=1.0, viewmatrix=c.world_view_transform, projmatrix=c.full_proj_transform, sh_degree=3,
            campos=c.camera_center, prefiltered=False, debug=False)
    all_settings = [settings_of(c) for c in cams_cpu]
    campos_views = torch.stack([c.camera_center for c in cams_cpu]).to(device)
    dist_on = world > 1 or dp.FORCE_COLLECTIVES
    info = {"i": 0, "D_seen": [], "xmode": args.exchange}

    def view_of(step_index, r):
        return (step_index * world + r) % V

    leaves = dict(means3D=scene.means3D, scales=scene.scales, rotations=scene.rotations, opacities=scene.opacities,
                  shs=scene.shs, ins_feat=scene.ins_feat)
    for v in leaves.values():
        v.requires_grad_(True)
    gen = torch.Generator().manual_seed(100 + rank)
    gC = torch.randn(3, H, W, generator=gen).to(device)
    gA = torch.randn(1, H, W, generator=gen).to(device)
    gF = torch.randn(6, H, W, generator=gen).to(device)

    names_a = ["means3D", "scales", "rotations", "opacities", "shs"]
    bucket_a = dp.GradBucket([leaves[n].shape for n in names_a], device) if dist_on else None
    bucket_b = dp.GradBucket([leaves["ins_feat"].shape], device) if dist_on and not args.rgb_only else None

    names_f = names_a + ["ins_feat"]
    fused = not (args.separate_passes or args.rgb_only)
    # N > 1 exchange per step (dp.py): ONE flat SUM bucket (per-Gaussian gradients + the two SUM-reducible
    # densification statistics), an all-gather of the [P,3] rank-1 factor of the SH gradient (rebuilt locally by
    # ogs_sh_grad_from_views: 4x fewer xGMI bytes than all-reducing [P,16,3]), and a MAX all-reduce of the radii.
    compress_sh = dist_on and fused and not args.dense_sh_allreduce
    names_x = [n for n in names_f if not (compress_sh and n == "shs")]
    nsets = 2                                     # double-buffered; 'sync' / 'none' use set 0 / nothing
    sets = []
    if dist_on and fused:
        for _ in range(nsets):
            sets.append(dict(bucket=dp.GradBucket([leaves[n].shape for n in names_x] + [(2, P)], device, average=False),
                             sh=dp.ShGradExchange(P, 16, device) if compress_sh else None,
                             dsh=torch.empty(P, 16, 3, device=device) if compress_sh else None,
                             pending=False, radii_work=None, campos=None))
    gCF = torch.cat([gC, gF])

    def finish(st):
        """make the reduced gradients of one exchange ready on the compute stream (what an optimizer would read)"""
        if not st["pending"]:
            return
        st["bucket"].wait()
        if st["sh"] is not None:
            st["sh"].rebuild(leaves["means3D"], st["campos"], 3, out=st["dsh"])
        if st["radii_work"] is not None:
            st["radii_work"].wait()
        st["pending"] = False

    def step_fused():
        """ONE rasterization pass: RGB (SH) in channels 0..2, ins_feat in 3..8; the feature loss is detached
        from geometry inside the backward kernel -> same gradients as the two passes of step()."""
        for v in leaves.values():
            v.grad = None
        m2 = torch.zeros(P, 3, device=device, requires_grad=True)
        sink = [] if compress_sh else None
        i = info["i"]
        info["i"] = i + 1
        color, radii, depth, alpha = rasterize_fused(leaves["means3D"], m2, leaves["opacities"], leaves["shs"],
                                                     leaves["ins_feat"], all_settings[view_of(i, rank)],
                                                     scales=leaves["scales"], rotations=leaves["rotations"],
                                                     sh_rgb_sink=sink)
        info["D"] = color.grad_fn.num_rendered
        info["D_seen"].append(info["D"])
        torch.autograd.backward([color, alpha], [gCF, gA])
        xmode = info["xmode"]
        if sets and xmode != "none":
            st = sets[i % nsets] if xmode == "pipelined" else sets[0]
            st["campos"] = campos_views[[view_of(i, r) for r in range(world)]]
            st["bucket"].pack([leaves[n].grad for n in names_x] + [dp.densification_stats(m2.grad, radii)])
            st["bucket"].allreduce_async()                               # RCCL, side stream
            if st["sh"] is not None:
                st["sh"].gather_async(sink[0])
            st["rmax"], st["radii_work"] = dp.reduce_max_radii(radii, async_op=True)   # the one non-SUM statistic
            st["pending"] = True
            # pipelined: this step's exchange keeps running while the NEXT render is enqueued; the previous
            # step's exchange (which overlapped this render) is completed now
            finish(sets[(i - 1) % nsets] if xmode == "pipelined" else st)
        return radii

    def drain():
        for st in sets:
            finish(st)

    def step():
        if fused:
            return step_fused()
        for v in leaves.values():
            v.grad = None
        i = info["i"]
        info["i"] = i + 1
        rast = GaussianRasterizer(all_settings[view_of(i, rank)])
        m2a = torch.zeros(P, 3, device=device, requires_grad=True)
        color, radii, depth, alpha = rast(means3D=leaves["means3D"], means2D=m2a, opacities=leaves["opacities"],
                                          shs=leaves["shs"], scales=leaves["scales"], rotations=leaves["rotations"])
        info["D"] = color.grad_fn.num_rendered
        info["D_seen"].append(info["D"])
        torch.autograd.backward([color, alpha], [gC, gA])
        if bucket_a is not None:
            bucket_a.pack([leaves[n].grad for n in names_a])
            bucket_a.allreduce_async()
        if not args.rgb_only:
            m2b = torch.zeros(P, 3, device=device, requires_grad=True)
            feat, _, _, _ = rast(means3D=leaves["means3D"].detach(), means2D=m2b, opacities=leaves["opacities"].detach(),
                                 colors_precomp=leaves["ins_feat"], scales=leaves["scales"].detach(),
                                 rotations=leaves["rotations"].detach())
            torch.autograd.backward([feat], [gF])
            if bucket_b is not None:
                bucket_b.pack([leaves["ins_feat"].grad])
                bucket_b.allreduce_async()
        if dist_on:
            dp.reduce_densification_stats(m2a.grad, radii)
            bucket_a.wait()
            if bucket_b is not None:
                bucket_b.wait()
        return radii

    # ---- the stage-1 training step (train.py:431-436: every parameter but ins_feat detached): fused 9-channel forward,
    # features-only backward; data-parallel exchange = ONE SUM all-reduce of dL/d ins_feat [P, 6], double-buffered like the
    # all-gradient step (the exchange of view i overlaps the render of view i + 1 and is completed right after it is enqueued)
    s1 = {"i": 0, "feat": None, "buckets": [], "mode": "pipelined"}

    def stage1_setup():
        if s1["feat"] is None:
            s1["feat"] = leaves["ins_feat"].detach().clone().requires_grad_(True)
            s1["det"] = {k: v.detach() for k, v in leaves.items()}
            s1["gC9"] = torch.cat([torch.zeros(3, H, W, device=device), gF])
            if dist_on:
                s1["buckets"] = [dp.GradBucket([leaves["ins_feat"].shape], device, average=False) for _ in range(2)]

    def step_stage1():
        stage1_setup()
        i = s1["i"]
        s1["i"] = i + 1
        det, feat = s1["det"], s1["feat"]
        m2 = torch.zeros(P, 3, device=device)                 # no grad: nothing consumes dL/dmeans2D after stage 0
        color, radii, depth, alpha = rasterize_fused(det["means3D"], m2, det["opacities"], det["shs"], feat,
                                                     all_settings[view_of(i, rank)], scales=det["scales"],
                                                     rotations=det["rotations"])
        info["D"] = color.grad_fn.num_rendered
        info["D_seen"].append(info["D"])
        feat.grad = None
        color.backward(s1["gC9"])
        if s1["buckets"] and s1["mode"] != "none":
            b = s1["buckets"][i % 2] if s1["mode"] == "pipelined" else s1["buckets"][0]
            b.pack([feat.grad])
            b.allreduce_async()
            (s1["buckets"][(i - 1) % 2] if s1["mode"] == "pipelined" else b).wait()
        return radii

    def drain_stage1():
        for b in s1["buckets"]:
            b.wait()

    def barrier():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    if args.stage1:
        if not fused:
            raise SystemExit("--stage1 times the fused 9-channel step (drop --separate-passes / --rgb-only)")
        s1["mode"] = args.exchange
        step_all, drain_all = step, drain              # keep the all-gradient step for the extras
        step, drain = step_stage1, drain_stage1

    # the warm-up steps bracket both blend kernels (mode 2, prefix "blend_") to learn which one dominates the step
    _lib.prof_filter("blend_")
    _lib.prof_enable(2)
    for i in range(args.warmup):
        radii = step()
        drain()
        torch.cuda.synchronize()
        log(f"warmup step {i} done (D={info.get('D')})")
    warm = _lib.prof_collect()
    dominant = max(warm, key=lambda k: warm[k]["total_ms"]).split("<")[0] if warm else "blend_"
    barrier()
    # Timed region.  HIP events on the launch stream bracket ONLY the dominant kernel here (the roofline's kernel:
    # its launch duration has to come from the timed region): an event pair serialises the queue for ~10 us per
    # launch, and a step issues ~35 launches, so bracketing every kernel inside the timed region would cost
    # ~0.35 ms/step; the other kernels are timed by the untimed pass below.
    _lib.prof_filter(dominant)
    _lib.prof_enable(2)
    stats0 = dict(R.PASS_STATS)
    info["D_seen"] = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        radii = step()
    drain()                                   # the last exchange(s) complete inside the timed region
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    log(f"timed region done: {elapsed / args.steps * 1e3:.3f} ms/step")
    prof = _lib.prof_collect()
    for v in prof.values():
        v["steps"] = args.steps
    stats_timed = dict(R.PASS_STATS)
    # second, untimed pass with every launch bracketed: the per-kernel breakdown
    _lib.prof_filter("blend_")
    if not args.no_extras:
        _lib.prof_enable(1)
        nb = min(args.steps, 40)
        for _ in range(nb):
            step()
        drain()
        torch.cuda.synchronize()
        prof_all = _lib.prof_collect()
        _lib.prof_enable(0)
        for name, v in prof_all.items():
            v["steps"] = nb
            prof.setdefault(name, v)
    else:
        _lib.prof_enable(0)
    if world > 1:
        te = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())

    # N > 1, untimed extras: the same steps with each step waiting for its own exchange ('sync') and with no exchange at
    # all ('none') -> how much of the exchange the pipelined mode leaves exposed (barrier + max over ranks, as above)
    xtimes = {}
    if dist_on and fused and not args.no_extras:
        nb = max(2, min(args.steps, 20))
        for mode in [m for m in ("pipelined", "sync", "none") if m != args.exchange]:
            info["xmode"] = mode
            s1["mode"] = mode
            for _ in range(2):
                step()
            drain()
            barrier()
            t1 = time.perf_counter()
            for _ in range(nb):
                step()
            drain()
            torch.cuda.synchronize()
            barrier()
            tm = torch.tensor([(time.perf_counter() - t1) / nb * 1e3], device=device, dtype=torch.float64)
            if world > 1:
                dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            xtimes[mode] = float(tm.item())
        info["xmode"] = args.exchange
        s1["mode"] = args.exchange
        log(f"exchange modes (ms/step): {xtimes}")

    def max_over_ranks(x):
        t = torch.tensor([x], device=device, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # N > 1, untimed extras: (a) every collective of the exchange ALONE -- issued on its side stream
    # exactly as the steps issue it, nothing else on the GPU, barrier + max over ranks -- so that whoever reads an 8-GPU
    # line can tell link time from compute time; (b) the other training step (stage 1 when the all-gradient step was
    # timed, and vice versa) through the same pipelined exchange: both scaling curves out of the driver's one command.
    coll, other_step = None, None
    if dist_on and fused and not args.no_extras:
        def time_collective(fn, reps=6):
            fn()
            barrier()
            t1 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            return max_over_ranks((time.perf_counter() - t1) / reps * 1e3)
        coll = {}
        ar_bus = 2.0 * (world - 1) / max(world, 1)              # ring all-reduce: bytes on each link per payload byte
        if sets:
            st = sets[0]
            finish(st)
            nbytes = st["bucket"].flat.numel() * 4
            st["bucket"].flat.zero_()
            ms = time_collective(lambda: (st["bucket"].allreduce_async(), st["bucket"].wait()))
            coll["grad_bucket_sum_allreduce"] = {"bytes": nbytes, "ms": ms, "busbw_GBps": nbytes * ar_bus / (ms * 1e-3) / 1e9}
            if st["sh"] is not None:
                fac = torch.zeros(P, 3, device=device)
                ms = time_collective(lambda: (st["sh"].gather_async(fac), st["sh"].wait()))
                coll["sh_factor_allgather"] = {"bytes_per_rank": P * 12, "ms": ms,
                                               "busbw_GBps": P * 12 * (world - 1) / (ms * 1e-3) / 1e9}
                ms = time_collective(lambda: st["sh"].rebuild(leaves["means3D"], campos_views[[view_of(0, r) for r in range(world)]],
                                                              3, out=st["dsh"]))
                coll["sh_rebuild_kernel"] = {"ms": ms}
        rad = torch.zeros(P, dtype=torch.int32, device=device)
        ms = time_collective(lambda: dp.reduce_max_radii(rad)[0])
        coll["radii_max_allreduce"] = {"bytes": P * 4, "ms": ms, "busbw_GBps": P * 4 * ar_bus / (ms * 1e-3) / 1e9}
        stage1_setup()
        if s1["buckets"]:
            b1 = s1["buckets"][0]
            b1.flat.zero_()
            ms = time_collective(lambda: (b1.allreduce_async(), b1.wait()))
            coll["stage1_ins_feat_sum_allreduce"] = {"bytes": P * 24, "ms": ms, "busbw_GBps": P * 24 * ar_bus / (ms * 1e-3) / 1e9}
        log(f"collectives alone (ms): { {k: round(v['ms'], 4) for k, v in coll.items()} }")
        # (b) the other step
        o_step, o_drain, o_name = ((step_all, drain_all, "all-gradient step (stage 0)") if args.stage1
                                   else (step_stage1, drain_stage1, "stage-1 step (features-only backward, ins_feat all-reduce)"))
        nb = max(4, min(args.steps, 40))
        for _ in range(3):
            o_step()
        o_drain()
        barrier()
        t1 = time.perf_counter()
        for _ in range(nb):
            o_step()
        o_drain()
        torch.cuda.synchronize()
        barrier()
        o_ms = max_over_ranks((time.perf_counter() - t1) / nb * 1e3)
        other_step = {"step": o_name, "ms_per_step": o_ms, "Mpix_per_s_all_ranks": world * W * H / (o_ms * 1e-3) / 1e6,
                      "exchange": "pipelined", "steps": nb,
                      "exchange_bytes_per_step_per_rank": (P * 24 if not args.stage1 else None)}
        log(f"other step: {other_step}")
    # who took part: every rank reports its device, gathered to rank 0 (lets a reader verify that N ranks on N
    # devices really ran under the backend named)
    dist_info = None
    if dist_on:
        prop = torch.cuda.get_device_properties(device)
        mine = {"rank": rank, "local_rank": local, "pid": os.getpid(), "cuda_device": torch.cuda.current_device(),
                "device_name": prop.name, "pci_bus_id": getattr(prop, "pci_bus_id", None),
                "uuid": str(getattr(prop, "uuid", "")) or None}
        everyone = [None] * world
        dist.all_gather_object(everyone, mine)
        xbytes = 0
        if args.stage1:
            xbytes = P * 24
        elif sets:
            xbytes = sets[0]["bucket"].flat.numel() * 4 + P * 4
            if sets[0]["sh"] is not None:
                xbytes += world * P * 3 * 4
        dist_info = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "ranks": everyone,
                     "devices_visible_per_rank": torch.cuda.device_count(),
                     "exchange_mode_timed": args.exchange,
                     "exchange_bytes_per_step_per_rank": xbytes,
                     "exchange_messages": (("SUM all-reduce of dL/d ins_feat, %d B" % (P * 24)) if args.stage1 else
                                           ("flat SUM all-reduce %d B + all-gather of the [P,3] SH-gradient factor %d B "
                                            "per rank + int32 MAX all-reduce %d B" %
                                            (sets[0]["bucket"].flat.numel() * 4,
                                             P * 12 if sets[0]["sh"] is not None else 0, P * 4)) if sets else None),
                     "collectives_alone": coll,
                     ("stage0_all_gradient_step" if args.stage1 else "stage1"): other_step,
                     "ms_per_step_by_exchange_mode": {args.exchange: elapsed / args.steps * 1e3, **xtimes},
                     "exposed_exchange_ms_per_step": (elapsed / args.steps * 1e3 - xtimes["none"]) if "none" in xtimes else None,
                     "sync_minus_pipelined_ms_per_step": ((xtimes.get("sync", elapsed / args.steps * 1e3) -
                                                          xtimes.get("pipelined", elapsed / args.steps * 1e3))
                                                         if xtimes else None)}

    ms_per_step = elapsed / args.steps * 1e3
    value = world * W * H * args.steps / elapsed / 1e6

    if rank == 0:
        D = int(info["D"])
        npx = W * H
        p_vis = int((radii > 0).sum().item())
        gxy = ((W + 15) // 16) * ((H + 15) // 16)
        per_kernel = {k: {"calls": v["calls"], "avg_ms": v["total_ms"] / max(v["calls"], 1), "total_ms": v["total_ms"],
                          "ms_per_step": v["total_ms"] / max(v["steps"], 1)} for k, v in prof.items()}
        known = set(algorithmic_bytes(P, D, npx, 9, 54))
        cands = [k for k in per_kernel if k.split("<")[0] in known] or list(per_kernel)
        dom = max(cands, key=lambda k: per_kernel[k]["ms_per_step"])       # the dominant kernel (among those SURVEY 8(d) prices)
        dom_base = dom.split("<")[0]
        dom_C = int(dom.split("<")[1].rstrip(">")) if "<" in dom and dom.split("<")[1].rstrip(">").isdigit() else 3
        ab = algorithmic_bytes(P, D, npx, dom_C, {3: 48, 9: 54}.get(dom_C, dom_C))
        dom_bytes = ab.get(dom_base)
        # Counter-derived figures are OFFLINE: collected by separate `rocprofv3 --pmc` passes of this same command
        # (scripts/collect_pmc.py, scripts/collect_sq.py -- a timed run cannot carry counters) and committed under
        # profiles/ together with the kernel version they were taken on; they are attached only when that version
        # is the library's (ogs_version()) and the workload is the one profiled.
        traffic, traffic_src, hbm_kernels, valu = None, None, None, None
        pmc_file = os.path.join(ROOT, "profiles", "pmc_traffic.json" if args.workload == "S1M-1080p"
                                else f"pmc_traffic_{args.workload}.json")
        sq_file = os.path.join(ROOT, "profiles", "sq_valu.json")
        libver = int(_lib.lib().ogs_version())
        profiled = fused and args.views == 8 and not args.stage1
        if os.path.exists(pmc_file) and profiled:
            try:
                pmc = json.load(open(pmc_file))
                if int(pmc.get("_ogs_version", -1)) == libver:
                    traffic = pmc.get(dom)
                    traffic_src = pmc.get("_source")
                    # measured HBM bytes per launch (PMC) / live launch duration, for every kernel of the step
                    hbm_kernels = {k: {"GBps": pmc[k] / (v["avg_ms"] * 1e-3) / 1e9,
                                       "frac_of_peak": pmc[k] / (v["avg_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS}
                                   for k, v in per_kernel.items() if k in pmc and v["avg_ms"] > 0}
            except Exception:
                traffic, hbm_kernels = None, None
        if os.path.exists(sq_file) and profiled and args.workload == "S1M-1080p":
            try:
                sq = json.load(open(sq_file))
                if int(sq.get("_ogs_version", -1)) == libver and dom in sq:
                    n_valu = float(sq[dom]["SQ_INSTS_VALU"])          # VALU wave-instructions per launch
                    rate = n_valu / (per_kernel[dom]["avg_ms"] * 1e-3) / 1e9
                    valu = {"bound": "valu", "kernel": dom, "achieved": rate, "peak": VALU_PEAK_GINST, "unit": "G wave-instr/s",
                            "frac": rate / VALU_PEAK_GINST, "valu_wave_instructions_per_launch": n_valu,
                            "peak_note": "2 cycles per wave64 instruction per SIMD (all-VGPR forms, measured); forms with an SGPR "
                                         "operand, DPP, compares / selects, min / max issue at 4",
                            "source": sq.get("_source")}
                    mix_file = os.path.join(ROOT, "profiles", "r04_issue_mix.json")
                    if os.path.exists(mix_file):
                        mix = json.load(open(mix_file)).get(dom.split("<")[0])
                        if mix:
                            cyc = float(mix["modelled_issue_cycles_per_valu"])
                            mix_peak = 256 * 4 * 2.4 / cyc
                            valu["issue_model"] = {"modelled_cycles_per_valu_instruction": cyc,
                                                   "share_of_half_rate_or_slower_forms": mix["share_half_rate_or_slower"],
                                                   "peak_for_this_mix": mix_peak, "frac_of_mix_peak": rate / mix_peak,
                                                   "source": "scripts/isa_issue_mix.py (static mix of the kernel's vector "
                                                             "instructions x the measured price list)"}
            except Exception:
                valu = None
        roofline = None
        if dom_bytes is not None:
            achieved = dom_bytes / (per_kernel[dom]["avg_ms"] * 1e-3) / 1e9
            # `bound` names the ruler of this object (HBM bytes, as the bench contract asks); the kernel's actual
            # limiter is VALU issue -- see `limiter` and the sibling `roofline_valu`
            roofline = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                        "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_src,
                        "limiter": {"blend_backward_kernel": "the L2's fp64 atomic path (sensitivity probes, DESIGN.md section 3b: every atomic "
                                                             "issued twice 0.62 -> 1.12 ms, none 0.47 ms; +8 VALU per entry or one "
                                                             "workgroup less per CU: no change) -- not HBM bytes, not vector issue",
                                    "pack_blend_chunked_kernel": "vector issue and occupancy (probes, DESIGN.md section 3b) -- not HBM bytes",
                                    "blend_backward_feat_lds_kernel": "vector issue of the walk (its atomics cost 0.02 of 0.36 ms)",
                                    "preprocess_backward_kernel": "HBM at the part's mixed read / write rate (4.6 TB/s for a 1 : 1 copy)",
                                    "preprocess_kernel": "HBM at the part's mixed read / write rate"}.get(dom_base, "see DESIGN.md section 4"),
                        "algorithmic_bytes_per_launch": dom_bytes, "avg_launch_ms": per_kernel[dom]["avg_ms"]}
        step_bytes = sum(algorithmic_bytes(P, D, npx, 3, 48)[k] for k in ("fwd", "bwd"))
        if fused:
            step_bytes = sum(algorithmic_bytes(P, D, npx, 9, 54)[k] for k in ("fwd", "bwd"))
        elif not args.rgb_only:
            step_bytes += sum(algorithmic_bytes(P, D, npx, 6, 6)[k] for k in ("fwd", "bwd"))
        out = {
            "metric": "fwd+bwd Mpix/s at 1080p, 1M Gaussians (RGB+6-D ins_feat); k-means it/s",
            "value": value, "unit": "Mpix/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.workload + (" STAGE-1 step: RGB(SH3)+6-ch ins_feat fused 9-channel forward, features-only backward "
                                                    "(everything but ins_feat detached, train.py:431-436)" if args.stage1 else
                                                    " RGB(SH3) fwd+bwd only" if args.rgb_only else
                                                    (" RGB(SH3)+depth+alpha (all grads) + 6-ch ins_feat (grad to ins_feat) fwd+bwd, " +
                                                     ("ONE fused 9-channel pass" if fused else "two passes (3ch SH + 6ch)"))),
                       "gaussians": P, "width": W, "height": H, "views_per_step": world, "views_cycled": V,
                       "parallelism": (f"view-dp{world}: one view per GPU, Gaussians replicated; per step one SUM "
                                       f"all-reduce of the per-Gaussian gradients, "
                                       + ("all-gather of the rank-1 SH-gradient factor, " if compress_sh else "")
                                       + f"MAX all-reduce of radii; exchange {args.exchange}"
                                       + (" (overlaps the next view's render; all exchanges complete in the timed region)"
                                          if args.exchange == "pipelined" else "")
                                       + " over RCCL") if dist_on else "single"},
            "scene": {"P_visible": p_vis, "D_num_rendered": D, "mean_tile_list": D / gxy, "views_cycled": V,
                      "D_min_max_over_timed_steps": [int(min(info["D_seen"][:args.steps] or [D])),
                                                     int(max(info["D_seen"][:args.steps] or [D]))]},
            # how the render phase was sized during the timed steps: sync-free (capacity hint) vs blocking read-back
            # vs overflow (render phase enqueued twice); the timed value pays for every one of them
            "render_phase_sizing_timed": {k: stats_timed[k] - stats0[k] for k in stats_timed},
            "host": {"autograd_multithreading": autograd_mt},
            "dist": dist_info,
            "roofline": roofline,
            "roofline_valu": valu,
            "pmc_hbm_rate_per_kernel": hbm_kernels,
            "step_algorithmic_GBps": step_bytes / (ms_per_step * 1e-3) / 1e9,
            "kernels_ms_per_step": {k: v["ms_per_step"] for k, v in sorted(per_kernel.items(), key=lambda kv: -kv[1]["ms_per_step"])},
        }
        if world == 1 and not args.no_kmeans:
            try:
                log("k-means bench")
                out["kmeans"] = kmeans_bench(device)
                try:
                    out["kmeans"]["leaf"] = kmeans_leaf_bench(device)
                except Exception as e:
                    out["kmeans"]["leaf"] = {"error": repr(e)}
            except Exception as e:   # k-means is the second half of the metric, never the headline value
                out["kmeans"] = {"error": repr(e)}
        if world == 1 and not args.no_extras:
            try:
                log("stage-1 pass (features-only backward)")
                out["stage1_pass"] = stage1_bench(leaves, all_settings, gF, P, device)
            except Exception as e:
                out["stage1_pass"] = {"error": repr(e)}
            try:
                out["frozen_rgb_call"] = frozen_rgb_bench(scene, cams_cpu, W, H, device)
            except Exception as e:
                out["frozen_rgb_call"] = {"error": repr(e)}
        if world == 1 and not args.no_cpu_baseline:
            log("cpu baseline (oracle)")
            out["cpu_baseline"] = cpu_baseline(scene_cpu, cam_cpu, W, H, f, args.cpu_tile_stride, args.rgb_only)
            if "kmeans" in out and "error" not in out["kmeans"]:
                out["kmeans"]["cpu_baseline"] = kmeans_cpu_baseline()
        if world == 1 and args.workload == "S1M-1080p" and not args.no_extras and not args.no_extra_workloads:
            out["extras"] = extra_workloads(args)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
