"""GPU parity tests: HIP rasterizer (through the drop-in GaussianRasterizer -> C ABI) vs the CPU oracle on
identical seeded inputs.  Tolerances (BASELINE.json north_star): images / depth / alpha within 1e-4 fp32;
radii, tile ranges, sorted (tile<<32 | depth_bits) keys and point lists BIT-EXACT; n_contrib exact except
for pixels where the device exp and the host exp disagree on a threshold (bounded fraction, documented)."""
import os

import numpy as np
import pytest
import torch

from tests import helpers

pytestmark = pytest.mark.gpu

IMG_TOL = 1e-4


def _fwd_case(P, W, H, f, seed, use_sh, use_cov, device, lsm=-3.0, ties=False, bg=(0.1, 0.2, 0.3)):
    from oracle import raster_oracle as ro
    sc, cam = helpers.tiny_scene(P, W, H, f, seed=seed, log_scale_mean=lsm, with_ties=ties)
    inp = helpers.oracle_inputs(sc, cam, use_sh=use_sh, use_cov=use_cov)
    ref = ro.render_forward(W=W, H=H, tanfovx=W / (2 * f), tanfovy=H / (2 * f), bg=np.array(bg, np.float32),
                            sh_degree=3, **inp)
    (color, radii, depth, alpha), leaves = helpers.hip_forward(inp, cam, bg, 3, device, requires_grad=True)
    keys, ranges, ncontrib, plist = helpers.hip_export_binning(color)
    return ref, color, radii, depth, alpha, keys, ranges, ncontrib, plist


CASES = [
    # P, W, H, f, seed, use_sh, use_cov
    (2000, 160, 96, 120.0, 0, True, False),
    (1500, 100, 77, 90.0, 1, False, False),      # sizes not multiples of 16, colours precomputed
    (1200, 64, 48, 60.0, 2, True, True),         # cov3D_precomp path
    (5000, 320, 200, 250.0, 3, True, False),
    # P <= 1024: the geometry phase is ONE workgroup (small_geometry_kernel: rank sort + block scan in LDS); depth ties
    (700, 200, 120, 150.0, 3, True, False),
    (1024, 96, 64, 80.0, 4, False, False),
    (1025, 96, 64, 80.0, 4, False, False),       # first size on the radix path again
    (257, 130, 70, 100.0, 6, True, True),
]


@pytest.mark.parametrize("P,W,H,f,seed,use_sh,use_cov", CASES)
def test_forward_parity(gpu_device, P, W, H, f, seed, use_sh, use_cov):
    ref, color, radii, depth, alpha, keys, ranges, ncontrib, plist = _fwd_case(
        P, W, H, f, seed, use_sh, use_cov, gpu_device, ties=(seed == 3))
    g, b = ref["geom"], ref["binning"]
    # integers: bit exact
    np.testing.assert_array_equal(radii.cpu().numpy(), g.radii)
    assert len(keys) == b.num_rendered
    np.testing.assert_array_equal(keys, b.keys_sorted)
    np.testing.assert_array_equal(plist, b.point_list)
    # duplicate's per-pair tile test: whatever it drops cannot contribute (float64, every pixel of the tile), and it does
    # drop a substantial share (the blend kernels' lane occupancy depends on it)
    flags = helpers.LAST_REACH_FLAGS[0]
    n_dropped = helpers.assert_reach_flags_keep_every_contributor(flags, keys, plist, g, W, H)
    assert n_dropped > 0.15 * len(plist), (n_dropped, len(plist))
    np.testing.assert_array_equal(ranges, b.ranges)
    # floats: 1e-4
    np.testing.assert_allclose(color.detach().cpu().numpy(), ref["color"], atol=IMG_TOL, rtol=0)
    np.testing.assert_allclose(alpha.detach().cpu().numpy(), ref["alpha"], atol=IMG_TOL, rtol=0)
    # depth is an un-normalised sum of z*w with z up to 10: scale the absolute tolerance by max depth
    np.testing.assert_allclose(depth.detach().cpu().numpy(), ref["depth"], atol=IMG_TOL * 10, rtol=0)
    mism = (ncontrib != ref["n_contrib"].astype(np.uint32)).mean()
    assert mism < 2e-3, f"n_contrib mismatch fraction {mism}"


def test_wave_fold16(gpu_device):
    from opengaussian_amd import _lib
    g = torch.Generator().manual_seed(0)
    x = torch.randn(64, 16, generator=g)
    # asymmetric integer pattern as well, to catch permutation errors exactly
    xi = (torch.arange(64)[:, None] * 17 + torch.arange(16)[None, :] * 3 + 1).float()
    for inp in (x, xi):
        d = inp.to(gpu_device).contiguous()
        out = torch.zeros(64, device=gpu_device)
        _lib.check(_lib.lib().ogs_selftest_wave_fold16(d.data_ptr(), out.data_ptr(), 0), "selftest")
        torch.cuda.synchronize()
        expect = inp.double().sum(0)[torch.arange(64) // 4].float()
        torch.testing.assert_close(out.cpu(), expect, rtol=1e-5, atol=1e-4)


def test_tile_order_is_a_heaviest_first_permutation(gpu_device):
    """blend_fwd.hip::tile_order_kernel: any permutation is a valid schedule; the point is that long lists start first
    (non-increasing pseudo-log length class) and that balanced lists keep the raster order."""
    from opengaussian_amd import _lib

    def run(lengths):
        n = len(lengths)
        start = torch.cumsum(torch.cat([torch.zeros(1, dtype=torch.int64), lengths[:-1]]), 0)
        ranges = torch.stack([start, start + lengths], 1).to(torch.int32).to(gpu_device).contiguous()
        order = torch.full((n,), -1, dtype=torch.int32, device=gpu_device)
        _lib.check(_lib.lib().ogs_selftest_tile_order(ranges.data_ptr(), n, order.data_ptr(), 0), "selftest")
        torch.cuda.synchronize()
        return order.cpu().long()

    def work_class(n):                       # the kernel's scale: exact below 8, then exponent * 8 + 3 mantissa bits
        n = int(n)
        if n < 8:
            return n
        e = n.bit_length() - 1
        return min(e * 8 + ((n >> (e - 3)) & 7), 255)

    g = torch.Generator().manual_seed(3)
    skewed = (torch.rand(8160, generator=g) ** 6 * 20000).long()            # two orders of magnitude, many empty tiles
    o = run(skewed)
    assert torch.equal(torch.sort(o).values, torch.arange(8160))            # a permutation
    cls = torch.tensor([work_class(skewed[t]) for t in o.tolist()])
    assert (cls[1:] <= cls[:-1]).all()                                       # heaviest class first
    balanced = 900 + (torch.rand(8160, generator=g) * 200).long()          # longest <= 2 x mean: raster order
    assert torch.equal(run(balanced), torch.arange(8160))
    assert torch.equal(torch.sort(run(torch.tensor([5, 0, 700, 3]))).values, torch.arange(4))


@pytest.mark.parametrize("mode", ["feat3", "feat6", "fused9", "grouped6"])
def test_features_only_backward(gpu_device, mode):
    """Stage >= 1 training graph (train.py:431-436): every Gaussian parameter but ins_feat is detached and nothing
    consumes dL/dmeans2D -> the pass runs its features-only backward kernels.  dL/d ins_feat must equal (a) the
    float64 autograd oracle and (b) what the full backward produces for the same inputs."""
    from oracle import raster_oracle as ro
    from opengaussian_amd.rasterizer import GaussianRasterizer, rasterize_fused, rasterize_groups
    P, W, H, f = 2200, 150, 100, 110.0
    sc, cam = helpers.tiny_scene(P, W, H, f, seed=77)
    dev = gpu_device
    Cf = 3 if mode == "feat3" else 6
    feat_cpu = sc.ins_feat[:, :Cf].contiguous()
    bgc = (0.3, 0.1, 0.2)
    rs = helpers.settings_for(cam, bgc, 3, dev)
    rng = np.random.default_rng(5)
    d = lambda t: t.to(dev)

    def run(full):
        feat = d(feat_cpu).clone().requires_grad_(True)
        geo = {k: d(getattr(sc, k)).clone().requires_grad_(full) for k in ("means3D", "opacities", "scales", "rotations")}
        m2 = torch.zeros(P, 3, device=dev, requires_grad=full)
        if mode == "fused9":
            shs = d(sc.shs).clone().requires_grad_(full)
            color, _, _, _ = rasterize_fused(geo["means3D"], m2, geo["opacities"], shs, feat, rs, scales=geo["scales"],
                                             rotations=geo["rotations"], detach_extra_from_geometry=False)
        elif mode == "grouped6":
            ids = (torch.arange(P, device=dev) % 4) - 1          # -1: in no group; 3 groups
            color, _, _, _ = rasterize_groups(geo["means3D"], m2, geo["opacities"], ids, 3, rs, colors_precomp=feat,
                                              scales=geo["scales"], rotations=geo["rotations"])
        else:
            color, _, _, _ = GaussianRasterizer(rs)(means3D=geo["means3D"], means2D=m2, opacities=geo["opacities"],
                                                    colors_precomp=feat, scales=geo["scales"], rotations=geo["rotations"])
        gC = torch.tensor(np.random.default_rng(5).standard_normal(tuple(color.shape)), dtype=torch.float32, device=dev)
        if mode == "fused9" and not full:
            gC[:3] = 0                                           # (RGB loss has no path to ins_feat either way)
        color.backward(gC)
        return feat.grad, gC, m2.grad

    g_feat, gC, m2g = run(full=False)
    assert m2g is None
    g_full, _, m2g_full = run(full=True)
    assert m2g_full is not None
    scale = float(g_full.abs().max())
    assert float((g_feat - g_full).abs().max()) / scale < 2e-5, mode
    if mode in ("feat3", "feat6"):
        inp = helpers.oracle_inputs(sc, cam, feat=feat_cpu)
        bg = np.array((bgc * 2)[:Cf], np.float32)
        ref = ro.render_forward(W=W, H=H, tanfovx=W / (2 * f), tanfovy=H / (2 * f), bg=bg, sh_degree=3, **inp)
        gref = ro.render_backward_f64(inp, ref["binning"], W, H, W / (2 * f), H / (2 * f), bg.astype(np.float64),
                                      gC.cpu().double().numpy(), np.zeros((1, H, W)), np.zeros((1, H, W)), sh_degree=3)
        want = gref["colors_precomp"].reshape(P, Cf)
        assert np.abs(g_feat.cpu().double().numpy() - want).max() / np.abs(want).max() < 1e-4


def test_sh_path_equals_precomputed_colors(gpu_device):
    """gaussian_renderer/__init__.py:92-97: rasterizer SH path == colors_precomp = clamp_min(eval_sh + 0.5, 0)."""
    from oracle import raster_oracle as ro
    sc, cam = helpers.tiny_scene(3000, 200, 120, 150.0, seed=5)
    inp_sh = helpers.oracle_inputs(sc, cam, use_sh=True)
    rgb, _ = ro.eval_sh_rgb(3, sc.shs.numpy(), sc.means3D.numpy(), cam.camera_center.numpy())
    inp_pc = helpers.oracle_inputs(sc, cam, feat=torch.from_numpy(rgb))
    (c1, r1, d1, a1), _ = helpers.hip_forward(inp_sh, cam, (0, 0, 0), 3, gpu_device)
    (c2, r2, d2, a2), _ = helpers.hip_forward(inp_pc, cam, (0, 0, 0), 3, gpu_device)
    assert torch.equal(r1, r2)
    torch.testing.assert_close(c1, c2, atol=2e-6, rtol=0)
    torch.testing.assert_close(a1, a2, atol=0, rtol=0)


def test_scale_rot_path_equals_cov3d_precomp(gpu_device):
    """gaussian_renderer/__init__.py:81-85 + scene/gaussian_model.py:41-45."""
    sc, cam = helpers.tiny_scene(3000, 200, 120, 150.0, seed=6)
    (c1, r1, d1, a1), _ = helpers.hip_forward(helpers.oracle_inputs(sc, cam, use_cov=False), cam, (0, 0, 0), 3, gpu_device)
    (c2, r2, d2, a2), _ = helpers.hip_forward(helpers.oracle_inputs(sc, cam, use_cov=True), cam, (0, 0, 0), 3, gpu_device)
    assert torch.equal(r1, r2)
    torch.testing.assert_close(c1, c2, atol=0, rtol=0)


def _grad_check(inp, cam, W, H, f, device, sh_degree=3, bg=(0.1, 0.2, 0.3), seed=0):
    from oracle import raster_oracle as ro
    ref = ro.render_forward(W=W, H=H, tanfovx=W / (2 * f), tanfovy=H / (2 * f), bg=np.array(bg, np.float32),
                            sh_degree=sh_degree, **inp)
    Cn = ref["color"].shape[0]
    rng = np.random.default_rng(seed)
    gC, gD, gA = rng.standard_normal((Cn, H, W)), rng.standard_normal((1, H, W)), rng.standard_normal((1, H, W))
    gref = ro.render_backward_f64(inp, ref["binning"], W, H, W / (2 * f), H / (2 * f), np.array(bg, np.float64),
                                  gC, gD, gA, sh_degree=sh_degree)
    (color, radii, depth, alpha), leaves = helpers.hip_forward(inp, cam, bg, sh_degree, device, requires_grad=True)
    t = lambda a: torch.tensor(a, dtype=torch.float32, device=device)
    loss = (color * t(gC)).sum() + (depth * t(gD)).sum() + (alpha * t(gA)).sum()
    loss.backward()
    g32 = helpers.lazy(lambda: ro.render_backward_f64(inp, ref["binning"], W, H, W / (2 * f), H / (2 * f), np.array(bg, np.float64),
                                                      gC, gD, gA, sh_degree=sh_degree, dtype=torch.float32))
    out = {}
    for k, v in leaves.items():
        if v is None or gref.get(k) is None:
            continue
        got = v.grad.detach().cpu().double().numpy()
        want = gref[k].reshape(got.shape)
        out[k] = helpers.assert_grads_close_modulo_threshold_flips(got, want, GRAD_TOL, want_fp32=lambda k=k: g32()[k], what=k)
    return out


# max |hip - f64 autograd| / max |f64 autograd| per gradient family.  Measured 1e-6 .. 1e-5 on these scenes since
# the backward starts from the forward's exact final transmittance and accumulates the per-Gaussian record in
# fp64 (round 1: ~1e-4 .. 1e-3 with T_final = 1 - alpha and fp32 atomics); what is left is the fp32 T recovery by
# division and the 1/(det^2 + 1e-7) chain of A.5.
GRAD_TOL = 2e-4


@pytest.mark.parametrize("use_sh,use_cov,seed", [(True, False, 0), (False, False, 1), (True, True, 2)])
def test_backward_parity(gpu_device, use_sh, use_cov, seed):
    W, H, f = 128, 80, 100.0
    sc, cam = helpers.tiny_scene(1500, W, H, f, seed=seed)
    inp = helpers.oracle_inputs(sc, cam, use_sh=use_sh, use_cov=use_cov)
    errs = _grad_check(inp, cam, W, H, f, gpu_device, seed=seed)
    print("backward_parity", use_sh, use_cov, {k: f"{e:.1e}" for k, e in errs.items()})
    assert errs, "no gradients compared"
    for k, e in errs.items():
        assert e < GRAD_TOL, f"{k}: relative error {e} (all: {errs})"


@pytest.mark.parametrize("P,keep", [(6000, 0.2), (1500, 0.05), (3000, 0.0)])
def test_view_that_culls_most_of_the_model(gpu_device, P, keep):
    """A camera inside a room-scale scene sees a fraction of the model.  The culled Gaussians (behind the near plane, empty
    tile rect) carry the drop key and leave the depth sort in its FIRST pass; the later passes, the scan of the tile counts
    and duplicate run on the visible count (a device word), the launches stay sized for P.  Keys, point list, ranges, radii:
    bit-exact against the oracle; images and every gradient at the usual bars -- with 80 %, 95 % and ALL of the model culled."""
    from oracle import raster_oracle as ro
    W, H, f = 144, 96, 110.0
    sc, cam = helpers.tiny_scene(P, W, H, f, seed=17)
    g = torch.Generator().manual_seed(3)
    gone = torch.rand(P, generator=g) >= keep
    # half of the culled ones behind the camera, the other half far off to the side (in front, empty tile rect)
    behind = gone & (torch.rand(P, generator=g) < 0.5)
    aside = gone & ~behind
    sc.means3D[behind, 2] = -sc.means3D[behind, 2].abs() - 1.0
    sc.means3D[aside, 0] = sc.means3D[aside, 0] + 500.0
    inp = helpers.oracle_inputs(sc, cam)
    bg = (0.1, 0.2, 0.3)
    ref = ro.render_forward(W=W, H=H, tanfovx=W / (2 * f), tanfovy=H / (2 * f), bg=np.array(bg, np.float32), sh_degree=3, **inp)
    (color, radii, depth, alpha), leaves = helpers.hip_forward(inp, cam, bg, 3, gpu_device, requires_grad=True)
    gm, b = ref["geom"], ref["binning"]
    np.testing.assert_array_equal(radii.cpu().numpy(), gm.radii)
    visible = int((gm.radii > 0).sum())
    assert visible <= max(1, int(1.3 * keep * P)) and (keep == 0.0) == (visible == 0)
    if visible:
        keys, ranges, ncontrib, plist = helpers.hip_export_binning(color)
        assert len(keys) == b.num_rendered
        np.testing.assert_array_equal(keys, b.keys_sorted)
        np.testing.assert_array_equal(plist, b.point_list)
        np.testing.assert_array_equal(ranges, b.ranges)
    np.testing.assert_allclose(color.detach().cpu().numpy(), ref["color"], atol=IMG_TOL, rtol=0)
    np.testing.assert_allclose(alpha.detach().cpu().numpy(), ref["alpha"], atol=IMG_TOL, rtol=0)
    if visible:
        errs = _grad_check(inp, cam, W, H, f, gpu_device, seed=4)
        assert errs
        for k, e in errs.items():
            assert e < GRAD_TOL, f"{k}: relative error {e} (all: {errs})"


def test_backward_parity_6ch(gpu_device):
    """fused 6-channel ins_feat pass (colors_precomp [P,6]) forward + backward."""
    W, H, f = 96, 64, 80.0
    sc, cam = helpers.tiny_scene(1000, W, H, f, seed=7)
    inp = helpers.oracle_inputs(sc, cam, feat=sc.ins_feat)
    errs = _grad_check(inp, cam, W, H, f, gpu_device, bg=(0,) * 6, seed=7)
    for k, e in errs.items():
        assert e < GRAD_TOL, f"{k}: relative error {e} (all: {errs})"


def test_dropping_unreachable_pairs_changes_no_image_and_no_gradient(gpu_device):
    """Default mode: the (Gaussian, tile) pairs that cannot reach a pixel of their tile leave the list before the tile sort
    (OgsRasterFwdArgs.full_binning = 0).  Against the same pass with the reference's full list (`rasterizer.full_binning()`):
    every image and every gradient BIT for bit -- the dropped pairs contributed nothing, and the surviving ones are blended and
    folded in the same order.  (The list itself is checked entry by entry in helpers.hip_export_binning.)"""
    from opengaussian_amd import rasterizer as R
    from opengaussian_amd.rasterizer import rasterize_fused
    W, H, f, P = 176, 112, 130.0, 3000
    sc, cam = helpers.tiny_scene(P, W, H, f, seed=21)
    dev = gpu_device
    rs = helpers.settings_for(cam, (0.1, 0.0, 0.2), 3, dev)
    g = torch.Generator().manual_seed(6)
    gCF, gA, gD = (torch.randn(9, H, W, generator=g).to(dev), torch.randn(1, H, W, generator=g).to(dev),
                   torch.randn(1, H, W, generator=g).to(dev))

    def run():
        lv = {k: getattr(sc, k).clone().to(dev).requires_grad_(True)
              for k in ("means3D", "scales", "rotations", "opacities", "shs", "ins_feat")}
        m2 = torch.zeros(P, 3, device=dev, requires_grad=True)
        c, r, d, a = rasterize_fused(lv["means3D"], m2, lv["opacities"], lv["shs"], lv["ins_feat"], rs, scales=lv["scales"],
                                     rotations=lv["rotations"])
        torch.autograd.backward([c, a, d], [gCF, gA, gD])
        return (c, r, d, a), {**{k: v.grad for k, v in lv.items()}, "means2D": m2.grad}, c.grad_fn

    assert R.FULL_BINNING is False
    out_c, grad_c, fn_c = run()
    with R.full_binning():
        out_f, grad_f, fn_f = run()
    assert fn_c.full_binning is False and fn_f.full_binning is True and fn_c.num_rendered == fn_f.num_rendered
    for x, y in zip(out_c, out_f):
        assert torch.equal(x, y)
    for k in grad_c:
        assert grad_c[k] is not None and torch.equal(grad_c[k], grad_f[k]), k


def test_fused_pass_equals_separate_passes(gpu_device):
    """rasterize_fused (RGB via SH + 6-D ins_feat in ONE pass, feature loss detached from geometry) must give
    what the reference's separate passes give: pass A = RGB with all gradients, pass B = 6 feature channels
    with everything but the features detached (gaussian_renderer/__init__.py:104-151, train.py:431-436)."""
    from opengaussian_amd.rasterizer import GaussianRasterizer, rasterize_fused
    W, H, f = 160, 96, 120.0
    sc, cam = helpers.tiny_scene(2500, W, H, f, seed=12)
    dev = gpu_device
    rs = helpers.settings_for(cam, (0.0, 0.0, 0.0), 3, dev)
    g = torch.Generator().manual_seed(5)
    gC, gA, gF = (torch.randn(3, H, W, generator=g).to(dev), torch.randn(1, H, W, generator=g).to(dev),
                  torch.randn(6, H, W, generator=g).to(dev))

    def leaves():
        return {k: getattr(sc, k).clone().to(dev).requires_grad_(True)
                for k in ("means3D", "scales", "rotations", "opacities", "shs", "ins_feat")}

    # separate passes
    a = leaves()
    m2 = torch.zeros(2500, 3, device=dev, requires_grad=True)
    rast = GaussianRasterizer(rs)
    cA, rA, dA, aA = rast(means3D=a["means3D"], means2D=m2, opacities=a["opacities"], shs=a["shs"], scales=a["scales"],
                          rotations=a["rotations"])
    m2b = torch.zeros(2500, 3, device=dev, requires_grad=True)
    cB, _, _, _ = rast(means3D=a["means3D"].detach(), means2D=m2b, opacities=a["opacities"].detach(),
                       colors_precomp=a["ins_feat"], scales=a["scales"].detach(), rotations=a["rotations"].detach())
    torch.autograd.backward([cA, aA, cB], [gC, gA, gF])
    # fused pass
    b = leaves()
    m2f = torch.zeros(2500, 3, device=dev, requires_grad=True)
    cF, rF, dF, aF = rasterize_fused(b["means3D"], m2f, b["opacities"], b["shs"], b["ins_feat"], rs, scales=b["scales"],
                                     rotations=b["rotations"])
    torch.autograd.backward([cF, aF], [torch.cat([gC, gF]), gA])
    assert torch.equal(rA, rF)
    torch.testing.assert_close(cF[:3], cA, atol=1e-6, rtol=0)
    torch.testing.assert_close(cF[3:], cB, atol=1e-6, rtol=0)
    torch.testing.assert_close(aF, aA, atol=1e-6, rtol=0)
    for k in ("means3D", "scales", "rotations", "opacities", "shs", "ins_feat"):
        ga, gb = a[k].grad, b[k].grad
        scale = float(ga.abs().max()) + 1e-12
        assert float((ga - gb).abs().max()) / scale < 1e-4, k      # float atomics reorder sums
    scale = float(m2.grad.abs().max())
    assert float((m2.grad - m2f.grad).abs().max()) / scale < 1e-4


def test_deferred_render_phase_and_capacity_overflow(gpu_device):
    """Second and later passes at one size skip the blocking num_rendered read-back (capacity from the previous
    pass).  Results must not depend on the path taken: first pass (blocking), steady state (deferred), and a
    forced capacity overflow (deferred result discarded, render phase redone with exact buffers)."""
    from opengaussian_amd import rasterizer as R
    W, H, f = 160, 96, 120.0
    sc, cam = helpers.tiny_scene(2500, W, H, f, seed=31)
    inp = helpers.oracle_inputs(sc, cam, use_sh=True)
    key = (2500, W, H, 1)
    R._LAST_NUM_RENDERED.pop(key, None)
    stats0 = dict(R.PASS_STATS)
    (c0, r0, d0, a0), _ = helpers.hip_forward(inp, cam, (0.1, 0.2, 0.3), 3, gpu_device, requires_grad=True)
    D = R._LAST_NUM_RENDERED[key][-1]
    assert D == c0.grad_fn.num_rendered > 0
    (c1, r1, d1, a1), _ = helpers.hip_forward(inp, cam, (0.1, 0.2, 0.3), 3, gpu_device, requires_grad=True)   # deferred
    R._LAST_NUM_RENDERED[key].clear(); R._LAST_NUM_RENDERED[key].append(7)                                # overflow
    (c2, r2, d2, a2), _ = helpers.hip_forward(inp, cam, (0.1, 0.2, 0.3), 3, gpu_device, requires_grad=True)
    assert list(R._LAST_NUM_RENDERED[key]) == [7, D] and c2.grad_fn.num_rendered == D
    assert {k: R.PASS_STATS[k] - stats0[k] for k in ("blocking", "deferred", "overflow")} == {"blocking": 1, "deferred": 2, "overflow": 1}
    for c, r, d, a in ((c1, r1, d1, a1), (c2, r2, d2, a2)):
        assert torch.equal(c, c0) and torch.equal(r, r0) and torch.equal(d, d0) and torch.equal(a, a0)
    k0 = helpers.hip_export_binning(c0)
    for cc in (c1, c2):
        k = helpers.hip_export_binning(cc)
        for x, y in zip(k0, k):
            np.testing.assert_array_equal(x, y)
    # and the backward pass works off either
    g = torch.ones_like(c0)
    for cc in (c1, c2):
        cc.backward(g)


def test_view_that_sees_nothing_rendered_twice_runs_the_deferred_phase_on_a_zero_count(gpu_device):
    """ADVICE r3 (high).  A pass whose camera sees nothing records num_rendered = 0, so the NEXT pass at that size is deferred with
    the minimum capacity (4096) and a device-side count of 0.  In drop mode tile 0 of the one-launch tile sort used to return
    before storing the kept count; the later passes, tile_ranges_kernel and pack then read an uninitialised word (torch.empty) as
    their element count -> ranges[] written at garbage tile ids / a stale list blended.  P above the tiny and small limits, G = 1;
    the allocator's free blocks are poisoned before every pass.  Expect: background only, every range empty, both ways of sizing
    the render phase, and a later pass that DOES see the scene at the same size (capacity overflow) unharmed."""
    from opengaussian_amd import rasterizer as R
    W, H, f = 160, 96, 120.0
    P = 3000
    sc, cam = helpers.tiny_scene(P, W, H, f, seed=77)
    inp = helpers.oracle_inputs(sc, cam, use_sh=True)
    hidden = dict(inp)
    m = inp["means3D"].copy(); m[:, 2] = -np.abs(m[:, 2]) - 1.0              # everything behind the camera
    hidden["means3D"] = m
    bg = (0.5, 0.25, 0.125)
    key = (P, W, H, 1)
    R._LAST_NUM_RENDERED.pop(key, None)
    tiles = ((W + 15) // 16) * ((H + 15) // 16)

    def poison():
        junk = [torch.full((n,), -1, dtype=torch.int32, device=gpu_device) for n in (1 << 10, 1 << 14, 1 << 18, 1 << 22)]
        del junk

    stats0 = dict(R.PASS_STATS)
    for sized in ("blocking", "deferred", "deferred"):
        poison()
        (c, r, d, a), _ = helpers.hip_forward(hidden, cam, bg, 3, gpu_device, requires_grad=True)
        assert c.grad_fn.num_rendered == 0 and int(r.abs().sum()) == 0
        want = torch.tensor(bg, device=gpu_device).view(3, 1, 1).expand(3, H, W)
        assert torch.equal(c, want), sized
        assert float(a.abs().max()) == 0.0 and float(d.abs().max()) == 0.0
        image = c.grad_fn.saved_tensors[14]                                  # ImageState: ranges[tiles] come first
        ranges = image[: tiles * 8].view(torch.int32)
        assert int(ranges.abs().sum()) == 0, sized
        c.sum().backward()                                                   # nothing to do, must not fault
    assert R.PASS_STATS["deferred"] - stats0["deferred"] == 2 and R.PASS_STATS["blocking"] - stats0["blocking"] == 1
    # the same size, now visible: the capacity hint (4096) overflows, the render phase is redone with exact buffers
    poison()
    (c1, r1, d1, a1), _ = helpers.hip_forward(inp, cam, bg, 3, gpu_device)
    R._LAST_NUM_RENDERED.pop(key, None)
    (c2, r2, d2, a2), _ = helpers.hip_forward(inp, cam, bg, 3, gpu_device)   # fresh blocking pass
    assert torch.equal(c1, c2) and torch.equal(r1, r2) and torch.equal(a1, a2)
    assert _lib_status_clear()


def _lib_status_clear():
    from opengaussian_amd import _lib
    return _lib.lib().ogs_check_async_status() == 0


def test_chunked_forward_leaves_long_lists_early_and_keeps_parity(gpu_device):
    """Round 4: the forward packs and blends a tile 256 list entries at a time and the workgroup leaves the list when every pixel
    of the tile is done (the reference's block early-out).  A small image behind many large, opaque Gaussians:
    tile lists of well over two chunks that saturate early.  Forward and backward are held to the oracle as everywhere else, and
    the kept-record counts of the pass (qcount[t][4], what the backward's walks and prefetches are bounded by) must show that
    the tiles really stopped: no more records than list entries, whole chunks, and far fewer than the lists are long."""
    W, H, f = 48, 48, 60.0
    P = 6000
    sc, cam = helpers.tiny_scene(P, W, H, f, seed=5, log_scale_mean=-1.6)
    sc.opacities[:] = torch.clamp(sc.opacities, min=0.85)                      # saturate fast
    inp = helpers.oracle_inputs(sc, cam, use_sh=True)
    out = _grad_check(inp, cam, W, H, f, gpu_device, seed=4)
    assert max(out.values()) < GRAD_TOL, out
    (c, r, d, a), _ = helpers.hip_forward(inp, cam, (0.1, 0.2, 0.3), 3, gpu_device, requires_grad=True)
    image = c.grad_fn.saved_tensors[14]
    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    off = lambda n: (n + 255) // 256 * 256
    ranges = image[: tiles * 8].view(torch.int32).view(tiles, 2).cpu().numpy()
    o_qc = off(tiles * 8) + off(W * H * 4)
    qc = image[o_qc:o_qc + tiles * 20].view(torch.int32).view(tiles, 5).cpu().numpy()
    lens = ranges[:, 1] - ranges[:, 0]
    assert lens.min() > 512, lens                                              # the scene is what it is meant to be
    packed = qc[:, 4]
    assert (packed <= lens).all() and (qc[:, :4] <= packed[:, None]).all()
    assert packed.sum() < 0.6 * lens.sum(), (packed, lens)                     # the workgroups left their lists early
    # and the pass is what an unchunked pass computes: same images as the oracle (checked in _grad_check) AND the last contributor
    # of every pixel lies inside what was packed
    keys, rng, ncontrib, plist = helpers.hip_export_binning(c)
    assert int(ncontrib.max()) > 0


def test_noncontiguous_inputs(gpu_device):
    """render() feeds sliced / boolean-indexed views (gaussian_renderer/__init__.py:133,204-212)."""
    from opengaussian_amd.rasterizer import GaussianRasterizer
    W, H, f = 96, 64, 80.0
    sc, cam = helpers.tiny_scene(900, W, H, f, seed=13)
    dev = gpu_device
    rast = GaussianRasterizer(helpers.settings_for(cam, (0, 0, 0), 3, dev))
    feat = sc.ins_feat.to(dev)
    kw = dict(means3D=sc.means3D.to(dev), means2D=torch.zeros(900, 3, device=dev), opacities=sc.opacities.to(dev),
              scales=sc.scales.to(dev), rotations=sc.rotations.to(dev))
    c1, *_ = rast(colors_precomp=feat[:, 3:6], **kw)                      # stride-6 view
    c2, *_ = rast(colors_precomp=feat[:, 3:6].contiguous(), **kw)
    assert torch.equal(c1, c2)
    mask = torch.arange(900, device=dev) % 3 != 0
    c3, r3, *_ = rast(means3D=kw["means3D"][mask], means2D=kw["means2D"][mask], opacities=kw["opacities"][mask],
                      scales=kw["scales"][mask] * 0.5, rotations=kw["rotations"][mask], colors_precomp=feat[:, :3][mask])
    assert c3.shape == (3, H, W) and r3.shape == (int(mask.sum()),)


def test_empty_and_degenerate(gpu_device):
    from opengaussian_amd.rasterizer import GaussianRasterizer
    sc, cam = helpers.tiny_scene(64, 64, 48, 60.0, seed=9)
    dev = gpu_device
    rs = helpers.settings_for(cam, (0.5, 0.25, 0.125), 3, dev)
    rast = GaussianRasterizer(rs)
    # P == 0: zero images, no bg (reference behaviour)
    e = lambda *s: torch.zeros(*s, device=dev)
    c, r, d, a = rast(means3D=e(0, 3), means2D=e(0, 3), opacities=e(0, 1), shs=e(0, 16, 3), scales=e(0, 3), rotations=e(0, 4))
    assert c.shape == (3, 48, 64) and r.shape == (0,) and float(c.abs().max()) == 0.0
    # everything behind the camera: bg only, radii all zero
    m = sc.means3D.clone(); m[:, 2] = -1.0
    c, r, d, a = rast(means3D=m.to(dev), means2D=e(64, 3), opacities=sc.opacities.to(dev), shs=sc.shs.to(dev),
                      scales=sc.scales.to(dev), rotations=sc.rotations.to(dev))
    assert int(r.abs().sum()) == 0
    torch.testing.assert_close(c[:, 0, 0].cpu(), torch.tensor([0.5, 0.25, 0.125]))
    assert float(a.abs().max()) == 0.0
    # P == 1 (the SAM refiner's single-Gaussian footprint query, utils/sam_refinement_utils.py:330-403)
    c, r, d, a = rast(means3D=torch.tensor([[0.0, 0.0, 3.0]], device=dev), means2D=e(1, 3),
                      opacities=torch.tensor([[0.9]], device=dev), shs=sc.shs[:1].to(dev),
                      scales=torch.tensor([[0.05, 0.05, 0.05]], device=dev),
                      rotations=torch.tensor([[1.0, 0, 0, 0]], device=dev))
    assert int(r[0]) > 0 and float(a.max()) > 0.5
    # validation errors of the facade
    with pytest.raises(Exception):
        rast(means3D=m.to(dev), means2D=e(64, 3), opacities=sc.opacities.to(dev), scales=sc.scales.to(dev),
             rotations=sc.rotations.to(dev))
    with pytest.raises(Exception):
        rast(means3D=m.to(dev), means2D=e(64, 3), opacities=sc.opacities.to(dev), shs=sc.shs.to(dev))
    vis = rast.markVisible(sc.means3D.to(dev))
    assert vis.dtype == torch.bool and vis.shape == (64,)
    assert torch.equal(vis.cpu(), sc.means3D[:, 2] > 0.2)


@pytest.mark.parametrize("P,W,H,use_sh", [(1, 96, 64, True), (2, 100, 77, False), (7, 33, 21, True), (63, 160, 96, True),
                                          (64, 128, 80, False), (200, 320, 200, True), (256, 150, 90, True)])
def test_tiny_pass_equals_streaming_path_and_oracle(gpu_device, P, W, H, use_sh):
    """P <= 256 ungrouped passes take the two-launch tiny path (ogs_raster_forward_tiny: one workgroup preprocesses
    + depth-sorts, one workgroup per tile collects its list and blends).  Images, depth, alpha and radii must
    equal the streaming path's bit for bit and the oracle's to 1e-4; backward() on a tiny pass re-renders through
    the streaming path and must give the streaming pass's gradients."""
    from oracle import raster_oracle as ro
    from opengaussian_amd import rasterizer as R
    f = 0.9 * max(W, H)
    sc, cam = helpers.tiny_scene(P, W, H, f, seed=300 + P, log_scale_mean=-2.5, with_ties=True)
    inp = helpers.oracle_inputs(sc, cam, use_sh=use_sh, feat=None if use_sh else sc.ins_feat)
    Cn = 3 if use_sh else 6
    bg = (0.2, 0.1, 0.3)
    st0 = dict(R.PASS_STATS)
    (ct, rt, dt, at), lt = helpers.hip_forward(inp, cam, bg, 3, gpu_device, requires_grad=True, tiny=True)
    assert R.PASS_STATS["tiny"] == st0["tiny"] + 1 and ct.grad_fn.tiny
    (cs, rs_, ds, as_), ls = helpers.hip_forward(inp, cam, bg, 3, gpu_device, requires_grad=True, tiny=False)
    assert R.PASS_STATS["tiny"] == st0["tiny"] + 1 and not cs.grad_fn.tiny
    assert torch.equal(rt, rs_)
    assert torch.equal(ct, cs) and torch.equal(dt, ds) and torch.equal(at, as_)
    ref = ro.render_forward(W=W, H=H, tanfovx=W / (2 * f), tanfovy=H / (2 * f), bg=np.array((bg * 2)[:Cn], np.float32),
                            sh_degree=3, **inp)
    np.testing.assert_array_equal(rt.cpu().numpy(), ref["geom"].radii)
    helpers.assert_close_modulo_threshold_flips(ct.detach().cpu().numpy(), ref["color"], IMG_TOL)
    helpers.assert_close_modulo_threshold_flips(at.detach().cpu().numpy(), ref["alpha"], IMG_TOL)
    helpers.assert_close_modulo_threshold_flips(dt.detach().cpu().numpy(), ref["depth"], IMG_TOL * 10, flip_tol=4e-2)
    g = torch.Generator().manual_seed(P)
    gC, gD, gA = (torch.randn(Cn, H, W, generator=g).to(gpu_device), torch.randn(1, H, W, generator=g).to(gpu_device),
                  torch.randn(1, H, W, generator=g).to(gpu_device))
    torch.autograd.backward([ct, dt, at], [gC, gD, gA])
    assert R.PASS_STATS["tiny_rerendered_for_backward"] == st0["tiny_rerendered_for_backward"] + 1
    torch.autograd.backward([cs, ds, as_], [gC, gD, gA])
    for k in lt:
        if lt[k] is None or lt[k].grad is None:
            assert ls[k] is None or ls[k].grad is None, k
            continue
        assert torch.equal(lt[k].grad, ls[k].grad), k


def test_single_gaussian_footprints_grouped_equals_p1_calls(gpu_device):
    """The SAM refiner renders ONE Gaussian per call, for every Gaussian and camera
    (utils/sam_refinement_utils.py:330-403: render_single_gaussian slices [idx:idx+1] and calls the rasterizer).  Two
    ways through this library: N tiny P = 1 calls, or one grouped pass with group_ids = arange(N).  Same footprints."""
    from opengaussian_amd import rasterizer as R
    from opengaussian_amd.rasterizer import GaussianRasterizer, rasterize_groups
    N, W, H, f = 48, 160, 112, 130.0
    sc, cam = helpers.tiny_scene(N, W, H, f, seed=91, log_scale_mean=-2.0)
    dev = gpu_device
    st = helpers.settings_for(cam, (0.0, 0.0, 0.0), 3, dev)
    t = {k: getattr(sc, k).to(dev) for k in ("means3D", "opacities", "scales", "rotations", "shs")}
    with torch.no_grad():
        color, radii, depth, alpha = rasterize_groups(t["means3D"], torch.zeros(N, 3, device=dev), t["opacities"],
                                                      torch.arange(N, device=dev), N, st, shs=t["shs"], scales=t["scales"],
                                                      rotations=t["rotations"])
    rast = GaussianRasterizer(st)
    tiny0 = R.PASS_STATS["tiny"]
    seen = 0
    for i in range(N):
        m2 = torch.zeros(1, 3, device=dev, requires_grad=True)           # as the refiner does (never back-propagated)
        c, r, d, a = rast(means3D=t["means3D"][i:i + 1], means2D=m2, opacities=t["opacities"][i:i + 1], shs=t["shs"][i:i + 1],
                          scales=t["scales"][i:i + 1], rotations=t["rotations"][i:i + 1])
        assert int(r[0]) == int(radii[i])
        assert torch.equal(c.detach(), color[i]) and torch.equal(a.detach(), alpha[i]) and torch.equal(d.detach(), depth[i])
        seen += int(a.max() > 0)
    assert R.PASS_STATS["tiny"] == tiny0 + N and seen >= N // 2


def _adversarial_scene(kind, P, W, H, f, seed):
    """Scenes built to stress one mechanism each (all compared against the oracle)."""
    g = torch.Generator().manual_seed(seed)
    sc, cam = helpers.tiny_scene(P, W, H, f, seed=seed)
    if kind == "long_lists":            # every Gaussian covers the whole image: tile lists of length P
        sc.means3D[:, 0] = (torch.rand(P, generator=g) - 0.5) * 0.4
        sc.means3D[:, 1] = (torch.rand(P, generator=g) - 0.5) * 0.4
        sc.means3D[:, 2] = torch.rand(P, generator=g) * 2 + 3
        sc.scales[:] = 0.8
        sc.opacities[:] = torch.rand(P, 1, generator=g) * 0.05 + 0.005      # faint: nothing saturates early
    elif kind == "faint_and_opaque":    # opacities below 1/255 (never contribute) mixed with fully opaque ones
        sc.opacities[::2] = 0.003
        sc.opacities[1::2] = 0.999
    elif kind == "offscreen_edges":     # centres outside the frustum whose footprint still reaches the border
        sc.means3D[:, 0] = torch.sign(torch.rand(P, generator=g) - 0.5) * sc.means3D[:, 2] * (W / (2 * f)) * 1.25
        sc.scales[:] = 0.15
    elif kind == "wide_footprint":      # footprints wider than 255 tiles: the duplicate kernel's per-thread fallback
        sc.scales[::7] = 40.0
        sc.opacities[::7] = 0.02
    elif kind == "depth_ties":          # many exactly equal depths: order must fall back to the Gaussian index
        sc.means3D[:, 2] = torch.round(sc.means3D[:, 2])
    elif kind == "needles":             # hundreds of pixels long, ~0.55 px thin (the 0.3 dilation), diagonal: the terms of the
        # quadratic form reach 1e5 .. 1e6 far from the centre while their sum stays O(1) (the reach test's
        # fixed log-space margin is smaller than the fp32 cancellation error there)
        sc.means3D[:, 2] = torch.rand(P, generator=g) * 2 + 3
        sc.means3D[:, 0] = (torch.rand(P, generator=g) - 0.5) * 2.0 * sc.means3D[:, 2] * (W / (2 * f))
        sc.means3D[:, 1] = (torch.rand(P, generator=g) - 0.5) * 2.0 * sc.means3D[:, 2] * (H / (2 * f))
        sc.scales[:, 0] = 3.0 + torch.rand(P, generator=g) * 3.0
        sc.scales[:, 1:] = 1e-4
        ang = (torch.rand(P, generator=g) - 0.5) * 0.5 + 0.785398                 # about the view axis, around 45 degrees
        sc.rotations[:] = torch.stack([torch.cos(ang / 2), torch.zeros(P), torch.zeros(P), torch.sin(ang / 2)], dim=1)
        sc.opacities[:] = 0.5 + 0.49 * torch.rand(P, 1, generator=g)
    return sc, cam


def test_reach_flags_on_needle_gaussians(gpu_device):
    """The (Gaussian, tile) reach test of duplicate_kernel must stay conservative where fp32 cancellation is at its worst:
    needle-shaped splats (sigma ratio ~500 : 1) far from their centre.  Every pair it drops must fail the reach predicate in
    float64 on all 256 pixel centres of its tile; binning stays bit-exact; and a substantial share IS dropped (a needle
    crosses its ceil(3 sigma) square of tiles along one diagonal only)."""
    from oracle import raster_oracle as ro
    W, H, f, P = 640, 480, 300.0, 160
    sc, cam = _adversarial_scene("needles", P, W, H, f, seed=43)
    inp = helpers.oracle_inputs(sc, cam, use_sh=True)
    g = ro.preprocess(inp["means3D"], inp["opacities"], inp["viewmatrix"], inp["projmatrix"], inp["campos"], W, H, W / (2 * f),
                      H / (2 * f), scales=inp["scales"], rotations=inp["rotations"], shs=inp["shs"], sh_degree=3)
    b = ro.bin_tiles(g, W, H)
    lam = 0.5 * (g.conic[:, 0] + g.conic[:, 2])
    assert int(g.radii.max()) > 300 and float((lam[g.radii > 0]).max()) > 1.0          # long AND thin in pixels
    (color, radii, depth, alpha), _ = helpers.hip_forward(inp, cam, (0.0, 0.0, 0.0), 3, gpu_device, requires_grad=True)
    keys, ranges, ncontrib, plist = helpers.hip_export_binning(color)
    np.testing.assert_array_equal(radii.cpu().numpy(), g.radii)
    np.testing.assert_array_equal(keys, b.keys_sorted)
    np.testing.assert_array_equal(plist, b.point_list)
    flags = helpers.LAST_REACH_FLAGS[0]
    n_dropped = helpers.assert_reach_flags_keep_every_contributor(flags, keys, plist, g, W, H)
    assert n_dropped > 0.5 * len(plist), (n_dropped, len(plist))
    a = alpha.detach().cpu().numpy()
    assert np.isfinite(color.detach().cpu().numpy()).all() and a.min() >= 0.0 and a.max() <= 1.0 + 1e-5 and a.max() > 0.3


@pytest.mark.parametrize("kind,P,W,H", [("long_lists", 3000, 48, 32), ("faint_and_opaque", 2000, 128, 80),
                                        ("offscreen_edges", 1500, 112, 64), ("depth_ties", 2500, 128, 96),
                                        ("wide_footprint", 210, 4400, 32)])
def test_adversarial_scenes_forward_and_backward(gpu_device, kind, P, W, H):
    from oracle import raster_oracle as ro
    f = 70.0
    sc, cam = _adversarial_scene(kind, P, W, H, f, seed=41)
    inp = helpers.oracle_inputs(sc, cam, use_sh=True)
    bg = (0.3, 0.2, 0.1)
    ref = ro.render_forward(W=W, H=H, tanfovx=W / (2 * f), tanfovy=H / (2 * f), bg=np.array(bg, np.float32), sh_degree=3, **inp)
    (color, radii, depth, alpha), leaves = helpers.hip_forward(inp, cam, bg, 3, gpu_device, requires_grad=True)
    keys, ranges, ncontrib, plist = helpers.hip_export_binning(color)
    np.testing.assert_array_equal(radii.cpu().numpy(), ref["geom"].radii)
    np.testing.assert_array_equal(keys, ref["binning"].keys_sorted)
    np.testing.assert_array_equal(plist, ref["binning"].point_list)
    np.testing.assert_array_equal(ranges, ref["binning"].ranges)
    helpers.assert_close_modulo_threshold_flips(color.detach().cpu().numpy(), ref["color"], IMG_TOL)
    helpers.assert_close_modulo_threshold_flips(alpha.detach().cpu().numpy(), ref["alpha"], IMG_TOL)
    if kind == "long_lists":
        assert int((ranges[:, 1] - ranges[:, 0]).max()) >= 0.9 * (radii > 0).sum().item()
    if kind == "wide_footprint":
        assert int((ranges[:, 1] > ranges[:, 0]).sum()) == ranges.shape[0] and int(radii.max()) > 255 * 16
    if kind == "depth_ties":
        d = ref["geom"].depth[plist]
        assert (np.diff(d) == 0).sum() > 100       # the tie-break path really is exercised
    # gradients vs float64 autograd
    rng = np.random.default_rng(1)
    gC, gD, gA = rng.standard_normal((3, H, W)), rng.standard_normal((1, H, W)), rng.standard_normal((1, H, W))
    gref = ro.render_backward_f64(inp, ref["binning"], W, H, W / (2 * f), H / (2 * f), np.array(bg, np.float64), gC, gD, gA,
                                  sh_degree=3)
    t = lambda a: torch.tensor(a, dtype=torch.float32, device=gpu_device)
    torch.autograd.backward([color, depth, alpha], [t(gC), t(gD), t(gA)])
    g32 = helpers.lazy(lambda: ro.render_backward_f64(inp, ref["binning"], W, H, W / (2 * f), H / (2 * f), np.array(bg, np.float64),
                                                      gC, gD, gA, sh_degree=3, dtype=torch.float32))
    for k in ("means3D", "scales", "rotations", "opacities", "shs", "means2D"):
        got = leaves[k].grad.cpu().double().numpy()
        want = gref[k].reshape(got.shape)
        e = helpers.assert_grads_close_modulo_threshold_flips(got, want, GRAD_TOL, want_fp32=lambda k=k: g32()[k],
                                                              what=f"{kind} {k}")
        print("adversarial", kind, k, f"{e:.1e} (rows held to the float64 oracle)",
              f"{np.abs(got - want).max() / (np.abs(want).max() + 1e-12):.1e} (all rows)")


@pytest.mark.parametrize("use_sh,G,P", [(False, 5, 4000), (True, 3, 4000), (False, 37, 4000), (True, 4, 900)])
def test_grouped_pass_equals_subset_renders(gpu_device, use_sh, G, P):
    """rasterize_groups (ONE pass, G images) == the reference's per-cluster loop of rasterizer calls on
    boolean-indexed subsets (gaussian_renderer/__init__.py:203-225,327-345): forward bit for bit, gradients up to
    the summation order of the float atomics."""
    from opengaussian_amd.rasterizer import GaussianRasterizer, rasterize_groups
    W, H, f = 150, 90, 110.0            # P = 900: the grouped pass itself takes the one-workgroup geometry phase
    sc, cam = helpers.tiny_scene(P, W, H, f, seed=21)
    st = helpers.settings_for(cam, (0.2, 0.1, 0.3), 3, gpu_device)
    g = torch.Generator().manual_seed(3)
    ids = torch.randint(-1, G, (P,), generator=g).to(gpu_device)           # -1: in no group
    dev = lambda t: t.to(gpu_device).clone().requires_grad_(True)
    names = ["means3D", "opacities", "scales", "rotations"] + (["shs"] if use_sh else ["ins_feat"])
    Cn = 3 if use_sh else 6

    def leaves():
        return {n: dev(getattr(sc, n)) for n in names}

    gC = torch.randn(G, Cn, H, W, generator=g).to(gpu_device)
    gA = torch.randn(G, 1, H, W, generator=g).to(gpu_device)
    gD = torch.randn(G, 1, H, W, generator=g).to(gpu_device)

    A = leaves()
    m2 = torch.zeros(P, 3, device=gpu_device, requires_grad=True)
    color, radii, depth, alpha = rasterize_groups(
        A["means3D"], m2, A["opacities"], ids, G, st, shs=A.get("shs"), colors_precomp=A.get("ins_feat"),
        scales=A["scales"], rotations=A["rotations"])
    assert color.shape == (G, Cn, H, W) and depth.shape == (G, 1, H, W) and alpha.shape == (G, 1, H, W)
    torch.autograd.backward([color, depth, alpha], [gC, gD, gA])

    B = leaves()
    m2b = torch.zeros(P, 3, device=gpu_device, requires_grad=True)
    rast = GaussianRasterizer(st)
    assert int((radii[ids < 0] != 0).sum()) == 0
    for gi in range(G):
        mask = ids == gi
        if int(mask.sum()) == 0:
            # an empty subset: the reference skips the call; the grouped pass returns the background image
            assert float(alpha[gi].abs().max()) == 0.0
            continue
        c, r, d, a = rast(means3D=B["means3D"][mask], means2D=m2b[mask], opacities=B["opacities"][mask],
                          shs=B["shs"][mask] if use_sh else None,
                          colors_precomp=None if use_sh else B["ins_feat"][mask], scales=B["scales"][mask],
                          rotations=B["rotations"][mask])
        assert torch.equal(c, color[gi]) and torch.equal(d, depth[gi]) and torch.equal(a, alpha[gi])
        assert torch.equal(r, radii[mask])
        torch.autograd.backward([c, d, a], [gC[gi], gD[gi], gA[gi]])
    for n in names:
        got, want = A[n].grad, B[n].grad
        scale = float(want.abs().max()) + 1e-20
        assert float((got - want).abs().max()) / scale < 2e-4, n
    assert float((m2.grad - m2b.grad).abs().max()) / (float(m2b.grad.abs().max()) + 1e-20) < 2e-4


@pytest.mark.parametrize("sh_degree,sh_coeffs,scale_modifier,use_cov", [(0, 16, 1.0, False), (1, 16, 0.7, False),
                                                                         (2, 9, 1.3, False), (0, 1, 1.0, True),
                                                                         (3, 16, 0.5, False)])
def test_sh_degrees_coefficient_counts_and_scale_modifier(gpu_device, sh_degree, sh_coeffs, scale_modifier, use_cov):
    """active SH degree below the stored one (the reference raises it every 1000 iterations, train.py:267-268),
    SH tensors with fewer than 16 coefficients, and scale_modifier != 1 (render(..., scaling_modifier)): forward
    parity with the oracle, and gradients against float64 autograd.  The reference's scale gradient omits the
    modifier factor (Appendix A.5(v)), so dL/dscales is compared for modifier 1 only."""
    from oracle import raster_oracle as ro
    from opengaussian_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
    import math
    P, W, H, f = 1800, 144, 88, 110.0
    sc, cam = helpers.tiny_scene(P, W, H, f, seed=50 + sh_degree)
    inp = helpers.oracle_inputs(sc, cam, use_sh=True, use_cov=False)
    inp["shs"] = np.ascontiguousarray(inp["shs"][:, :sh_coeffs])
    if use_cov:
        inp["cov3D_precomp"] = ro.cov3d_from_scale_rot(inp.pop("scales"), inp.pop("rotations"), scale_modifier)
    bg = (0.05, 0.1, 0.15)
    ref = ro.render_forward(W=W, H=H, tanfovx=W / (2 * f), tanfovy=H / (2 * f), bg=np.array(bg, np.float32),
                            sh_degree=sh_degree, scale_modifier=scale_modifier, **inp)
    dev = gpu_device
    t = lambda k: (None if inp.get(k) is None else torch.tensor(np.asarray(inp[k]), dtype=torch.float32, device=dev, requires_grad=True))
    leaves = {k: t(k) for k in ("means3D", "opacities", "scales", "rotations", "cov3D_precomp", "shs")}
    m2 = torch.zeros(P, 3, device=dev, requires_grad=True)
    st = GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5),
        bg=torch.tensor(bg, device=dev), scale_modifier=scale_modifier, viewmatrix=cam.world_view_transform.to(dev),
        projmatrix=cam.full_proj_transform.to(dev), sh_degree=sh_degree, campos=cam.camera_center.to(dev),
        prefiltered=False, debug=True)                      # debug: sync + error check after every kernel
    color, radii, depth, alpha = GaussianRasterizer(st)(
        means3D=leaves["means3D"], means2D=m2, opacities=leaves["opacities"], shs=leaves["shs"],
        scales=leaves["scales"], rotations=leaves["rotations"], cov3D_precomp=leaves["cov3D_precomp"])
    np.testing.assert_array_equal(radii.cpu().numpy(), ref["geom"].radii)
    helpers.assert_close_modulo_threshold_flips(color.detach().cpu().numpy(), ref["color"], IMG_TOL)
    helpers.assert_close_modulo_threshold_flips(alpha.detach().cpu().numpy(), ref["alpha"], IMG_TOL)
    rng = np.random.default_rng(2)
    gC, gD, gA = rng.standard_normal((3, H, W)), rng.standard_normal((1, H, W)), rng.standard_normal((1, H, W))
    gref = ro.render_backward_f64(inp, ref["binning"], W, H, W / (2 * f), H / (2 * f), np.array(bg, np.float64), gC, gD, gA,
                                  sh_degree=sh_degree, scale_modifier=scale_modifier)
    tt = lambda a: torch.tensor(a, dtype=torch.float32, device=dev)
    torch.autograd.backward([color, depth, alpha], [tt(gC), tt(gD), tt(gA)])
    for k in ("means3D", "opacities", "shs", "rotations", "cov3D_precomp", "scales"):
        if leaves[k] is None or gref.get(k) is None:
            continue
        if k == "scales" and scale_modifier != 1.0:
            continue
        got = leaves[k].grad.cpu().double().numpy()
        want = gref[k].reshape(got.shape)
        helpers.assert_grads_close_modulo_threshold_flips(
            got, want, GRAD_TOL, what=k, want_fp32=lambda k=k: ro.render_backward_f64(
                inp, ref["binning"], W, H, W / (2 * f), H / (2 * f), np.array(bg, np.float64), gC, gD, gA, sh_degree=sh_degree,
                scale_modifier=scale_modifier, dtype=torch.float32)[k])
    if sh_coeffs > (sh_degree + 1) ** 2:                  # coefficients above the active degree get exact zeros
        assert float(leaves["shs"].grad[:, (sh_degree + 1) ** 2:].abs().max()) == 0.0


def test_random_configurations_backward_parity(gpu_device):
    """12 seeded random configurations of the BACKWARD against float64 autograd through the oracle: image sizes that are
    no multiples of 16, 40 .. 2500 Gaussians (both sides of the one-workgroup geometry phase), footprints from a few
    pixels to most of the image (many partial batches of the matrix-core reduction, moment shifts over hundreds of
    pixels), SH / precomputed colours / 6-D features, covariance input, near-threshold opacities."""
    rng = np.random.default_rng(77)
    worst = {}
    for it in range(12):
        W, H = int(rng.integers(40, 170)), int(rng.integers(30, 120))
        P = int(rng.choice([40, 300, 900, 1024, 1025, 2500]))
        f = float(rng.uniform(0.5, 1.4) * max(W, H))
        lsm = float(rng.uniform(-4.0, -1.2))
        mode = it % 3                                             # 0: SH colours, 1: 6-D features, 2: SH + cov3D input
        sc, cam = helpers.tiny_scene(P, W, H, f, seed=4000 + it, log_scale_mean=lsm, with_ties=bool(it % 4 == 0))
        if it % 4 == 1:
            sc.opacities[:] = torch.rand(sc.opacities.shape, generator=torch.Generator().manual_seed(5000 + it)) ** 3
        if mode == 1:
            inp, bg = helpers.oracle_inputs(sc, cam, feat=sc.ins_feat), (0.0,) * 6
        else:
            inp, bg = helpers.oracle_inputs(sc, cam, use_sh=True, use_cov=(mode == 2)), tuple(float(x) for x in rng.uniform(0, 1, 3))
        tag = f"config {it}: {W}x{H} P={P} f={f:.1f} lsm={lsm:.2f} mode={mode}"
        try:
            errs = _grad_check(inp, cam, W, H, f, gpu_device, bg=bg, seed=it)
        except AssertionError as e:
            raise AssertionError(f"{tag}: {e}") from None
        assert errs, tag
        for k, e in errs.items():
            assert e < GRAD_TOL, f"{tag}: {k}: relative error {e} (all: {errs})"
            worst[k] = max(worst.get(k, 0.0), e)
    print("random backward parity, worst per family:", {k: f"{e:.1e}" for k, e in worst.items()})


def test_random_configurations_forward_parity(gpu_device):
    """24 seeded random configurations (image sizes incl. non-multiples of 16 and single-tile images, point counts
    from 1 to a few thousand, focal lengths, scale / opacity statistics, SH or precomputed colours, covariance
    input, background): integers bit-exact, images within 1e-4 (modulo documented threshold flips)."""
    from oracle import raster_oracle as ro
    rng = np.random.default_rng(2024)
    for it in range(24):
        W, H = int(rng.integers(5, 200)), int(rng.integers(5, 140))
        P = int(rng.choice([1, 2, 7, 63, 64, 65, 300, 1500, 4000]))
        f = float(rng.uniform(0.4, 1.6) * max(W, H))
        lsm = float(rng.uniform(-4.5, -1.5))
        use_sh, use_cov = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        bg = tuple(float(x) for x in rng.uniform(0, 1, 3))
        sc, cam = helpers.tiny_scene(P, W, H, f, seed=1000 + it, log_scale_mean=lsm, with_ties=bool(it % 3 == 0))
        if it % 4 == 1:
            sc.opacities[:] = torch.rand(sc.opacities.shape, generator=torch.Generator().manual_seed(6000 + it)) ** 3   # many near-threshold opacities
        inp = helpers.oracle_inputs(sc, cam, use_sh=use_sh, use_cov=use_cov)
        ref = ro.render_forward(W=W, H=H, tanfovx=W / (2 * f), tanfovy=H / (2 * f), bg=np.array(bg, np.float32),
                                sh_degree=3, **inp)
        (_c, radii, _d, _a), _ = helpers.hip_forward(inp, cam, bg, 3, gpu_device, requires_grad=True)
        color, depth, alpha = _c.detach(), _d.detach(), _a.detach()
        keys, ranges, ncontrib, plist = (helpers.hip_export_binning(_c) if ref["binning"].num_rendered > 0 else (None,) * 4)
        tag = f"config {it}: {W}x{H} P={P} f={f:.1f} lsm={lsm:.2f} sh={use_sh} cov={use_cov}"
        np.testing.assert_array_equal(radii.cpu().numpy(), ref["geom"].radii, err_msg=tag)
        if keys is not None:
            np.testing.assert_array_equal(keys, ref["binning"].keys_sorted, err_msg=tag)
            np.testing.assert_array_equal(plist, ref["binning"].point_list, err_msg=tag)
            np.testing.assert_array_equal(ranges, ref["binning"].ranges, err_msg=tag)
        try:
            helpers.assert_close_modulo_threshold_flips(color.cpu().numpy(), ref["color"], IMG_TOL)
            helpers.assert_close_modulo_threshold_flips(alpha.cpu().numpy(), ref["alpha"], IMG_TOL)
            helpers.assert_close_modulo_threshold_flips(depth.cpu().numpy(), ref["depth"], IMG_TOL * 10, flip_tol=4e-2)
        except AssertionError as e:
            raise AssertionError(f"{tag}: {e}") from None


@pytest.mark.parametrize("env", [{"OGS_BLEND_ROWS": "0"}, {"OGS_BLEND_FOLD": "bf16"},
                                 {"OGS_BLEND_FEAT_LDS": "0"}, {"OGS_PACK_FUSED": "0"}, {"OGS_PACK_FUSED": "2"}])
def test_alternative_blend_kernels_keep_parity(gpu_device, env):
    """The forward blend exists in two structures: the quadrant walk (records in SGPRs) and the per-4x4-block walk (records in
    VGPRs through LDS, the default).  The full backward reduces its gradient records on the matrix cores with exact-fp32 MFMAs by
    default and with a two-term bf16 split of both operands under OGS_BLEND_FOLD=bf16 (same speed: the kernel is bound by its
    atomics; 2^-15 per product shows when a Gaussian's sum cancels, scripts/fuzz_parity.py); the forward packs and blends a tile chunk by chunk with a workgroup-wide exit by default
    (OGS_PACK_FUSED=2: the whole list packed first, round 3; =0: two launches); the
    features-only backward walks quadrants with its records through LDS by default, through scalar loads otherwise; pack and
    forward blend of a tile run in one workgroup by default, as two launches otherwise.  The
    non-default ones are selected by environment variables read once per process, so they are checked in a child process:
    forward / backward parity against the oracle, the adversarial scenes, the fused and grouped passes."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_10_raster_gpu.py"), "-x", "-q", "-m", "gpu", "-k",
           "test_forward_parity or test_backward_parity or test_adversarial_scenes or test_fused_pass or test_grouped_pass or test_tiny_pass"
           " or test_features_only_backward"]
    r = subprocess.run(cmd, cwd=root, env=dict(os.environ, **env), capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:]
    assert " passed" in r.stdout and " failed" not in r.stdout
