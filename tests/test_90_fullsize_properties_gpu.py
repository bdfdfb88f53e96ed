"""GPU: BASELINE.json configurations at their real sizes.

* C2 (100k Gaussians, 800x800): forward parity against the CPU oracle (images 1e-4, integers bit-exact).
* S1M-1080p (1M Gaussians, 1920x1080): the oracle would take minutes, so the size-independent properties
  of the domain are checked instead: sortedness and partition of the binning state, conservation
  (alpha + T_final), determinism of the forward pass, fused == separate passes, linearity of the backward
  pass in the upstream gradient, and run-to-run agreement of the float-atomic gradients."""
import math

import numpy as np
import pytest
import torch

from opengaussian_amd.synthetic import make_camera, make_scene
from tests import helpers

pytestmark = pytest.mark.gpu


def test_c2_forward_parity_vs_oracle(gpu_device):
    from oracle import raster_oracle as ro
    torch.set_flush_denormal(True)
    P, W, H, f = 100_000, 800, 800, 700.0
    sc = make_scene(P, W, H, f, f, seed=0)
    cam = make_camera(W, H, f, f)
    inp = helpers.oracle_inputs(sc, cam, use_sh=True)
    ref = ro.render_forward(W=W, H=H, tanfovx=W / (2 * f), tanfovy=H / (2 * f), bg=np.zeros(3, np.float32),
                            sh_degree=3, **inp)
    (color, radii, depth, alpha), _ = helpers.hip_forward(inp, cam, (0, 0, 0), 3, gpu_device, requires_grad=True)
    keys, ranges, ncontrib, plist = helpers.hip_export_binning(color)
    np.testing.assert_array_equal(radii.cpu().numpy(), ref["geom"].radii)
    np.testing.assert_array_equal(keys, ref["binning"].keys_sorted)
    np.testing.assert_array_equal(plist, ref["binning"].point_list)
    np.testing.assert_array_equal(ranges, ref["binning"].ranges)
    # 1e-4 everywhere except threshold flips: where the device exp and the host exp land on different sides of
    # alpha >= 1/255 (or T < 1e-4) one contribution of size <= alpha*T*c ~ 4e-3 appears/disappears.  At 1.9M
    # values a handful of such pixels exist (3 were observed); they are the same pixels n_contrib disagrees on.
    def close(got, want, tol, flip):
        diff = np.abs(got - want)
        assert (diff > tol).mean() < 1e-5, f"{(diff > tol).sum()} values off by more than {tol}"
        assert diff.max() < flip, f"max diff {diff.max()}"
    close(color.detach().cpu().numpy(), ref["color"], 1e-4, 4e-3)
    close(alpha.detach().cpu().numpy(), ref["alpha"], 1e-4, 4e-3)
    close(depth.detach().cpu().numpy(), ref["depth"], 1e-3, 4e-2)
    assert (ncontrib != ref["n_contrib"].astype(np.uint32)).mean() < 2e-3


@pytest.fixture(scope="module")
def s1m(gpu_device):
    P, W, H, f = 1_000_000, 1920, 1080, 1000.0
    sc = make_scene(P, W, H, f, f, seed=0).to(gpu_device)
    cam = make_camera(W, H, f, f).to(gpu_device)
    return sc, cam, W, H, f


def _render(sc, cam, dev, feats=None, requires_grad=False):
    from opengaussian_amd.rasterizer import GaussianRasterizer, rasterize_fused
    rs = helpers.settings_for(cam, (0.0, 0.0, 0.0), 3, dev)
    leaves = {k: getattr(sc, k).detach().clone().requires_grad_(requires_grad)
              for k in ("means3D", "scales", "rotations", "opacities", "shs", "ins_feat")}
    m2 = torch.zeros_like(leaves["means3D"], requires_grad=requires_grad)
    if feats == "fused":
        out = rasterize_fused(leaves["means3D"], m2, leaves["opacities"], leaves["shs"], leaves["ins_feat"], rs,
                              scales=leaves["scales"], rotations=leaves["rotations"])
    elif feats == "feat":
        out = GaussianRasterizer(rs)(means3D=leaves["means3D"], means2D=m2, opacities=leaves["opacities"],
                                     colors_precomp=leaves["ins_feat"], scales=leaves["scales"], rotations=leaves["rotations"])
    else:
        out = GaussianRasterizer(rs)(means3D=leaves["means3D"], means2D=m2, opacities=leaves["opacities"], shs=leaves["shs"],
                                     scales=leaves["scales"], rotations=leaves["rotations"])
    return out, leaves, m2


def test_s1m_binning_invariants_and_conservation(gpu_device, s1m):
    sc, cam, W, H, f = s1m
    (color, radii, depth, alpha), leaves, _ = _render(sc, cam, gpu_device, requires_grad=True)
    keys, ranges, ncontrib, plist = helpers.hip_export_binning(color)
    D = len(keys)
    assert D == 7_416_179                                                  # same D as the CPU oracle computes
    assert (np.diff(keys.astype(np.uint64)) >= 0).all()                     # globally sorted (tile, depth bits)
    tiles = (keys >> np.uint64(32)).astype(np.int64)
    lens = ranges[:, 1].astype(np.int64) - ranges[:, 0]
    assert lens.sum() == D and (lens >= 0).all()
    nz = np.nonzero(lens)[0]
    assert (tiles[ranges[nz, 0]] == nz).all() and (tiles[ranges[nz, 1] - 1] == nz).all()
    # stable tie-break: equal keys keep ascending Gaussian index
    same = np.diff(keys.astype(np.uint64)) == 0
    assert (np.diff(plist.astype(np.int64))[same] > 0).all()
    # depth bits of the key are the depth of the listed Gaussian; radii > 0 exactly for listed Gaussians
    assert set(np.unique(plist).tolist()) == set(np.nonzero(radii.cpu().numpy() > 0)[0].tolist())
    a = alpha.detach()
    assert float(a.min()) >= 0.0 and float(a.max()) <= 1.0 - 1e-4 + 1e-5    # blending stops before T < 1e-4
    assert torch.isfinite(color).all() and torch.isfinite(depth).all()
    gx = (W + 15) // 16
    per_tile_max = torch.nn.functional.max_pool2d(torch.from_numpy(ncontrib.astype(np.float32))[None, None], 16, ceil_mode=True)[0, 0]
    assert (per_tile_max.numpy().reshape(-1) <= lens.reshape(-1, gx).reshape(-1)).all()


def test_s1m_forward_deterministic_and_fused_equals_separate(gpu_device, s1m):
    sc, cam, W, H, f = s1m
    with torch.no_grad():
        (c1, r1, d1, a1), _, _ = _render(sc, cam, gpu_device)
        (c2, r2, d2, a2), _, _ = _render(sc, cam, gpu_device)
        assert torch.equal(c1, c2) and torch.equal(d1, d2) and torch.equal(a1, a2) and torch.equal(r1, r2)
        (cf, _, _, af), _, _ = _render(sc, cam, gpu_device, feats="fused")
        (cB, _, _, _), _, _ = _render(sc, cam, gpu_device, feats="feat")
    torch.testing.assert_close(cf[:3], c1, atol=1e-6, rtol=0)
    torch.testing.assert_close(cf[3:], cB, atol=1e-6, rtol=0)
    torch.testing.assert_close(af, a1, atol=1e-6, rtol=0)


def test_s1m_backward_linearity_and_repeatability(gpu_device, s1m):
    sc, cam, W, H, f = s1m
    g = torch.Generator().manual_seed(3)
    g1 = torch.randn(3, H, W, generator=g).to(gpu_device)
    g2 = torch.randn(3, H, W, generator=g).to(gpu_device)

    def grads(gc):
        (color, _, _, alpha), leaves, m2 = _render(sc, cam, gpu_device, requires_grad=True)
        torch.autograd.backward([color], [gc])
        return {k: v.grad for k, v in leaves.items() if v.grad is not None} | {"means2D": m2.grad}

    ga, gb, gab, ga2 = grads(g1), grads(g2), grads(g1 + g2), grads(g1)
    for k in ga:
        scale = float(gab[k].abs().max()) + 1e-12
        # the backward pass is linear in dL/dcolor
        assert float((ga[k] + gb[k] - gab[k]).abs().max()) / scale < 1e-3, k
        # float atomics: same inputs agree run to run to rounding
        assert float((ga[k] - ga2[k]).abs().max()) / scale < 1e-4, k
