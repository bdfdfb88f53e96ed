"""GPU: opengaussian_amd.renderer.render() (fused passes) reproduces the result dict of the reference's
render() structure (gaussian_renderer/__init__.py:22-373), re-enacted here pass by pass -- 4 separate
3-channel rasterizer calls, 2 per cluster -- through the same drop-in GaussianRasterizer."""
import math
import types

import pytest
import torch

from tests import helpers

pytestmark = pytest.mark.gpu


class FakeGaussians:
    """Only what render() reads from scene/gaussian_model.py:GaussianModel (:122-169)."""

    def __init__(self, sc, dev):
        self._xyz = sc.means3D.to(dev).requires_grad_(True)
        self._scaling = sc.scales.to(dev).requires_grad_(True)
        self._rotation = sc.rotations.to(dev).requires_grad_(True)
        self._opacity = sc.opacities.to(dev).requires_grad_(True)
        self._features = sc.shs.to(dev).requires_grad_(True)
        self._ins_feat = (sc.ins_feat.to(dev) * 2 - 1).requires_grad_(True)
        self.active_sh_degree = 3
        self.max_sh_degree = 3

    get_xyz = property(lambda s: s._xyz)
    get_scaling = property(lambda s: s._scaling)
    get_rotation = property(lambda s: s._rotation)
    get_opacity = property(lambda s: s._opacity)
    get_features = property(lambda s: s._features)

    def get_ins_feat(self, origin=False):
        return torch.nn.functional.normalize(self._ins_feat, dim=1)

    def leaves(self):
        return [self._xyz, self._scaling, self._rotation, self._opacity, self._features, self._ins_feat]


def reference_structured_render(cam, pc, bg, rescale_factor, cluster_idx=None, selected_root_id=None):
    """The reference's pass structure (:104-163, :184-232), with the drop-in 3-channel rasterizer."""
    from opengaussian_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
    rs = GaussianRasterizationSettings(
        image_height=cam.image_height, image_width=cam.image_width, tanfovx=math.tan(cam.FoVx * 0.5),
        tanfovy=math.tan(cam.FoVy * 0.5), bg=bg, scale_modifier=1.0, viewmatrix=cam.world_view_transform,
        projmatrix=cam.full_proj_transform, sh_degree=pc.active_sh_degree, campos=cam.camera_center, prefiltered=False,
        debug=False)
    rast = GaussianRasterizer(rs)
    m2 = torch.zeros_like(pc.get_xyz, requires_grad=True)
    kw = dict(means3D=pc.get_xyz, means2D=m2, opacities=pc.get_opacity, rotations=pc.get_rotation, cov3D_precomp=None)
    img, radii, depth, alpha = rast(shs=pc.get_features, colors_precomp=None, scales=pc.get_scaling, **kw)
    feat = (pc.get_ins_feat() + 1) / 2
    f1, _, _, _ = rast(shs=None, colors_precomp=feat[:, :3], scales=pc.get_scaling * rescale_factor, **kw)
    f2, _, _, _ = rast(shs=None, colors_precomp=feat[:, 3:6], scales=pc.get_scaling * rescale_factor, **kw)
    _, _, _, sil = rast(shs=pc.get_features, colors_precomp=None, scales=pc.get_scaling * rescale_factor, **kw)
    out = {"render": img, "alpha": alpha, "depth": depth, "silhouette": sil, "ins_feat": torch.cat((f1, f2), 0),
           "radii": radii, "viewspace_points": m2}
    if cluster_idx is not None:
        sel = (cluster_idx == selected_root_id) & (radii > 0)
        c1, _, _, _ = rast(means3D=pc.get_xyz[sel], means2D=m2[sel], shs=None, colors_precomp=feat[:, :3][sel],
                           opacities=pc.get_opacity[sel], scales=pc.get_scaling[sel] * rescale_factor,
                           rotations=pc.get_rotation[sel], cov3D_precomp=None)
        c2, _, _, csil = rast(means3D=pc.get_xyz[sel], means2D=m2[sel], shs=None, colors_precomp=feat[:, 3:][sel],
                              opacities=pc.get_opacity[sel], scales=pc.get_scaling[sel] * rescale_factor,
                              rotations=pc.get_rotation[sel], cov3D_precomp=None)
        out["cluster_img"] = torch.cat((c1, c2), 0)
        out["cluster_sil"] = csil
    return out


@pytest.mark.parametrize("seed,expect_rescale", [(0, False), (1, True)])
def test_render_matches_reference_pass_structure(gpu_device, seed, expect_rescale):
    from opengaussian_amd.renderer import render
    dev = gpu_device
    W, H, f = 176, 112, 130.0
    sc, cam = helpers.tiny_scene(3000, W, H, f, seed=21)
    cam = cam.to(dev)
    bg = torch.tensor([0.2, 0.1, 0.3], device=dev)
    pipe = types.SimpleNamespace(debug=False, compute_cov3D_python=False, convert_SHs_python=False)
    cluster_idx = (torch.arange(3000, device=dev) % 4)

    # find out what the reference's RNG draws would be for this seed, then replay them on both sides
    torch.manual_seed(seed)
    prob = torch.rand(1)
    factor = torch.rand(1) if prob > 0.5 else torch.tensor(1.0)
    assert bool(prob > 0.5) == expect_rescale

    pc_a = FakeGaussians(sc, dev)
    torch.manual_seed(seed)
    out = render(cam, pc_a, pipe, bg, iteration=1, cluster_idx=cluster_idx, render_cluster=True, selected_root_id=2)
    pc_b = FakeGaussians(sc, dev)
    ref = reference_structured_render(cam, pc_b, bg, factor.to(dev), cluster_idx, 2)

    assert set(out) == {"render", "alpha", "depth", "silhouette", "ins_feat", "cluster_imgs", "cluster_silhouettes",
                        "leaf_clusters_imgs", "leaf_cluster_silhouettes", "occured_leaf_id", "cluster_occur",
                        "viewspace_points", "visibility_filter", "radii"}
    assert torch.equal(out["radii"], ref["radii"]) and torch.equal(out["visibility_filter"], ref["radii"] > 0)
    for k in ("render", "alpha", "depth", "silhouette", "ins_feat"):
        assert out[k].shape == ref[k].shape, k
        torch.testing.assert_close(out[k], ref[k], atol=2e-6, rtol=0, msg=lambda m, k=k: f"{k}: {m}")
    assert out["ins_feat"].shape == (6, H, W) and out["silhouette"].shape == (1, H, W)
    assert len(out["cluster_imgs"]) == 1 and bool(out["cluster_occur"][2])
    torch.testing.assert_close(out["cluster_imgs"][0], ref["cluster_img"], atol=2e-6, rtol=0)
    torch.testing.assert_close(out["cluster_silhouettes"], ref["cluster_sil"], atol=2e-6, rtol=0)

    # gradients of a loss on render + ins_feat + silhouette agree (float atomics and the 2x3 vs 1x6 channel grouping
    # reorder fp32 sums: 1e-3 of the largest entry)
    g = torch.Generator().manual_seed(1)
    gi, gf, gs = (torch.randn(3, H, W, generator=g).to(dev), torch.randn(6, H, W, generator=g).to(dev),
                  torch.randn(1, H, W, generator=g).to(dev))
    ((out["render"] * gi).sum() + (out["ins_feat"] * gf).sum() + (out["silhouette"] * gs).sum()).backward()
    ((ref["render"] * gi).sum() + (ref["ins_feat"] * gf).sum() + (ref["silhouette"] * gs).sum()).backward()
    for a, b, name in zip(pc_a.leaves(), pc_b.leaves(), ("xyz", "scaling", "rotation", "opacity", "features", "ins_feat")):
        scale = float(b.grad.abs().max()) + 1e-12
        assert float((a.grad - b.grad).abs().max()) / scale < 1e-3, name
    ga, gb = out["viewspace_points"].grad, ref["viewspace_points"].grad
    assert float((ga - gb).abs().max()) / float(gb.abs().max()) < 1e-3


def test_render_stage1_detached_geometry_runs_features_only_backward(gpu_device):
    """train.py:431-436: from stage 1 on xyz / SH / opacity / scaling / rotation are detached and only `_ins_feat`
    trains.  render() then stops asking for viewspace_points.grad (consumed by densification only, train.py:594-598)
    and the rasterizer runs its features-only backward: dL/d ins_feat must equal what the all-gradients graph gives,
    nothing else receives a gradient; viewspace_grad=True restores the reference's means2D gradient."""
    from opengaussian_amd.renderer import render
    dev = gpu_device
    W, H, f = 160, 96, 120.0
    sc, cam = helpers.tiny_scene(2500, W, H, f, seed=21)
    cam = cam.to(dev)
    pipe = types.SimpleNamespace(debug=False, compute_cov3D_python=False, convert_SHs_python=False)
    bg = torch.tensor([0.2, 0.1, 0.3], device=dev)
    g = torch.Generator().manual_seed(4)
    gF, gR = torch.randn(6, H, W, generator=g).to(dev), torch.randn(3, H, W, generator=g).to(dev)

    def run(detach, **kw):
        pc = FakeGaussians(sc, dev)
        if detach:                                             # what train.py:431-436 does to the model
            for name in ("_xyz", "_scaling", "_rotation", "_opacity", "_features"):
                setattr(pc, name, getattr(pc, name).detach())
        torch.manual_seed(0)                                   # prob <= 0.5: no rescale -> the fused 9-channel pass
        out = render(cam, pc, pipe, bg, iteration=1, **kw)
        ((out["ins_feat"] * gF).sum() + (out["render"] * gR).sum()).backward()
        return out, pc

    out_d, pc_d = run(True)
    assert out_d["viewspace_points"].grad is None
    out_f, pc_f = run(False)
    assert out_f["viewspace_points"].grad is not None
    assert torch.equal(out_d["ins_feat"], out_f["ins_feat"]) and torch.equal(out_d["render"], out_f["render"])
    scale = float(pc_f._ins_feat.grad.abs().max())
    assert float((pc_d._ins_feat.grad - pc_f._ins_feat.grad).abs().max()) / scale < 2e-5
    out_v, pc_v = run(True, viewspace_grad=True)
    torch.testing.assert_close(out_v["viewspace_points"].grad, out_f["viewspace_points"].grad, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(pc_v._ins_feat.grad, pc_f._ins_feat.grad, rtol=1e-5, atol=1e-7)


def test_render_post_process_and_leaf_path(gpu_device):
    from opengaussian_amd.renderer import render
    dev = gpu_device
    W, H, f = 96, 64, 80.0
    sc, cam = helpers.tiny_scene(1200, W, H, f, seed=22)
    cam = cam.to(dev)
    pipe = types.SimpleNamespace(debug=False, compute_cov3D_python=False, convert_SHs_python=True)
    pc = FakeGaussians(sc, dev)
    leaf_idx = torch.arange(1200, device=dev) % 6
    with torch.no_grad():
        out = render(cam, pc, pipe, torch.zeros(3, device=dev), iteration=1, rescale=False,
                     leaf_cluster_idx=leaf_idx, selected_leaf_id=torch.tensor([1, 4], device=dev), post_process=True)
    assert out["occured_leaf_id"] == [1] and len(out["leaf_clusters_imgs"]) == 1
    assert out["leaf_clusters_imgs"][0].shape == (6, H, W) and out["leaf_cluster_silhouettes"].shape == (1, H, W)
    assert out["render"].shape == (3, H, W) and float(out["render"].max()) > 0


@pytest.mark.parametrize("mode", ["leaf_range", "leaf_seg_rgb", "coarse_better_vis", "leaf_all_premask"])
def test_batched_cluster_loops_equal_per_subset_calls(gpu_device, mode):
    """render()'s coarse / fine cluster loops through grouped rasterizer passes == the reference's one call per
    boolean-indexed subset (gaussian_renderer/__init__.py:179-236,245-356): same kept ids, identical images."""
    from opengaussian_amd import renderer as R
    dev = gpu_device
    W, H, f = 112, 80, 90.0
    P = 6000
    sc, cam = helpers.tiny_scene(P, W, H, f, seed=31, log_scale_mean=-3.5)
    cam = cam.to(dev)
    pipe = types.SimpleNamespace(debug=False, compute_cov3D_python=False, convert_SHs_python=False)
    root_num, leaf_num = 4, 5
    g = torch.Generator().manual_seed(2)
    leaf_idx = torch.randint(0, root_num * leaf_num, (P,), generator=g).to(dev)
    leaf_idx[leaf_idx == 7] = 6                       # leaf 7 empty, leaf 13 almost empty (< 10 points)
    few = torch.nonzero(leaf_idx == 13).flatten()
    leaf_idx[few[5:]] = 12
    cluster_idx = leaf_idx // leaf_num
    kw = dict(iteration=1, rescale=False, render_feat_map=False, root_num=root_num, leaf_num=leaf_num)
    if mode == "leaf_range":
        kw.update(leaf_cluster_idx=leaf_idx, selected_root_id=2)
    elif mode == "leaf_seg_rgb":
        kw.update(leaf_cluster_idx=leaf_idx, selected_root_id=1, seg_rgb=True)
    elif mode == "leaf_all_premask":
        kw.update(leaf_cluster_idx=leaf_idx, pre_mask=(torch.arange(P, device=dev) % 3 != 0))
    else:
        kw.update(cluster_idx=cluster_idx, render_cluster=True, better_vis=True)
    outs = []
    for batch in (True, False):
        R.BATCH_SUBSETS = batch
        old = R.GROUPS_PER_PASS
        R.GROUPS_PER_PASS = 6                         # several passes + a single-subset tail in "leaf_all_premask"
        try:
            pc = FakeGaussians(sc, dev)
            out = R.render(cam, pc, pipe, torch.tensor([0.1, 0.0, 0.2], device=dev), **kw)
            key_img = "cluster_imgs" if mode == "coarse_better_vis" else "leaf_clusters_imgs"
            key_sil = "cluster_silhouettes" if mode == "coarse_better_vis" else "leaf_cluster_silhouettes"
            loss = sum((im * (i + 1)).sum() for i, im in enumerate(out[key_img])) + out[key_sil].sum()
            loss.backward()
            outs.append((out, pc, key_img, key_sil))
        finally:
            R.BATCH_SUBSETS = True
            R.GROUPS_PER_PASS = old
    (a, pa, ki, ks), (b, pb, _, _) = outs
    assert len(a[ki]) == len(b[ki]) and len(a[ki]) >= 2
    if mode != "coarse_better_vis":
        assert a["occured_leaf_id"] == b["occured_leaf_id"]
        assert 7 not in a["occured_leaf_id"] and 13 not in a["occured_leaf_id"]
    else:
        assert torch.equal(a["cluster_occur"], b["cluster_occur"])
    for x, y in zip(a[ki], b[ki]):
        assert x.shape == y.shape and torch.equal(x, y)
    assert torch.equal(a[ks], b[ks])
    for la, lb, name in zip(pa.leaves(), pb.leaves(), ("xyz", "scaling", "rotation", "opacity", "features", "ins_feat")):
        if lb.grad is None:
            assert la.grad is None or float(la.grad.abs().max()) == 0.0, name
            continue
        scale = float(lb.grad.abs().max()) + 1e-20
        assert float((la.grad - lb.grad).abs().max()) / scale < 5e-4, name


# ---- render() against fixtures produced by RUNNING the reference's own render() over the CPU oracle -------------------
# tests/golden/make_render_golden.py (build container only) -> tests/golden/render_golden.npz; inputs are rebuilt from the
# seeds by tests/golden/render_cases.py on both sides.  Images 1e-4 (depth 1e-3, un-normalised sum of z * w), integers and
# list structure exact, gradients 2e-4 of the family's largest entry AND 2e-2 relative on every row above 1 % of it.
def _golden():
    import os
    import numpy as np
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "render_golden.npz"))


def _render_case_names():
    from tests.golden import render_cases as rc
    return list(rc.CASES)


@pytest.mark.parametrize("name", _render_case_names())
def test_render_matches_reference_run_fixture(gpu_device, name):
    import numpy as np
    from opengaussian_amd.renderer import render
    from tests.golden import render_cases as rc
    dev = gpu_device
    gold = _golden()
    case = rc.build(name)
    has = lambda k: f"{name}/{k}" in gold.files
    want = lambda k: gold[f"{name}/{k}"]
    pc = rc.TinyModel(case["params"], dev, requires_grad=case["grads"])
    cam = case["cam"].to(dev)
    if cam.bClusterOccur is not None:
        cam.bClusterOccur = cam.bClusterOccur.to(dev)
    kwargs = rc.move_kwargs(case["kwargs"], dev)
    torch.manual_seed(case["rng_seed"])                     # the reference's global-RNG draws (:121-124), replayed
    with (torch.enable_grad() if case["grads"] else torch.no_grad()):
        out = render(cam, pc, case["pipe"], case["bg"].to(dev), 1, **kwargs)
        loss = rc.fixed_loss(out) if case["grads"] else None
        if loss is not None:
            loss.backward()
    assert list(out) == ["render", "alpha", "depth", "silhouette", "ins_feat", "cluster_imgs", "cluster_silhouettes",
                         "leaf_clusters_imgs", "leaf_cluster_silhouettes", "occured_leaf_id", "cluster_occur",
                         "viewspace_points", "visibility_filter", "radii"]
    # integers, masks, ids: exact
    assert np.array_equal(out["radii"].cpu().numpy(), want("radii"))
    assert np.array_equal(out["visibility_filter"].cpu().numpy(), want("visibility_filter"))
    assert tuple(out["viewspace_points"].shape) == tuple(want("viewspace_points_shape"))
    if has("cluster_occur/none"):
        assert out["cluster_occur"] is None
    else:
        assert np.array_equal(out["cluster_occur"].cpu().numpy(), want("cluster_occur"))
    if has("occured_leaf_id/none"):
        assert out["occured_leaf_id"] is None
    else:
        assert [int(x) for x in out["occured_leaf_id"]] == want("occured_leaf_id").tolist()
    # images
    for k in rc.TENSOR_KEYS:
        v = out[k]
        if has(f"{k}/none"):
            assert v is None, k
        elif has(f"{k}/emptylist"):
            assert isinstance(v, list) and len(v) == 0, k
        else:
            assert tuple(v.shape) == want(k).shape, k
            helpers.assert_close_modulo_threshold_flips(v.detach().cpu().numpy(), want(k), tol=1e-3 if k == "depth" else 1e-4,
                                                        flip_tol=4e-2 if k == "depth" else 4e-3)
    for k in rc.LIST_KEYS:
        v = out[k]
        if has(f"{k}/none"):
            assert v is None, k
            continue
        assert isinstance(v, list) and len(v) == int(want(f"{k}/len")), k
        for i, img in enumerate(v):
            assert tuple(img.shape) == want(f"{k}/{i}").shape
            helpers.assert_close_modulo_threshold_flips(img.detach().cpu().numpy(), want(f"{k}/{i}"))
    # gradients of the fixed loss through the whole call
    if case["grads"]:
        assert abs(float(loss.detach()) - float(want("loss"))) <= 2e-4 * max(1.0, abs(float(want("loss"))))
        fam = {n: getattr(pc, n).grad for n in rc.PARAM_NAMES}
        fam["viewspace_points"] = out["viewspace_points"].grad
        for n, g in fam.items():
            w = want(f"grad/{n}").astype(np.float64)
            got = np.zeros_like(w) if g is None else g.detach().cpu().double().numpy()
            helpers.assert_grad_family_close(got, w, what=f"{name}:{n}")
