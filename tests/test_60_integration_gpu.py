"""GPU: the pieces together, as a stage-1 iteration of the reference wires them (train.py:425-456,594-611):
render() -> mask_feature_mean / cohesion / separation on the rendered feature map -> backward through the
rasterizer -> FusedAdam on the instance features; then the two-level k-means on the learned features."""
import types

import pytest
import torch

from tests import helpers
from tests.test_11_render_gpu import FakeGaussians

pytestmark = pytest.mark.gpu


def test_stage1_iterations_reduce_the_loss_and_only_touch_ins_feat(gpu_device):
    from opengaussian_amd import mask_ops as mk
    from opengaussian_amd.kmeans import Quantize_kMeans
    from opengaussian_amd.optim import FusedAdam
    from opengaussian_amd.renderer import render
    dev = gpu_device
    W, H, f, P = 160, 112, 120.0, 6000
    sc, cam = helpers.tiny_scene(P, W, H, f, seed=77, log_scale_mean=-3.2)
    cam = cam.to(dev)
    pc = FakeGaussians(sc, dev)
    pipe = types.SimpleNamespace(debug=False, compute_cov3D_python=False, convert_SHs_python=False)
    bg = torch.zeros(3, device=dev)
    # SAM-like label image: 4 x 3 blocky regions, label 0 = invalid
    lab = (torch.arange(H, device=dev)[:, None] // 40) * 4 + (torch.arange(W, device=dev)[None, :] // 40) + 1
    lab[:, :8] = 0
    masks = torch.stack([lab == (n + 1) for n in range(int(lab.max()))])
    opt = FusedAdam([{"params": [pc._ins_feat], "lr": 0.02, "name": "ins_feat"}], lr=0.0, eps=1e-15)
    frozen = [t.detach().clone() for t in (pc._xyz, pc._scaling, pc._rotation, pc._opacity, pc._features)]
    losses = []
    torch.manual_seed(0)
    for it in range(12):
        # stage 1 freezes everything but the instance features (train.py:431-436)
        geo = types.SimpleNamespace(
            get_xyz=pc._xyz.detach(), get_scaling=pc._scaling.detach(), get_rotation=pc._rotation.detach(),
            get_opacity=pc._opacity.detach(), get_features=pc._features.detach(),
            get_ins_feat=pc.get_ins_feat, active_sh_degree=3, max_sh_degree=3)
        out = render(cam, geo, pipe, bg, iteration=it, rescale=False)
        feat, sil = out["ins_feat"], out["silhouette"]
        mean = mk.mask_feature_mean(feat, masks, image_mask=sil)
        loss = mk.separation_loss(mean, it) + 0.1 * mk.cohesion_loss(feat, masks, mean)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        assert pc._ins_feat.grad is not None and torch.isfinite(pc._ins_feat.grad).all()
        opt.step()
        losses.append(float(loss.detach()))
    # the features start to separate the regions: steady decrease (12 Adam steps at lr 0.02 buy ~5 %)
    assert losses[-1] < losses[0] * 0.97 and all(b <= a + 1e-3 for a, b in zip(losses, losses[1:])), losses
    for a, b in zip(frozen, (pc._xyz, pc._scaling, pc._rotation, pc._opacity, pc._features)):
        assert torch.equal(a, b.detach())                            # nothing else moved
    # two-level codebook on the learned features (train.py:586-588)
    q = Quantize_kMeans(num_clusters=8, num_leaf_clusters=3, num_iters=5, dim=9)
    g = types.SimpleNamespace(_xyz=pc._xyz.detach(), _ins_feat=pc._ins_feat, _ins_feat_q=None)
    q.forward(g, 1, assign=True, mode="root", pos_weight=0.5)
    assert q.centers.shape == (8, 9) and q.nn_index.shape == (P,) and int(q.nn_index.max()) < 8
    assert g._ins_feat_q.shape == (P, 6) and g._ins_feat_q.requires_grad


class _Model:
    """The slice of scene/gaussian_model.py:GaussianModel a stage-0 iteration touches: raw parameters + activations
    (:122-169), the seven-group optimizer (:216-230) and the densification state, with the method bodies replaced
    as INTEGRATION.md section 7b shows."""

    def __init__(self, sc, dev):
        from opengaussian_amd import densify
        from opengaussian_amd.optim import FusedAdam
        P = torch.nn.Parameter
        self._xyz = P(sc.means3D.to(dev))
        self._features_dc = P(sc.shs[:, :1].contiguous().to(dev))
        self._features_rest = P(sc.shs[:, 1:].contiguous().to(dev))
        self._opacity = P(torch.logit(sc.opacities.clamp(1e-4, 1 - 1e-4)).to(dev))
        self._scaling = P(torch.log(sc.scales).to(dev))
        self._rotation = P(sc.rotations.to(dev))
        self._ins_feat = P((sc.ins_feat * 2 - 1).to(dev))
        self.active_sh_degree = self.max_sh_degree = 3
        lrs = {"xyz": 1.6e-4, "f_dc": 2.5e-3, "f_rest": 1.25e-4, "opacity": 0.05, "scaling": 5e-3, "rotation": 1e-3, "ins_feat": 1e-3}
        self.optimizer = FusedAdam([{"params": [getattr(self, a)], "lr": lrs[n], "name": n} for n, a in self.ATTR.items()],
                                   lr=0.0, eps=1e-15)
        n = self._xyz.shape[0]
        self.state = densify.DensifyState(self.optimizer, torch.zeros(n, 1, device=dev), torch.zeros(n, 1, device=dev),
                                          torch.zeros(n, device=dev), 0.01)

    ATTR = {"xyz": "_xyz", "f_dc": "_features_dc", "f_rest": "_features_rest", "opacity": "_opacity", "scaling": "_scaling",
            "rotation": "_rotation", "ins_feat": "_ins_feat"}
    get_xyz = property(lambda s: s._xyz)
    get_scaling = property(lambda s: torch.exp(s._scaling))
    get_rotation = property(lambda s: torch.nn.functional.normalize(s._rotation))
    get_opacity = property(lambda s: torch.sigmoid(s._opacity))
    get_features = property(lambda s: torch.cat((s._features_dc, s._features_rest), dim=1))

    def get_ins_feat(self, origin=False):
        return torch.nn.functional.normalize(self._ins_feat, dim=1)

    def adopt(self, params):
        for n, a in self.ATTR.items():
            setattr(self, a, params[n])


def test_stage0_iterations_with_densification(gpu_device):
    """Stage 0 as train.py:352-358,497-498,594-611 wires it: render() -> L1 loss -> backward through the rasterizer
    (all gradient families) -> max_radii2D / add_densification_stats -> densify_and_prune every few iterations ->
    FusedAdam over the seven groups.  The pieces must compose: the optimizer state follows the row map, the point
    count changes between passes (capacity hint, gradient record sizes), the loss keeps going down."""
    from opengaussian_amd import densify
    from opengaussian_amd.renderer import render
    dev = gpu_device
    W, H, f, P = 160, 112, 120.0, 5000
    sc, cam = helpers.tiny_scene(P, W, H, f, seed=5, log_scale_mean=-3.0)
    cam = cam.to(dev)
    target_sc, _ = helpers.tiny_scene(P, W, H, f, seed=6, log_scale_mean=-3.0)
    pipe = types.SimpleNamespace(debug=False, compute_cov3D_python=False, convert_SHs_python=False)
    bg = torch.zeros(3, device=dev)
    with torch.no_grad():
        target = render(cam, _Model(target_sc, dev), pipe, bg, iteration=0, rescale=False, render_feat_map=False)["render"]
    m = _Model(sc, dev)
    losses, counts = [], []
    torch.manual_seed(0)
    for it in range(1, 25):
        out = render(cam, m, pipe, bg, iteration=it, rescale=False, render_feat_map=False)
        loss = (out["render"] - target).abs().mean()
        m.optimizer.zero_grad(set_to_none=True)
        loss.backward()
        losses.append(float(loss.detach()))
        counts.append(int(m._xyz.shape[0]))
        vsp, radii = out["viewspace_points"], out["radii"]
        assert vsp.grad is not None and vsp.grad.shape == (counts[-1], 3)
        # train.py:597-598 in one pass: max_radii2D = max(., radii) on visible points, gradient-norm statistics
        densify.add_densification_stats(m.state, vsp.grad, None, radii)
        assert float(m.state.denom.max()) >= 1.0 and float(m.state.max_radii2D.max()) >= float(radii.max())
        if it % 8 == 0:                                       # train.py:600-602
            thr = float((m.state.xyz_gradient_accum / m.state.denom.clamp_min(1)).quantile(0.9))
            m.adopt(densify.densify_and_prune(m.state, thr, 0.005, 4.0, 20))
            plan = m.state.last_plan
            assert plan["clones"] + plan["split_children"] > 0 and m._xyz.shape[0] == plan["kept"] + plan["clones"] + plan["split_children"]
            for n, a in m.ATTR.items():                       # the optimizer follows: same tensors, aligned moments
                p = getattr(m, a)
                assert m.optimizer.param_groups[list(m.ATTR).index(n)]["params"][0] is p
                st = m.optimizer.state.get(p, {})
                if n == "ins_feat":                           # never rendered here: no gradient, no Adam state (as in torch)
                    assert "exp_avg" not in st
                else:
                    assert st["exp_avg"].shape == p.shape and st["exp_avg_sq"].shape == p.shape
        m.optimizer.step()                                    # train.py:608-609 (fresh parameters have no gradient: skipped)
    assert len(set(counts)) >= 3, counts                       # the point count really changed between passes
    assert min(losses[-4:]) < losses[0], losses
    assert all(torch.isfinite(getattr(m, a)).all() for a in m.ATTR.values())
    m.adopt({**m.state.params(), **densify.reset_opacity(m.state)})          # train.py:604-605
    assert float(torch.sigmoid(m._opacity.detach()).max()) <= 0.01 + 1e-6
    out = render(cam, m, pipe, bg, iteration=99, rescale=False, render_feat_map=False)
    out["render"].sum().backward()
    m.optimizer.step()


def test_stage1_training_with_kept_passes_follows_the_uncached_trajectory(gpu_device):
    """Stage-1 iterations over three cameras drawn in a shuffled order (train.py:296-299), everything but `_ins_feat` detached
    (train.py:431-436), rescale=False (train.py:346-350), with the frozen-geometry cache on and off: same losses, same learned
    features (the re-blend returns a full pass' bits; the feature gradients are fp64 sums rounded once), and after the first
    sweep every pass is a re-blend."""
    from opengaussian_amd import mask_ops as mk
    from opengaussian_amd import rasterizer as R
    from opengaussian_amd.optim import FusedAdam
    from opengaussian_amd.renderer import render
    from opengaussian_amd.synthetic import make_scene, orbit_camera
    from tests.test_14_kept_pass_gpu import ReferenceShapedGaussians
    dev = gpu_device
    W, H, f, P = 176, 128, 130.0, 8000
    sc = make_scene(P, W, H, f, f, seed=31, log_scale_mean=-3.2)
    cams = [orbit_camera(W, H, f, f, view_index=v, num_views=3).to(dev) for v in range(3)]
    pipe = types.SimpleNamespace(debug=False, compute_cov3D_python=False, convert_SHs_python=False)
    bg = torch.zeros(3, device=dev)
    lab = (torch.arange(H, device=dev)[:, None] // 44) * 4 + (torch.arange(W, device=dev)[None, :] // 44) + 1
    masks = torch.stack([lab == (n + 1) for n in range(int(lab.max()))])
    order = [0, 2, 1, 1, 0, 2, 2, 1, 0, 0, 2, 1]

    def train(budget):
        saved, R.KEPT_PASSES = R.KEPT_PASSES, R.KeptPasses(budget_bytes=budget)
        try:
            pc = ReferenceShapedGaussians(sc, dev)
            opt = FusedAdam([{"params": [pc._ins_feat], "lr": 0.02, "name": "ins_feat"}], lr=0.0, eps=1e-15)
            n0, losses = R.PASS_STATS["reblend"], []
            for it, v in enumerate(order):
                for name in ("_xyz", "_scaling", "_rotation", "_opacity", "_features_dc", "_features_rest"):
                    setattr(pc, name, getattr(pc, name).detach())                 # train.py:431-436, every iteration
                out = render(cams[v], pc, pipe, bg, iteration=30001 + it, rescale=False)
                mean = mk.mask_feature_mean(out["ins_feat"], masks, image_mask=out["silhouette"])
                loss = mk.separation_loss(mean, it) + 0.1 * mk.cohesion_loss(out["ins_feat"], masks, mean)
                opt.zero_grad(set_to_none=True)
                loss.backward()
                opt.step()
                losses.append(float(loss.detach()))
            return losses, pc._ins_feat.detach().clone(), R.PASS_STATS["reblend"] - n0
        finally:
            R.KEPT_PASSES = saved

    pc0_feat = (sc.ins_feat.to(dev) * 2 - 1)                                  # where the features start
    ref_losses, ref_feat, ref_reblends = train(0)
    got_losses, got_feat, got_reblends = train(1 << 30)
    assert ref_reblends == 0 and got_reblends == len(order) - 3          # one full pass per camera, then re-blends
    assert got_losses[-1] < got_losses[0]
    torch.testing.assert_close(torch.tensor(got_losses), torch.tensor(ref_losses), rtol=1e-5, atol=1e-7)
    # Adam with eps = 1e-15 turns a last-bit difference of a near-zero gradient into a step of its own (measured: 51 of 48 000
    # entries off by <= 3.2e-5 after 12 steps of lr 0.02, i.e. of up to 0.24 of movement); a wrong gradient moves entries by 1e-2
    torch.testing.assert_close(got_feat, ref_feat, rtol=0.0, atol=2e-4)
    assert float((got_feat - ref_feat).abs().max()) < 2e-4 < 0.01 * float((got_feat - pc0_feat).abs().max())
