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
