"""GPU: the HIP mask reductions (opengaussian_amd/mask_ops.py -> include/ogs_mask.h) against the goldens of the
reference's functions and, at full image size, against the CPU oracle.  fp32 sums in a different order:
relative tolerance 2e-5 on values, 1e-4 on gradients (of the largest entry)."""
import os

import numpy as np
import pytest
import torch

from tests.golden.make_mask_golden import CASES, case_inputs

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "mask_golden.npz"))


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"s{c[0]}")
def test_values_match_reference_goldens(gpu_device, case):
    from opengaussian_amd import mask_ops as mk
    seed, C, H, W, N, overlap = case
    feat, masks, sil, masks2 = (t.to(gpu_device) for t in case_inputs(*case))
    k = f"s{seed}"
    near = lambda a, b, rt=2e-5: np.testing.assert_allclose(a.detach().cpu().numpy(), b, rtol=rt, atol=1e-6)
    near(mk.mask_feature_mean(feat, masks, image_mask=sil), GOLD[k + "_mean_w"])
    near(mk.mask_feature_mean(feat, masks), GOLD[k + "_mean"])
    # the permuted int64 one-hot layout of get_SAM_mask_and_feat (opengs_utlis.py:147-149)
    onehot = masks.permute(1, 2, 0).long().contiguous().permute(2, 0, 1)
    assert not onehot.is_contiguous()
    near(mk.mask_feature_mean(feat, onehot, image_mask=sil), GOLD[k + "_mean_int64"])
    mean, var, cnt = mk.mask_feature_mean(feat, masks, return_var=True)
    near(mean, GOLD[k + "_mean"]); near(var, GOLD[k + "_var"], 2e-4)
    np.testing.assert_array_equal(cnt.cpu().numpy(), GOLD[k + "_cnt"])
    mean_w = torch.from_numpy(GOLD[k + "_mean_w"]).to(gpu_device)
    np.testing.assert_allclose(float(mk.cohesion_loss(feat, masks, mean_w)), float(GOLD[k + "_cohesion"]), rtol=2e-5)
    for base in (None, "former", "later"):
        near(mk.calculate_iou(masks, masks2, base=base), GOLD[k + f"_iou_{base}"], 1e-5)


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"s{c[0]}")
def test_stage1_loss_gradient_matches_reference(gpu_device, case):
    """train.py:450-456: loss = separation + 0.1 * cohesion, differentiated w.r.t. the rendered feature map."""
    from opengaussian_amd import mask_ops as mk
    seed, C, H, W, N, overlap = case
    feat, masks, sil, _ = (t.to(gpu_device) for t in case_inputs(*case))
    fm = feat.clone().requires_grad_(True)
    sw = sil.clone().requires_grad_(True)             # the silhouette is a rasterizer output: part of the graph
    mean_w = mk.mask_feature_mean(fm, masks, image_mask=sw)
    loss = mk.separation_loss(mean_w, 1000) + 0.1 * mk.cohesion_loss(fm, masks, mean_w)
    loss.backward()
    want = GOLD[f"s{seed}_dfeat"]
    assert np.abs(fm.grad.cpu().numpy() - want).max() <= 1e-4 * np.abs(want).max()
    want = GOLD[f"s{seed}_dsil"]
    assert np.abs(sw.grad.cpu().numpy() - want).max() <= 1e-4 * np.abs(want).max()
    np.testing.assert_allclose(float(loss.detach()), float(GOLD[f"s{seed}_separation"]) + 0.1 * float(GOLD[f"s{seed}_cohesion"]), rtol=2e-5)


def _big_case(H, W, N, C, seed):
    g = torch.Generator().manual_seed(seed)
    feat = torch.rand(C, H, W, generator=g)
    coarse = torch.randint(0, N + 1, ((H + 15) // 16, (W + 15) // 16), generator=g)
    labels = coarse.repeat_interleave(16, 0).repeat_interleave(16, 1)[:H, :W]
    masks = torch.stack([labels == (n + 1) for n in range(N)])
    sil = torch.rand(1, H, W, generator=g)
    return feat, masks, sil


@pytest.mark.parametrize("H,W,N,C", [(1080, 1920, 96, 6), (484, 648, 40, 6), (127, 333, 9, 3)])
def test_full_size_against_oracle(gpu_device, H, W, N, C):
    """full image sizes (incl. H*W not a multiple of 4 -> scalar load path) vs the CPU oracle, values + gradient"""
    from opengaussian_amd import mask_ops as mk
    from oracle import mask_oracle as mo
    feat, masks, sil = _big_case(H, W, N, C, seed=7)
    fm_ref = feat.clone().requires_grad_(True)
    mean_ref = mo.mask_feature_mean(fm_ref, masks, image_mask=sil)
    coh_ref = mo.cohesion_loss(fm_ref, masks, mean_ref)
    (mean_ref.square().sum() + coh_ref).backward()
    fm = feat.to(gpu_device).requires_grad_(True)
    mg, sg = masks.to(gpu_device), sil.to(gpu_device)
    mean = mk.mask_feature_mean(fm, mg, image_mask=sg)
    coh = mk.cohesion_loss(fm, mg, mean)
    (mean.square().sum() + coh).backward()
    np.testing.assert_allclose(mean.detach().cpu().numpy(), mean_ref.detach().numpy(), rtol=3e-5, atol=1e-6)
    np.testing.assert_allclose(float(coh.detach()), float(coh_ref.detach()), rtol=3e-5)
    want = fm_ref.grad.numpy()
    assert np.abs(fm.grad.cpu().numpy() - want).max() <= 1e-4 * np.abs(want).max()
    # size-independent property: the masks partition the labelled pixels, so the per-mask weighted sums add up to
    # the weighted sum over all labelled pixels (checked in float64 on the GPU values)
    ones = torch.ones(1, H, W, device=gpu_device)
    cnt = mk.mask_feature_mean(ones.expand(3, H, W).contiguous(), mg, image_mask=sg, return_var=True)[2]
    wsum = (mean.detach().double() * torch.maximum(cnt.double(), torch.ones_like(cnt).double())[:, None]).sum(0)
    # counts returned are clamped at 1; the clamp only bites on empty masks, whose sums are 0 anyway
    labelled = mg.any(dim=0)
    direct = (fm.detach().double() * sg.double() * labelled).sum(dim=(1, 2))
    nonempty = (mg.flatten(1).any(dim=1)).double()
    wsum_true = (mean.detach().double() * (cnt.double() * nonempty)[:, None]).sum(0)
    torch.testing.assert_close(wsum_true, direct, rtol=1e-4, atol=1e-3)


def test_edge_cases(gpu_device):
    from opengaussian_amd import mask_ops as mk
    feat = torch.rand(6, 20, 28, device=gpu_device)
    none = torch.zeros(0, 20, 28, dtype=torch.bool, device=gpu_device)
    assert mk.mask_feature_mean(feat, none).shape == (0, 6)
    empty = torch.zeros(3, 20, 28, dtype=torch.bool, device=gpu_device)
    m = mk.mask_feature_mean(feat, empty)
    assert m.shape == (3, 6) and float(m.abs().max()) == 0.0                 # counts clamp at 1, sums 0
    assert float(mk.cohesion_loss(feat, empty, m)) == 0.0
    full = torch.ones(1, 20, 28, dtype=torch.bool, device=gpu_device)
    torch.testing.assert_close(mk.mask_feature_mean(feat, full)[0], feat.mean(dim=(1, 2)), rtol=1e-5, atol=1e-6)
    with pytest.raises(RuntimeError):
        mk.mask_feature_mean(torch.rand(5, 20, 28, device=gpu_device), full)  # C = 5 unsupported: loud, no fallback
    # a pixel exactly at its mask mean: zero distance, zero (not NaN) gradient
    const = torch.full((6, 8, 8), 0.25, device=gpu_device, requires_grad=True)
    one = torch.ones(1, 8, 8, dtype=torch.bool, device=gpu_device)
    mu = mk.mask_feature_mean(const, one)
    mk.cohesion_loss(const, one, mu).backward()
    assert torch.isfinite(const.grad).all() and float(const.grad.abs().max()) == 0.0


@pytest.mark.parametrize("N,Cc", [(2, 6), (32, 6), (96, 6), (200, 3), (257, 9)])
def test_separation_loss_fused_equals_torch_formulation(gpu_device, N, Cc):
    """ogs_separation_loss (two launches: pairwise inverse distances, rank of every element inside its row, rank
    weights, sum and gradient) against the literal torch formulation of train.py:124-155 (argsort().argsort()), early
    and late (iteration > 35 000) weights, value and gradient; and the degenerate case of identical rows (empty masks
    share the zero mean: exact ties, where the reference's unstable argsort is implementation defined)."""
    from opengaussian_amd import mask_ops as mk
    g = torch.Generator().manual_seed(100 + N)
    base = torch.rand(N, Cc, generator=g)
    for it in (1000, 40000):
        a = base.clone().to(gpu_device).requires_grad_(True)
        b = base.clone().to(gpu_device).requires_grad_(True)
        la = mk.separation_loss(a, it)
        lb = mk._separation_loss_torch(b, it)
        (la * 3.0).backward()
        (lb * 3.0).backward()
        torch.testing.assert_close(la.detach(), lb.detach(), rtol=2e-6, atol=1e-7)
        # every gradient element is a sum of N signed terms: compare at the scale of the largest one.  Two inverse
        # distances of a row that differ in the last bit may swap ranks between the two evaluations of |m_i - m_j|^2
        # (a near-tie decision, as in k-means): that moves ONE weight pair by 0.9 / (N-1), i.e. a gradient element by at
        # most 2 * 2 * 0.9 / (N-1) / (N (N-1)) * max |d| inv^2 <= 3.6 / ((N-1)^2 N) -- allowed once per element
        flip = 3.6 / ((N - 1) ** 2 * N)
        assert float((a.grad - b.grad).abs().max()) <= 1e-5 * float(b.grad.abs().max()) + flip
        # no gradient requested: value only
        assert float(mk.separation_loss(base.to(gpu_device), it)) == float(la.detach())
    tied = base.clone()
    tied[: N // 2] = 0.0                                   # identical rows
    t1 = mk.separation_loss(tied.to(gpu_device).requires_grad_(True), 1000)
    t2 = mk.separation_loss(tied.to(gpu_device).requires_grad_(True), 1000)
    assert torch.isfinite(t1) and float(t1.detach()) == float(t2.detach())   # deterministic, stable tie-breaking
