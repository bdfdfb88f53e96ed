"""CPU: the mask-reduction oracle (oracle/mask_oracle.py) against the goldens produced by the reference's own
functions (tests/golden/make_mask_golden.py), plus the torch-only host functions of the product module."""
import os

import numpy as np
import pytest
import torch

from tests.golden.make_mask_golden import CASES, case_inputs

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "mask_golden.npz"))


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"s{c[0]}")
def test_oracle_matches_reference_goldens(case):
    from oracle import mask_oracle as mo
    seed, C, H, W, N, overlap = case
    feat, masks, sil, masks2 = case_inputs(*case)
    k = f"s{seed}"
    np.testing.assert_allclose(mo.mask_feature_mean(feat, masks, image_mask=sil).numpy(), GOLD[k + "_mean_w"], rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(mo.mask_feature_mean(feat, masks).numpy(), GOLD[k + "_mean"], rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(mo.mask_feature_mean(feat, masks.long(), image_mask=sil).numpy(), GOLD[k + "_mean_int64"], rtol=2e-5, atol=1e-6)
    mean, var, cnt = mo.mask_feature_mean(feat, masks, return_var=True)
    np.testing.assert_allclose(var.numpy(), GOLD[k + "_var"], rtol=1e-4, atol=1e-6)
    np.testing.assert_array_equal(cnt.numpy(), GOLD[k + "_cnt"])
    mean_w = torch.from_numpy(GOLD[k + "_mean_w"])
    np.testing.assert_allclose(float(mo.cohesion_loss(feat, masks, mean_w)), float(GOLD[k + "_cohesion"]), rtol=2e-5)
    for base in (None, "former", "later"):
        np.testing.assert_allclose(mo.calculate_iou(masks, masks2, base=base).numpy(), GOLD[k + f"_iou_{base}"], rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"s{c[0]}")
def test_oracle_gradient_of_stage1_loss_matches_reference(case):
    """d(separation + 0.1 * cohesion)/d feat_map through autograd of the oracle == the reference's (train.py:450-456)."""
    from oracle import mask_oracle as mo
    from opengaussian_amd.mask_ops import separation_loss
    seed, C, H, W, N, overlap = case
    feat, masks, sil, _ = case_inputs(*case)
    fm = feat.clone().requires_grad_(True)
    sw = sil.clone().requires_grad_(True)
    mean_w = mo.mask_feature_mean(fm, masks, image_mask=sw)
    loss = separation_loss(mean_w, 1000) + 0.1 * mo.cohesion_loss(fm, masks, mean_w)
    loss.backward()
    want = GOLD[f"s{seed}_dfeat"]
    assert np.abs(fm.grad.numpy() - want).max() <= 2e-5 * np.abs(want).max() + 1e-9
    want = GOLD[f"s{seed}_dsil"]
    assert np.abs(sw.grad.numpy() - want).max() <= 2e-5 * np.abs(want).max() + 1e-9


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"s{c[0]}")
def test_host_side_torch_functions_match_reference(case):
    from opengaussian_amd.mask_ops import pair_mask_feature_mean, separation_loss
    seed, C, H, W, N, overlap = case
    feat, masks, sil, _ = case_inputs(*case)
    k = f"s{seed}"
    mean_w = torch.from_numpy(GOLD[k + "_mean_w"])
    np.testing.assert_allclose(float(separation_loss(mean_w, 1000)), float(GOLD[k + "_separation"]), rtol=1e-5)
    np.testing.assert_allclose(float(separation_loss(mean_w, 40000)), float(GOLD[k + "_sep_late"]), rtol=1e-5)
    pm = torch.rand(N, C, H, W, generator=torch.Generator().manual_seed(seed))
    np.testing.assert_allclose(pair_mask_feature_mean(pm, masks).numpy(), GOLD[k + "_pair"], rtol=1e-5, atol=1e-7)


def test_mask_ops_refuse_cpu_tensors():
    from opengaussian_amd import mask_ops
    feat, masks, sil, _ = case_inputs(*CASES[0])
    with pytest.raises(RuntimeError):
        mask_ops.mask_feature_mean(feat, masks)
    with pytest.raises(RuntimeError):
        mask_ops.cohesion_loss(feat, masks, torch.zeros(masks.shape[0], feat.shape[0]))
    with pytest.raises(RuntimeError):
        mask_ops.calculate_iou(masks, masks)
