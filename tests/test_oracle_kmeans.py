"""CPU: the k-means oracle is PINNED against vectors produced by running the reference's own
Quantize_kMeans (tests/golden/make_kmeans_golden.py).  Tolerances: centres 1e-4;
ids exact except rows whose two best distances tie within rounding of the reference's matmul-based cdist."""
import os

import numpy as np
import pytest
import torch

from oracle import kmeans_oracle as ko
from tests.golden.make_kmeans_golden import NUM_ITERS, POS_WEIGHT, case_inputs

GOLD = os.path.join(os.path.dirname(__file__), "golden", "kmeans_golden.npz")
ID_MISMATCH_FRAC = 1e-3
CENTER_TOL = 1e-4


def assert_centers_close(got, want, k_note=""):
    """Centres within 1e-4 -- except that ONE point changing cluster on a near-tie (the reference's cdist goes
    through a matmul, ours is a direct sum of squares) moves two centres by ~|x|/n.  Allow at most
    max(2, 5%) such rows, each bounded by 0.05."""
    diff = np.abs(np.asarray(got, np.float64) - np.asarray(want, np.float64)).max(axis=1)
    bad = int((diff > CENTER_TOL).sum())
    assert bad <= max(2, int(0.05 * len(diff))), f"{bad} centre rows differ by > {CENTER_TOL} {k_note}: {diff.max()}"
    assert diff.max() < 0.05, f"centre row off by {diff.max()} {k_note}"


def _cases():
    g = np.load(GOLD)
    return [tuple(int(v) for v in row) for row in g["cases"]]


def run_oracle_case(seed, N, k1, k2):
    ins_feat, xyz, init_root, init_leaf, sub = case_inputs(seed, N, k1, k2)
    feat9 = torch.cat((ins_feat, xyz * POS_WEIGHT), dim=1).numpy()
    o = ko.KMeansOracle(k1, k2, NUM_ITERS)
    o.centers = feat9[init_root.numpy()].copy()
    o.assign_root(feat9)
    return o, ins_feat, init_leaf, sub


@pytest.mark.parametrize("seed,N,k1,k2", _cases())
def test_oracle_matches_reference_golden(seed, N, k1, k2):
    g = np.load(GOLD)
    key = lambda name: g[f"s{seed}_n{N}_{name}"]
    ins_feat, xyz, *_ = case_inputs(seed, N, k1, k2)
    np.testing.assert_allclose(key("input_checksum"), [float(ins_feat.double().sum()), float(xyz.double().sum())],
                               rtol=0, atol=1e-9, err_msg="seeded inputs drifted: regenerate the goldens")
    o, ins_feat, init_leaf, sub = run_oracle_case(seed, N, k1, k2)
    ids_ref = key("root_ids").astype(np.int64)
    assert (o.nn_index != ids_ref).mean() <= ID_MISMATCH_FRAC
    assert_centers_close(o.centers, key("root_centers"), "root")
    row_ok = np.abs(o.centers - key("root_centers")).max(axis=1) <= CENTER_TOL
    same = (o.nn_index == ids_ref) & row_ok[o.nn_index]
    np.testing.assert_allclose(o.quantized("root")[same], key("root_q")[same], atol=CENTER_TOL, rtol=0)
    # leaf level, continuing from the REFERENCE's coarse ids so both sides see the same subsets
    o.cls_ids = ids_ref
    o.iLeafSubNum = sub.numpy()
    o.leaf_centers = ins_feat.numpy()[init_leaf.numpy()].copy()
    for c in key("leaf_sel"):
        o.assign_leaf(ins_feat.numpy(), int(c))
    leaf_ref = key("leaf_ids").astype(np.int64)
    assert (o.leaf_cls_ids != leaf_ref).mean() <= ID_MISMATCH_FRAC
    assert_centers_close(o.leaf_centers, key("leaf_centers"), "leaf")
    # never-visited coarse clusters keep the dummy id k1*k2 (kmeans_quantize.py:160)
    untouched = ~np.isin(ids_ref, key("leaf_sel"))
    assert (o.leaf_cls_ids[untouched] == k1 * k2).all()
    # cluster_len bookkeeping of equalize_cluster_size (:130,138)
    np.testing.assert_array_equal(np.bincount(leaf_ref, minlength=k1 * k2 + 1), key("cluster_len_leaf"))


@pytest.mark.parametrize("seed,N,k1,k2", _cases())
def test_oracle_follows_reference_trajectory_step_by_step(seed, N, k1, k2):
    """Every Lloyd iteration of the reference's own trajectory (root_centers_iter / root_ids_iter, produced by
    running the reference with num_iters = 1..5) re-done by the oracle FROM THE REFERENCE'S centres: each id and
    centre difference is attributed to a near-tie row (tests/helpers.py::kmeans_step_attribution)."""
    from tests import helpers
    g = np.load(GOLD)
    key = lambda name: g[f"s{seed}_n{N}_{name}"]
    ins_feat, xyz, init_root, _, _ = case_inputs(seed, N, k1, k2)
    feat9 = torch.cat((ins_feat, xyz * POS_WEIGHT), dim=1).numpy()
    traj_c, traj_i = key("root_centers_iter"), key("root_ids_iter").astype(np.int64)
    c_prev = feat9[init_root.numpy()].copy()
    for t in range(NUM_ITERS):
        ids_pre = ko._argmin_sqdist(feat9, c_prev)
        c_next, _ = ko.lloyd(feat9, c_prev, iters=1)
        helpers.kmeans_step_attribution(feat9, c_prev, traj_c[t], traj_i[t - 1] if t > 0 else None, ids_pre, c_next,
                                        what=f"iteration {t + 1}")
        c_prev = traj_c[t]
    assert np.array_equal(traj_c[-1], key("root_centers"))


def test_empty_cluster_collapses_to_zero():
    """kmeans_quantize.py:209,213-214: an empty cluster's centre becomes ~0 and is never re-seeded."""
    feat = np.random.default_rng(0).random((500, 6)).astype(np.float32)
    cent = np.concatenate([feat[:3], np.full((1, 6), 50.0, np.float32)])   # 4th centre far away: stays empty
    c, ids = ko.lloyd(feat, cent, iters=1)
    assert np.abs(c[3]).max() == 0.0          # 0 / (2e-6) == 0: collapsed onto the origin
    assert np.abs(c[:3]).min() > 0.0


def test_chunk_boundary_extra_trip():
    """N % 10000 == 0 makes the reference loop once more over an empty chunk (:193): only counts change."""
    feat = np.random.default_rng(1).random((10000, 6)).astype(np.float32)
    c1, i1 = ko.lloyd(feat, feat[:8], iters=2, nchunks=2)
    c2, i2 = ko.lloyd(feat, feat[:8], iters=2, nchunks=1)
    assert np.array_equal(i1, i2)
    np.testing.assert_allclose(c1, c2, atol=1e-6)
