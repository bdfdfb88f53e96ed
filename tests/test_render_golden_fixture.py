"""CPU: the render() fixtures (tests/golden/render_golden.npz, produced by running the reference's own
gaussian_renderer.render() over the CPU oracle -- tests/golden/make_render_golden.py) still describe the inputs that
tests/golden/render_cases.py rebuilds from its seeds, and cover every branch of render() the GPU test relies on."""
import os

import numpy as np
import torch

from tests.golden import render_cases as rc

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "render_golden.npz")


def test_fixture_inputs_are_reproducible_from_the_seeds():
    gold = np.load(GOLD)
    for name in rc.CASES:
        case = rc.build(name)
        chk = np.array([float(case["params"][n].double().sum()) for n in sorted(case["params"])])
        assert np.array_equal(chk, gold[f"{name}/inputs_checksum"]), name
        torch.manual_seed(case["rng_seed"])
        coin = float(torch.rand(1))
        assert (coin > 0.5 and case["kwargs"].get("rescale", True)) == case["expect_rescale"], name


def test_fixture_covers_every_render_branch():
    gold = np.load(GOLD)
    f = set(gold.files)
    # stage-1 call with and without the rescale draw; gradients of all seven parameter tensors + viewspace_points
    for name in ("stage1_norescale", "stage1_rescale", "stage22_selected_root", "rgb_only_cov3d_python"):
        for n in rc.PARAM_NAMES + ("viewspace_points",):
            assert f"{name}/grad/{n}" in f
    assert gold["stage1_norescale/ins_feat"].shape[0] == 6 and gold["stage1_rescale/silhouette"].shape[0] == 1
    # coarse cluster block: selected root; better_vis over every cluster with one excluded by bClusterOccur; nothing selected
    assert gold["stage22_selected_root/cluster_occur"].tolist() == [False, False, True, False]
    assert int(gold["stage22_selected_root/cluster_imgs/len"]) == 1 and gold["stage22_selected_root/cluster_imgs/0"].shape[0] == 6
    assert gold["better_vis_all_clusters/cluster_occur"].tolist() == [True, False, True, True]
    assert int(gold["better_vis_all_clusters/cluster_imgs/len"]) == 3
    assert gold["cluster_none_selected/cluster_occur"].tolist() == [False, False, False]
    assert "cluster_none_selected/cluster_silhouettes/emptylist" in f and int(gold["cluster_none_selected/cluster_imgs/len"]) == 0
    # leaf block: range of one root, root the camera never saw, union of selected leaves with pre-mask / seg_rgb / kNN filter
    assert gold["stage22_selected_root/occured_leaf_id"].tolist() == [6, 7, 8]
    assert gold["lang_leaves_of_root/occured_leaf_id"].tolist() == [3, 4, 5]
    assert gold["lang_leaves_root_unseen/occured_leaf_id"].tolist() == []
    assert "lang_leaves_root_unseen/leaf_cluster_silhouettes/emptylist" in f
    assert gold["selected_leaf_union_seg_rgb/occured_leaf_id"].tolist() == [4]
    assert gold["selected_leaf_union_seg_rgb/leaf_clusters_imgs/0"].shape[0] == 6      # the RGB image stacked twice (:336-346)
    # feature map switched off
    assert "rgb_only_cov3d_python/ins_feat/none" in f and "rgb_only_cov3d_python/silhouette/none" in f
