"""Generate tests/golden/kmeans_golden.npz by RUNNING THE REFERENCE's Quantize_kMeans on the CPU.

Run in the build container only (``python tests/golden/make_kmeans_golden.py``): /root/reference does not
exist on the GPU box and never travels; only the resulting vectors are committed.  The module is loaded by
file path (scene/__init__.py would pull in plyfile) with ``torch.Tensor.cuda`` shimmed to the identity, as
probed in SURVEY.md Appendix C.

Cases: seeds {0,1,2} x N {1000, 10000, 20001} with (k1,k2) rotating over {(64,10),(32,10),(64,5)}.  Inputs
are regenerated from the seed (``case_inputs``); the file stores a checksum of them plus the reference's
outputs: root centres + ids, leaf centres + ids after two leaf assignments, and the quantised features.
The random centre initialisation is bypassed by presetting ``centers`` / ``leaf_centers`` / ``leaf_cls_ids``
(kmeans_quantize.py:155-160 only initialises when they are empty), so no RNG state is involved.
"""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/scene/kmeans_quantize.py"

CASES = [(seed, N, kk) for seed, kks in zip((0, 1, 2), (((64, 10), (32, 10), (64, 5)),
                                                         ((32, 10), (64, 5), (64, 10)),
                                                         ((64, 5), (64, 10), (32, 10))))
         for N, kk in zip((1000, 10000, 20001), kks)]
NUM_ITERS = 5
POS_WEIGHT = 0.5


def case_inputs(seed, N, k1, k2):
    """Deterministic inputs of one case (shared with the tests)."""
    g = torch.Generator().manual_seed(1000 + seed)
    ins_feat = torch.rand(N, 6, generator=g) * 2 - 1
    xyz = torch.randn(N, 3, generator=g)
    perm = torch.randperm(N, generator=g)
    init_root = perm[:k1]
    init_leaf = torch.randperm(N, generator=g)[:k1 * k2 + 1]
    sub = torch.randint(1, k2 + 1, (k1,), generator=g)
    return ins_feat, xyz, init_root, init_leaf, sub


class _G:
    pass


def run_reference(seed, N, k1, k2):
    torch.Tensor.cuda = lambda self, *a, **k: self
    spec = importlib.util.spec_from_file_location("kq_ref", REF)
    kq = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(kq)
    ins_feat, xyz, init_root, init_leaf, sub = case_inputs(seed, N, k1, k2)
    g = _G()
    g._xyz = xyz.clone()
    g._ins_feat = ins_feat.clone().requires_grad_(True)
    q = kq.Quantize_kMeans(num_clusters=k1, num_leaf_clusters=k2, num_iters=NUM_ITERS, dim=9)
    feat9 = torch.cat((ins_feat, xyz * POS_WEIGHT), dim=1)
    q.centers = feat9[init_root].clone()
    q.forward(g, 1, assign=True, mode="root", pos_weight=POS_WEIGHT)
    out = {"root_centers": q.centers.numpy().copy(), "root_ids": q.nn_index.numpy().astype(np.int16),
           "root_q": g._ins_feat_q.detach().numpy().copy()}
    # the whole Lloyd trajectory, from the reference itself: running it with num_iters = t gives the centres C_t after
    # t iterations and the ids of the re-assignment with C_t -- exactly the assignment iteration t+1 starts from
    # (kmeans_quantize.py:173-240).  Lets the GPU test check ONE iteration at a time from the reference's own state,
    # so that a near-tie flip cannot cascade and every difference is attributable to a row (tests/test_20_kmeans_gpu.py).
    traj_c, traj_i = [], []
    for t in range(1, NUM_ITERS + 1):
        gt = _G()
        gt._xyz = xyz.clone()
        gt._ins_feat = ins_feat.clone().requires_grad_(True)
        qt = kq.Quantize_kMeans(num_clusters=k1, num_leaf_clusters=k2, num_iters=t, dim=9)
        qt.centers = feat9[init_root].clone()
        qt.forward(gt, 1, assign=True, mode="root", pos_weight=POS_WEIGHT)
        traj_c.append(qt.centers.numpy().copy())
        traj_i.append(qt.nn_index.numpy().astype(np.int16))
    assert np.array_equal(traj_c[-1], out["root_centers"]) and np.array_equal(traj_i[-1], out["root_ids"])
    out["root_centers_iter"] = np.stack(traj_c)
    out["root_ids_iter"] = np.stack(traj_i)
    # non-assign iteration: codebook must stay frozen (update_centers discards its result, :58-78)
    before = q.centers.clone()
    q.forward(g, 2, assign=False, mode="root", pos_weight=POS_WEIGHT)
    assert torch.equal(before, q.centers)
    # gradient of the straight-through estimator
    g._ins_feat_q.sum().backward()
    assert torch.equal(g._ins_feat.grad, torch.ones_like(g._ins_feat))
    # leaf level: two coarse clusters
    q.iLeafSubNum = sub.clone()
    q.leaf_centers = ins_feat[init_leaf].clone()
    q.leaf_cls_ids = torch.ones(N, dtype=torch.int64) * k1 * k2
    sel = [int(torch.bincount(q.cls_ids, minlength=k1).argmax()), 1]
    for c in sel:
        q.forward(g, 3, assign=True, mode="leaf", selected_leaf=c)
    out.update({"leaf_sel": np.array(sel), "leaf_centers": q.leaf_centers.numpy().copy(),
                "leaf_ids": q.leaf_cls_ids.numpy().astype(np.int16), "leaf_q": g._ins_feat_q.detach().numpy().copy(),
                "cluster_len_leaf": q.cluster_len.numpy().reshape(-1).astype(np.int32),
                "input_checksum": np.array([float(ins_feat.double().sum()), float(xyz.double().sum())])})
    return out


def main():
    if not os.path.exists(REF):
        sys.exit("reference not mounted: goldens can only be (re)generated in the build container")
    store = {}
    for seed, N, (k1, k2) in CASES:
        r = run_reference(seed, N, k1, k2)
        for k, v in r.items():
            store[f"s{seed}_n{N}_{k}"] = v
        print("case", seed, N, k1, k2, "ok")
    store["cases"] = np.array([(s, n, k[0], k[1]) for s, n, k in CASES], dtype=np.int64)
    path = os.path.join(HERE, "kmeans_golden.npz")
    np.savez_compressed(path, **store)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
