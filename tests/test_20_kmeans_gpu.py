"""GPU parity: HIP k-means (through the drop-in Quantize_kMeans / C ABI) vs the reference-generated goldens
and the NumPy oracle; size-independent properties at the BASELINE size (N = 2M, d = 9, k = 64)."""
import os

import numpy as np
import pytest
import torch

from oracle import kmeans_oracle as ko
from tests.golden.make_kmeans_golden import NUM_ITERS, POS_WEIGHT, case_inputs

pytestmark = pytest.mark.gpu
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


class _G:
    pass


@pytest.mark.parametrize("seed,N,k1,k2", _cases())
def test_quantize_kmeans_matches_reference_golden(gpu_device, seed, N, k1, k2):
    from opengaussian_amd.kmeans import Quantize_kMeans
    gold = np.load(GOLD)
    key = lambda name: gold[f"s{seed}_n{N}_{name}"]
    ins_feat, xyz, init_root, init_leaf, sub = case_inputs(seed, N, k1, k2)
    dev = gpu_device
    g = _G()
    g._xyz = xyz.to(dev)
    g._ins_feat = ins_feat.to(dev).requires_grad_(True)
    q = Quantize_kMeans(num_clusters=k1, num_leaf_clusters=k2, num_iters=NUM_ITERS, dim=9)
    feat9 = torch.cat((ins_feat, xyz * POS_WEIGHT), dim=1)
    q.centers = feat9[init_root].clone().to(dev)
    q.forward(g, 1, assign=True, mode="root", pos_weight=POS_WEIGHT)
    ids = q.nn_index.cpu().numpy()
    ids_ref = key("root_ids").astype(np.int64)
    assert q.nn_index.dtype == torch.int64 and q.cls_ids is q.nn_index
    assert (ids != ids_ref).mean() <= ID_MISMATCH_FRAC
    assert_centers_close(q.centers.cpu().numpy(), key("root_centers"), "root")
    # forward value of the STE == own centres gathered by own ids (bit exact), and == the reference's on rows
    # whose centre row agrees
    assert torch.equal(g._ins_feat_q.detach(), q.centers[q.nn_index][:, :6])
    row_ok = np.abs(q.centers.cpu().numpy() - key("root_centers")).max(axis=1) <= CENTER_TOL
    same = (ids == ids_ref) & row_ok[ids]
    np.testing.assert_allclose(g._ins_feat_q.detach().cpu().numpy()[same], key("root_q")[same], atol=CENTER_TOL, rtol=0)
    # straight-through gradient == ones (kmeans_quantize.py:275)
    g._ins_feat_q.sum().backward()
    assert torch.equal(g._ins_feat.grad, torch.ones_like(g._ins_feat))
    # non-assign call: codebook frozen (:58-78)
    before = q.centers.clone()
    q.forward(g, 2, assign=False, mode="root", pos_weight=POS_WEIGHT)
    assert torch.equal(before, q.centers)
    # leaf level from the reference's coarse ids
    q.cls_ids = torch.from_numpy(ids_ref).to(dev)
    q.iLeafSubNum = sub.clone()
    q.leaf_centers = ins_feat[init_leaf].clone().to(dev)
    q.leaf_cls_ids = torch.ones(N, dtype=torch.int64, device=dev) * k1 * k2
    for c in key("leaf_sel"):
        q.forward(g, 3, assign=True, mode="leaf", selected_leaf=int(c))
    leaf_ref = key("leaf_ids").astype(np.int64)
    assert (q.leaf_cls_ids.cpu().numpy() != leaf_ref).mean() <= ID_MISMATCH_FRAC
    assert_centers_close(q.leaf_centers.cpu().numpy(), key("leaf_centers"), "leaf")
    assert torch.equal(g._ins_feat_q.detach(), q.leaf_centers[q.nn_index][:, :6])
    lids = q.leaf_cls_ids.cpu().numpy()
    row_ok = np.abs(q.leaf_centers.cpu().numpy() - key("leaf_centers")).max(axis=1) <= CENTER_TOL
    same = (lids == leaf_ref) & row_ok[lids]
    np.testing.assert_allclose(g._ins_feat_q.detach().cpu().numpy()[same], key("leaf_q")[same], atol=CENTER_TOL, rtol=0)
    np.testing.assert_array_equal(q.cluster_len.cpu().numpy().reshape(-1)[: k1 * k2 + 1],
                                  np.bincount(q.leaf_cls_ids.cpu().numpy(), minlength=k1 * k2 + 1))


def test_lloyd_matches_oracle_with_inactive_rows(gpu_device):
    """leaf quirk: argmin over the first k_active rows, but all k rows are rewritten (inactive -> ~0)."""
    from opengaussian_amd import kmeans
    g = torch.Generator().manual_seed(3)
    feat = torch.rand(4321, 6, generator=g)
    cent0 = feat[:10].clone()
    cref, iref = ko.lloyd(feat.numpy(), cent0.numpy(), iters=4, nchunks=1, k_active=6, id_offset=30)
    cent = cent0.to(gpu_device).clone()
    ids = kmeans.lloyd(feat.to(gpu_device), cent, iters=4, nchunks=1, k_active=6, id_offset=30)
    assert (ids.cpu().numpy() != iref).mean() <= ID_MISMATCH_FRAC
    np.testing.assert_allclose(cent.cpu().numpy(), cref, atol=CENTER_TOL)
    assert float(cent[6:].abs().max()) == 0.0 and int(ids.min()) >= 30 and int(ids.max()) < 36


def test_edge_sizes(gpu_device):
    from opengaussian_amd import kmeans
    dev = gpu_device
    cent = torch.rand(4, 6, device=dev)
    ids = kmeans.lloyd(torch.zeros(0, 6, device=dev), cent, iters=2, nchunks=1)     # empty subset
    assert ids.shape == (0,) and float(cent.abs().max()) == 0.0                     # 0 / tiny -> 0 (reference too)
    one = torch.tensor([[0.5] * 6], device=dev)
    cent = torch.tensor([[0.0] * 6, [1.0] * 6], device=dev)
    ids = kmeans.lloyd(one, cent, iters=1, nchunks=1)
    assert ids.tolist() == [0]                                                      # tie -> first minimum
    with pytest.raises(RuntimeError):
        kmeans.lloyd(torch.zeros(4, 17, device=dev), torch.zeros(2, 17, device=dev), 1, 1)   # d > 16


def test_full_size_properties(gpu_device):
    """N = 2M (config C4): assignment is a true nearest-centre map, centres are means of their members, and
    re-assigning with the final centres is idempotent."""
    from opengaussian_amd import kmeans
    dev = gpu_device
    g = torch.Generator().manual_seed(0)
    N, d, k = 2_000_000, 9, 64
    feat = torch.cat([torch.rand(N, 6, generator=g), torch.randn(N, 3, generator=g)], dim=1).to(dev)
    cent = feat[torch.randperm(N, generator=g)[:k].to(dev)].clone()
    ids = kmeans.lloyd(feat, cent, iters=5, nchunks=N // 10000 + 1)
    assert torch.equal(ids, kmeans.assign(feat, cent))                              # idempotent
    samp = torch.randint(0, N, (20000,), generator=g).to(dev)
    dist = torch.cdist(feat[samp].double(), cent.double())
    best = dist.min(dim=1)[0]
    chosen = dist.gather(1, ids[samp][:, None])[:, 0]
    assert float((chosen - best).max()) < 1e-5                                      # nearest centre
    # one more Lloyd step from `cent`: new centres == member means (float64 check)
    c2 = cent.clone()
    kmeans.lloyd(feat, c2, iters=1, nchunks=N // 10000 + 1)
    sums = torch.zeros(k, d, dtype=torch.float64, device=dev).index_add_(0, ids, feat.double())
    cnt = torch.bincount(ids, minlength=k).double()
    torch.testing.assert_close(c2.double(), sums / cnt[:, None], atol=1e-4, rtol=1e-4)
    assert int(cnt.sum()) == N


def test_sharded_lloyd_single_rank_equals_fused_lloyd(gpu_device):
    """ogs_kmeans_accumulate + ogs_kmeans_update (the per-iteration pieces of the point-sharded Lloyd) chain to
    what ogs_kmeans_lloyd does in one call; leaf mode (k_active < k, id offset) included."""
    from opengaussian_amd import kmeans as km
    g = torch.Generator().manual_seed(4)
    for (N, d, k, ka, off) in [(30000, 9, 64, 64, 0), (5000, 6, 10, 7, 130)]:
        feat = torch.rand(N, d, generator=g).to(gpu_device)
        init = feat[torch.randperm(N, generator=g)[:k].to(gpu_device)].clone()
        nch = N // 10000 + 1
        c1, c2 = init.clone(), init.clone()
        ids1 = km.lloyd(feat, c1, 5, nch, k_active=ka, id_offset=off)
        ids2 = km.lloyd_sharded(feat, c2, 5, nch, k_active=ka, id_offset=off)
        torch.testing.assert_close(c2, c1, rtol=1e-5, atol=1e-6)
        assert float((ids1 != ids2).float().mean()) < 1e-3
        assert int(ids2.min()) >= off and int(ids2.max()) < off + ka


def _kmeans_rank(rank, world, port, q):
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), OGS_DIST_BACKEND="gloo")
    import torch.distributed as dist
    from opengaussian_amd import dp, kmeans as km
    dp.init_from_env("cuda")
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(8)
    N, d, k = 40001, 9, 64
    feat = torch.rand(N, d, generator=g)
    init = feat[torch.randperm(N, generator=g)[:k]].clone()
    lo, hi = rank * N // world, (rank + 1) * N // world
    c = init.clone().to(dev)
    ids = km.lloyd_sharded(feat[lo:hi].to(dev), c, 5, N // 10000 + 1)
    # single-process truth on the full point set
    c_ref = init.clone().to(dev)
    ids_ref = km.lloyd(feat.to(dev), c_ref, 5, N // 10000 + 1)
    err = float((c - c_ref).abs().max())
    mism = float((ids != ids_ref[lo:hi]).float().mean())
    q.put((rank, err, mism))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_lloyd_two_ranks_match_single_process(gpu_device):
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_kmeans_rank, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, err, mism in res:
        assert err < 1e-4, (rank, err)          # fp32 summation order differs (per-rank partial tables)
        assert mism < 1e-3, (rank, mism)
