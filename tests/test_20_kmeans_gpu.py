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
from tests import helpers

CENTER_TOL = helpers.KM_CENTER_TOL      # 1e-4 (north_star); no fraction of rows is exempt


def assert_centers_close(got, want, k_note=""):
    """EVERY centre row within 1e-4 of the reference's (leaf level: small subsets, no flips on the goldens)."""
    diff = np.abs(np.asarray(got, np.float64) - np.asarray(want, np.float64)).max(axis=1)
    assert diff.max() <= CENTER_TOL, f"{int((diff > CENTER_TOL).sum())} centre rows differ by > {CENTER_TOL} {k_note}: {diff.max()}"


def attribute_root_run(feat9, init_centers, traj_c, traj_i, final_centers, final_ids, device):
    """End-to-end root assignment (5 Lloyd iterations + re-assignment) against the reference's result.  If every centre
    row is within 1e-4 the ids are attributed directly.  Otherwise a near-tie row flipped in some iteration and
    the difference CASCADED: the GPU's own trajectory is replayed one iteration at a time and every step is
    attributed against the reference's trajectory (helpers.kmeans_step_attribution with the two sides' previous
    centres), and the fused 5-iteration call must land on the replayed trajectory."""
    from opengaussian_amd import kmeans
    N = feat9.shape[0]
    diff = np.abs(final_centers.astype(np.float64) - traj_c[-1]).max()
    if diff <= CENTER_TOL:
        return helpers.kmeans_final_ids_attribution(feat9.numpy(), traj_c[-1], traj_i[-1], final_centers, final_ids, "root")
    fdev = feat9.to(device)
    c_ref, c_got = init_centers.copy(), init_centers.copy()
    flips = 0
    for t in range(NUM_ITERS):
        cdev = torch.from_numpy(c_got).to(device).contiguous()
        ids_pre = kmeans.assign(fdev, cdev).cpu().numpy()
        kmeans.lloyd(fdev, cdev, iters=1, nchunks=N // 10000 + 1)
        n, _ = helpers.kmeans_step_attribution(feat9.numpy(), c_ref, traj_c[t], traj_i[t - 1] if t > 0 else None, ids_pre,
                                               cdev.cpu().numpy(), what=f"own trajectory, iteration {t + 1}", c_prev_got=c_got)
        flips += n
        c_ref, c_got = traj_c[t], cdev.cpu().numpy()
    assert flips > 0, "centres differ by more than 1e-4 although no row flipped"
    np.testing.assert_allclose(final_centers, c_got, atol=2e-6, rtol=0)      # fused call == replayed single iterations
    return flips


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
    # centres 1e-4 and ids exact, except what named near-tie rows (float64 gap < 1e-5) explain, step by step
    attribute_root_run(feat9, feat9[init_root].numpy().copy(), key("root_centers_iter"),
                       key("root_ids_iter").astype(np.int64), q.centers.cpu().numpy(), ids, dev)
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
    assert_centers_close(q.leaf_centers.cpu().numpy(), key("leaf_centers"), "leaf")
    touched = np.isin(ids_ref, key("leaf_sel"))
    assert np.array_equal(q.leaf_cls_ids.cpu().numpy()[~touched], leaf_ref[~touched])       # dummy id k1*k2 kept
    helpers.kmeans_final_ids_attribution(ins_feat.numpy()[touched], key("leaf_centers"), leaf_ref[touched],
                                         q.leaf_centers.cpu().numpy(), q.leaf_cls_ids.cpu().numpy()[touched], "leaf")
    assert torch.equal(g._ins_feat_q.detach(), q.leaf_centers[q.nn_index][:, :6])
    lids = q.leaf_cls_ids.cpu().numpy()
    row_ok = np.abs(q.leaf_centers.cpu().numpy() - key("leaf_centers")).max(axis=1) <= CENTER_TOL
    same = (lids == leaf_ref) & row_ok[lids]
    np.testing.assert_allclose(g._ins_feat_q.detach().cpu().numpy()[same], key("leaf_q")[same], atol=CENTER_TOL, rtol=0)
    np.testing.assert_array_equal(q.cluster_len.cpu().numpy().reshape(-1)[: k1 * k2 + 1],
                                  np.bincount(q.leaf_cls_ids.cpu().numpy(), minlength=k1 * k2 + 1))
    if np.array_equal(lids, leaf_ref):          # the reference's own equalize_cluster_size bookkeeping (:130,138)
        np.testing.assert_array_equal(q.cluster_len.cpu().numpy().reshape(-1), key("cluster_len_leaf"))


@pytest.mark.parametrize("seed,N,k1,k2", _cases())
def test_lloyd_follows_reference_trajectory_step_by_step(gpu_device, seed, N, k1, k2):
    """Each of the reference's five Lloyd iterations (root_centers_iter / root_ids_iter: the reference run with
    num_iters = 1..5, tests/golden/make_kmeans_golden.py) re-done on the GPU FROM THE REFERENCE'S centres, so a
    near-tie flip cannot cascade: every id that differs from the float64 nearest centre sits on a tie < 1e-5, every
    centre is the mean of its own members, and a centre may differ from the reference's by more than 1e-4 only by
    what the named near-tie rows touching it can move (tests/helpers.py::kmeans_step_attribution)."""
    from opengaussian_amd import kmeans
    gold = np.load(GOLD)
    key = lambda name: gold[f"s{seed}_n{N}_{name}"]
    ins_feat, xyz, init_root, _, _ = case_inputs(seed, N, k1, k2)
    feat9 = torch.cat((ins_feat, xyz * POS_WEIGHT), dim=1)
    fdev = feat9.to(gpu_device)
    traj_c, traj_i = key("root_centers_iter"), key("root_ids_iter").astype(np.int64)
    c_prev = feat9[init_root].numpy().copy()
    for t in range(NUM_ITERS):
        cdev = torch.from_numpy(c_prev).to(gpu_device).contiguous()
        ids_pre = kmeans.assign(fdev, cdev).cpu().numpy()                       # the assignment the iteration uses
        kmeans.lloyd(fdev, cdev, iters=1, nchunks=N // 10000 + 1)               # cdev <- centres after ONE iteration
        helpers.kmeans_step_attribution(feat9.numpy(), c_prev, traj_c[t], traj_i[t - 1] if t > 0 else None, ids_pre,
                                        cdev.cpu().numpy(), what=f"iteration {t + 1}")
        c_prev = traj_c[t]


def test_lloyd_matches_oracle_with_inactive_rows(gpu_device):
    """leaf quirk: argmin over the first k_active rows, but all k rows are rewritten (inactive -> ~0)."""
    from opengaussian_amd import kmeans
    g = torch.Generator().manual_seed(3)
    feat = torch.rand(4321, 6, generator=g)
    cent0 = feat[:10].clone()
    cref, iref = ko.lloyd(feat.numpy(), cent0.numpy(), iters=4, nchunks=1, k_active=6, id_offset=30)
    cent = cent0.to(gpu_device).clone()
    ids = kmeans.lloyd(feat.to(gpu_device), cent, iters=4, nchunks=1, k_active=6, id_offset=30)
    np.testing.assert_array_equal(ids.cpu().numpy(), iref)        # same fp32 operation order as the oracle: exact
    np.testing.assert_allclose(cent.cpu().numpy(), cref, atol=1e-6)
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
        # different fp32 summation order of the centre update: ids may differ only on float64 near-ties
        helpers.kmeans_final_ids_attribution(feat.cpu().numpy(), c1[:ka].cpu().numpy(), ids1.cpu().numpy() - off,
                                             c2[:ka].cpu().numpy(), ids2.cpu().numpy() - off, "sharded vs fused")
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
    from tests import helpers as h
    mism = h.kmeans_final_ids_attribution(feat[lo:hi].numpy(), c_ref.cpu().numpy(), ids_ref[lo:hi].cpu().numpy(),
                                          c.cpu().numpy(), ids.cpu().numpy(), f"rank {rank}")   # raises unless near-ties
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
        assert err < 1e-5, (rank, err)          # fp32 summation order differs (per-rank partial tables)
        assert mism <= 2, (rank, mism)          # every differing row was attributed to a float64 near-tie in the child
