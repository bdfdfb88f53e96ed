"""CPU: host-side logic of the drop-in Quantize_kMeans that needs no GPU -- the padded index table of
equalize_cluster_size (kmeans_quantize.py:89-144), attribute surface, and the loud refusal of CPU tensors."""
import pytest
import torch

from opengaussian_amd.kmeans import Quantize_kMeans


def _loop_restatement(nn_index, num_clusters, max_cnt, excl):
    """Per-cluster loop exactly as the reference builds cluster_ids / cluster_len (:119-140)."""
    all_ids, cls_len, excl_ids = [], [], []
    for i in range(num_clusters):
        cur = torch.where(nn_index == i)[0]
        cls_len.append(len(cur))
        if i in excl:
            excl_ids.append(cur[max_cnt:])
            cur = cur[:max_cnt]
        all_ids.append(torch.cat([cur, -1 * torch.ones(max_cnt - len(cur), dtype=torch.long)]))
    return torch.cat(all_ids), torch.tensor(cls_len), excl_ids


@pytest.mark.parametrize("mode,k1,k2,N,th", [("root", 16, 4, 5000, 10000), ("leaf", 8, 3, 3000, 10000),
                                              ("root", 6, 2, 4000, 500)])
def test_equalize_cluster_size_matches_reference_loop(mode, k1, k2, N, th):
    g = torch.Generator().manual_seed(N)
    q = Quantize_kMeans(num_clusters=k1, num_leaf_clusters=k2)
    q.max_cnt_th = th
    num = k1 if mode == "root" else k1 * k2 + 1
    # skewed sizes so that the max_cnt_th exclusion logic triggers in the third case
    probs = torch.rand(num, generator=g) ** 3 + 1e-3
    q.nn_index = torch.multinomial(probs, N, replacement=True, generator=g)
    q.equalize_cluster_size(mode=mode)
    max_cnt = int(q.max_cnt)
    excl = [int(c) for c in q.excl_clusters]
    ids, lens, excl_ids = _loop_restatement(q.nn_index, num, max_cnt, excl)
    assert torch.equal(q.cluster_ids, ids)
    assert torch.equal(q.cluster_len.reshape(-1), lens)
    assert q.cluster_len.shape == (num, 1)
    assert len(q.excl_cluster_ids) == len(excl_ids) == q.n_excl_cls
    for a, b in zip(q.excl_cluster_ids, excl_ids):
        assert torch.equal(a, b)
    if th == 500:
        assert q.n_excl_cls > 0
    assert (q.cls_ids if mode == "root" else q.leaf_cls_ids) is q.nn_index


def test_attribute_surface_and_noop_update():
    q = Quantize_kMeans(num_clusters=32, num_leaf_clusters=10, num_iters=5, dim=9)
    for name in ("num_clusters", "leaf_num_clusters", "num_kmeans_iters", "vec_dim", "leaf_vec_dim", "centers",
                 "leaf_centers", "iLeafSubNum", "cls_ids", "leaf_cls_ids", "nn_index", "cluster_ids", "excl_clusters",
                 "excl_cluster_ids", "cluster_len", "max_cnt", "max_cnt_th", "n_excl_cls", "pos_centers"):
        assert hasattr(q, name), name
    assert (q.num_clusters, q.leaf_num_clusters, q.num_kmeans_iters, q.vec_dim, q.leaf_vec_dim) == (32, 10, 5, 9, 6)
    assert q.update_centers(torch.zeros(4, 9)) is None and len(q.centers) == 0      # frozen codebook (:58-78)
    x, y = torch.rand(7, 6), torch.rand(3, 6)
    torch.testing.assert_close(q.get_dist(x, y), torch.cdist(x, y))


def test_cpu_features_are_refused():
    class G:
        _xyz = torch.rand(10, 3)
        _ins_feat = torch.rand(10, 6, requires_grad=True)
    q = Quantize_kMeans(num_clusters=4, num_leaf_clusters=2, num_iters=1)
    with pytest.raises(RuntimeError, match="no CPU path"):
        q.forward(G(), 1, assign=True, mode="root")


def test_index_table_is_built_lazily_and_matches_the_eager_result():
    """Round 4: equalize_cluster_size sets cls_ids / leaf_cls_ids at once but builds its padded index table (kmeans_quantize.py:
    89-144; read by nothing in the training loop) on the first access of cluster_ids / cluster_len / max_cnt / excl_* -- from the
    nn_index of the most recent call, so a reader sees exactly what the eager version held."""
    g = torch.Generator().manual_seed(5)
    q = Quantize_kMeans(num_clusters=8, num_leaf_clusters=3)
    q.nn_index = torch.randint(0, 8, (2000,), generator=g)
    q.equalize_cluster_size(mode="root")
    assert q.cls_ids is q.nn_index and q._table_mode == "root"           # ids at once, table still pending
    first = q.nn_index
    q.nn_index = torch.randint(0, 8, (2000,), generator=g)                 # a second assign before anyone looked
    q.equalize_cluster_size(mode="root")
    lens = q.cluster_len                                                   # first access builds it, from the LATEST ids
    assert q._table_mode is None
    assert torch.equal(lens.reshape(-1), torch.bincount(q.nn_index, minlength=8))
    assert not torch.equal(lens.reshape(-1), torch.bincount(first, minlength=8))
    ids, _, _ = _loop_restatement(q.nn_index, 8, int(q.max_cnt), [int(c) for c in q.excl_clusters])
    assert torch.equal(q.cluster_ids, ids)
    q.cluster_ids = torch.zeros(3, dtype=torch.long)                       # attributes stay plain assignable
    assert q.cluster_ids.shape == (3,)
