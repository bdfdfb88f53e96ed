"""MI355X-native ``Quantize_kMeans`` -- drop-in for /root/reference/scene/kmeans_quantize.py:12-280.

Same constructor, attributes (``centers, leaf_centers, iLeafSubNum, cls_ids, leaf_cls_ids, nn_index,
cluster_ids, cluster_len, max_cnt, excl_clusters ...``) and ``forward(gaussian, iteration, assign, mode,
selected_leaf, pos_weight)`` contract that train.py:190-215,299,310,330-332,355,586-588,626 and save_kmeans
(train.py:62-100) read and write directly.  The chunked cdist -> argmin -> one_hot -> mask^T @ feat loop of
``cluster_assign`` runs as HIP kernels behind the C ABI of include/ogs_kmeans.h (one fused pass over the
features per Lloyd iteration); torch is only used for device memory, the centre initialisation
(``feat[randperm]``, identical RNG call to the reference) and the tiny index bookkeeping.  No CPU path.
"""
from __future__ import annotations

import torch

from . import _lib
from ._lib import check, ptr


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _need_gpu(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise RuntimeError(f"{what} must live on the GPU (got {t.device}); the MI355X k-means has no CPU path")


def lloyd(feat: torch.Tensor, centers: torch.Tensor, iters: int, nchunks: int, k_active: int | None = None,
          id_offset: int = 0):
    """Run `iters` Lloyd iterations + the final re-assignment on the GPU.  ``centers`` [k,d] is updated in
    place; returns int64 ids [N] (+ id_offset).  Semantics: scene/kmeans_quantize.py:162-240."""
    _need_gpu(feat, "feat")
    lib = _lib.lib()
    f = feat.detach().to(torch.float32).contiguous()
    if not (centers.is_cuda and centers.dtype == torch.float32 and centers.is_contiguous()):
        raise RuntimeError("centers must be a contiguous fp32 GPU tensor (updated in place)")
    N, d = int(f.shape[0]), int(f.shape[1])
    k = int(centers.shape[0])
    ids = torch.empty(N, dtype=torch.int64, device=f.device)
    tmp = torch.empty(int(lib.ogs_kmeans_tmp_bytes(N, d, k)), dtype=torch.uint8, device=f.device)
    check(lib.ogs_kmeans_lloyd(ptr(f) if N else None, N, d, ptr(centers), k, int(k if k_active is None else k_active),
                               int(iters), int(nchunks), ptr(ids) if N else None, int(id_offset), ptr(tmp), _stream()),
          "ogs_kmeans_lloyd")
    return ids


def lloyd_sharded(feat: torch.Tensor, centers: torch.Tensor, iters: int, nchunks: int, k_active: int | None = None,
                  id_offset: int = 0, group=None):
    """Lloyd iterations with the points sharded by range over the ranks of `group`:
    `feat` holds THIS rank's rows, `centers` [k,d] is replicated and updated in place identically on every rank,
    `nchunks` is computed from the GLOBAL point count.  Per iteration one HIP pass over the local rows, one
    all-reduce of the [k, d+1] sums|counts table (640 floats at k=64, d=9) and the centre update with the
    reference's count rule.  Returns this rank's int64 ids.  With one rank it equals lloyd() up to the order of
    the fp32 summation."""
    import torch.distributed as dist
    _need_gpu(feat, "feat")
    lib = _lib.lib()
    f = feat.detach().to(torch.float32).contiguous()
    if not (centers.is_cuda and centers.dtype == torch.float32 and centers.is_contiguous()):
        raise RuntimeError("centers must be a contiguous fp32 GPU tensor (updated in place)")
    N, d = int(f.shape[0]), int(f.shape[1])
    k = int(centers.shape[0])
    ka = int(k if k_active is None else k_active)
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    table = torch.empty(k, d + 1, dtype=torch.float32, device=f.device)
    counts = torch.full((k,), 1e-6, dtype=torch.float32, device=f.device)
    tmp = torch.empty(int(lib.ogs_kmeans_tmp_bytes(N, d, k)), dtype=torch.uint8, device=f.device)
    for _ in range(int(iters)):
        check(lib.ogs_kmeans_accumulate(ptr(f) if N else None, N, d, ptr(centers), k, ka, ptr(table), ptr(tmp),
                                        _stream()), "ogs_kmeans_accumulate")
        if multi:
            dist.all_reduce(table, op=dist.ReduceOp.SUM, group=group)
        check(lib.ogs_kmeans_update(ptr(table), k, d, int(nchunks), ptr(counts), ptr(centers), _stream()),
              "ogs_kmeans_update")
    return assign(f, centers[:ka], id_offset) if N else torch.empty(0, dtype=torch.int64, device=f.device)


def assign(feat: torch.Tensor, centers: torch.Tensor, id_offset: int = 0) -> torch.Tensor:
    """argmin_j ||feat_i - centers_j|| (first minimum wins) as int64 [N]."""
    _need_gpu(feat, "feat")
    lib = _lib.lib()
    f = feat.detach().to(torch.float32).contiguous()
    c = centers.detach().to(torch.float32).contiguous()
    N, d = int(f.shape[0]), int(f.shape[1])
    ids = torch.empty(N, dtype=torch.int64, device=f.device)
    check(lib.ogs_kmeans_assign(ptr(f), N, d, ptr(c), int(c.shape[0]), ptr(ids), int(id_offset), _stream()),
          "ogs_kmeans_assign")
    return ids


class _GatherSTE(torch.autograd.Function):
    """value = centres[ids][:, :out_dim]; gradient goes straight through to the instance features
    (kmeans_quantize.py:273-275: ins_feat - ins_feat.detach() + sampled_centers[:, :6])."""

    @staticmethod
    def forward(ctx, ins_feat, centers, ids, out_dim):
        lib = _lib.lib()
        c = centers.detach().to(torch.float32).contiguous()
        N = int(ids.shape[0])
        out = torch.empty(N, out_dim, dtype=torch.float32, device=ins_feat.device)
        check(lib.ogs_kmeans_gather(ptr(c), ptr(ids.contiguous()), N, int(c.shape[1]), int(out_dim), ptr(out),
                                    _stream()), "ogs_kmeans_gather")
        return out

    @staticmethod
    def backward(ctx, g):
        return g, None, None, None


class Quantize_kMeans():
    def __init__(self, num_clusters=64, num_leaf_clusters=10, num_iters=10, dim=9, dim_leaf=6):
        self.num_clusters = num_clusters            # k1
        self.leaf_num_clusters = num_leaf_clusters  # k2
        self.num_kmeans_iters = num_iters
        self.vec_dim = dim                          # coarse level: 6 feat + 3 xyz
        self.leaf_vec_dim = dim_leaf                # fine level: 6 feat
        self.centers = torch.empty(0)               # [k1, 9]
        self.leaf_centers = torch.empty(0)          # [k1*k2+1, 6]
        self.iLeafSubNum = torch.empty(0)           # fine clusters per coarse cluster
        self.cls_ids = torch.empty(0)               # [N] coarse id
        self.leaf_cls_ids = torch.empty(0)          # [N] fine id (k1*k2 = "unassigned")
        self.nn_index = torch.empty(0)              # [N]
        # bookkeeping of equalize_cluster_size (consumed only by update_centers, which is a no-op on state): built lazily,
        # see equalize_cluster_size; the public names are properties below
        self._table_mode = None
        self._cluster_ids = torch.empty(0)
        self._excl_clusters = []
        self._excl_cluster_ids = []
        self._cluster_len = torch.empty(0)
        self._max_cnt = 0
        self.max_cnt_th = 10000
        self._n_excl_cls = 0
        self.pos_centers = torch.empty(0)

    def _lazy(name):                                   # noqa: N805 -- class-body helper, not a method
        priv = "_" + name

        def get(self):
            self._build_table()
            return getattr(self, priv)

        def set_(self, value):
            self._build_table()                        # a pending rebuild must not overwrite what the caller assigns
            setattr(self, priv, value)
        return property(get, set_)
    cluster_ids = _lazy("cluster_ids")
    excl_clusters = _lazy("excl_clusters")
    excl_cluster_ids = _lazy("excl_cluster_ids")
    cluster_len = _lazy("cluster_len")
    max_cnt = _lazy("max_cnt")
    n_excl_cls = _lazy("n_excl_cls")
    del _lazy

    # ---- reference helpers kept for API parity ----------------------------------------------------------
    def get_dist(self, x, y, mode='sq_euclidean'):
        """Euclidean distance matrix (kmeans_quantize.py:38-55; despite the name it is not squared)."""
        return torch.cdist(x.unsqueeze(0).detach(), y.unsqueeze(0).detach())[0]

    def update_centers(self, feat, mode="root", selected_leaf=-1):
        """Non-assign iterations: the reference computes new centres into a LOCAL and discards them
        (kmeans_quantize.py:58-78), so the codebook is frozen between assigns.  Nothing to do."""
        return None

    def equalize_cluster_size(self, mode="root"):
        """Padded per-cluster index table + lengths (kmeans_quantize.py:89-144).

        What the training loop reads after an assign is ``cls_ids`` / ``leaf_cls_ids`` (set here, at once).  The table itself --
        ``cluster_ids, cluster_len, max_cnt, excl_clusters, excl_cluster_ids, n_excl_cls`` -- is consumed only by
        ``update_centers``, which is a no-op on state in the reference (:58-78), and by nothing in train.py / render(); rebuilding
        it over ALL N points on every leaf assign (the reference does) was 90 % of a leaf assign here (bench.py `kmeans.leaf`:
        2.0 ms of 2.2).  So the table is built LAZILY, on the first read of any of those attributes, from the ``nn_index`` of the
        most recent call -- the same values a reader of the eager version would see."""
        if mode == "root":
            self.cls_ids = self.nn_index
        elif mode == "leaf":
            self.leaf_cls_ids = self.nn_index
        self._table_mode = mode

    def _build_table(self):
        """The eager body of equalize_cluster_size, with one stable sort instead of the reference's Python loop over clusters."""
        mode, self._table_mode = self._table_mode, None
        if mode is None:
            return
        nn = self.nn_index
        dev = nn.device
        num_clusters = self.num_clusters if mode == "root" else self.num_clusters * self.leaf_num_clusters + 1
        # Oversized clusters are capped (kmeans_quantize.py:99-117): among the (at most) 100 most populated
        # clusters, the leading run whose size exceeds max_cnt_th is "excluded" -- its surplus members go to
        # excl_cluster_ids -- and the table width is the size of the first cluster that is not.
        unq, n_unq = torch.unique(nn, return_counts=True)
        top_cnt, top_idx = torch.topk(n_unq, min(100, n_unq.numel()))        # descending; same tie order as the reference
        n_excl = int((top_cnt > self.max_cnt_th).sum())
        self._excl_clusters = sorted(unq[top_idx[:n_excl]])
        self._excl_cluster_ids = []
        self._n_excl_cls = n_excl
        self._max_cnt = top_cnt[min(n_excl, top_cnt.numel() - 1)]
        max_cnt = int(self._max_cnt)
        counts = torch.bincount(nn, minlength=num_clusters)[:num_clusters]
        order = torch.argsort(nn, stable=True)
        sorted_ids = nn[order]
        starts = torch.cumsum(counts, 0) - counts
        rank = torch.arange(nn.numel(), device=dev) - starts[sorted_ids]
        table = torch.full((num_clusters * max_cnt,), -1, dtype=torch.long, device=dev)
        keep = rank < max_cnt
        table[sorted_ids[keep] * max_cnt + rank[keep]] = order[keep]
        for c in self._excl_clusters:
            sel = (sorted_ids == c) & ~keep
            self._excl_cluster_ids.append(order[sel])
        self._cluster_ids = table
        self._cluster_len = counts.to(torch.long).unsqueeze(1)

    # ---- the hot path -----------------------------------------------------------------------------------------
    def cluster_assign(self, feat, feat_scaled=None, mode="root", selected_leaf=-1):
        feat = feat.detach()
        _need_gpu(feat, "feat")
        N = feat.shape[0]
        if len(self.centers) == 0 and mode == "root":
            self.centers = feat[torch.randperm(N)[:self.num_clusters].to(feat.device), :]
        if len(self.leaf_centers) == 0 and mode == "leaf":
            n_leaf = self.num_clusters * self.leaf_num_clusters + 1
            self.leaf_centers = feat[torch.randperm(N)[:n_leaf].to(feat.device), :]
            self.leaf_cls_ids = torch.ones(N, dtype=torch.int64, device=feat.device) * (n_leaf - 1)

        chunk = 10000
        if mode == "root":
            centers = self.centers.detach().to(torch.float32).contiguous().clone()
            self.nn_index = lloyd(feat, centers, self.num_kmeans_iters, nchunks=N // chunk + 1)
            self.centers = centers
        elif mode == "leaf":
            k2 = self.leaf_num_clusters
            start_id = int(selected_leaf) * k2
            k_active = max(1, min(int(self.iLeafSubNum[selected_leaf]), k2))
            selected_pts = self.cls_ids == selected_leaf
            sub = feat[selected_pts]
            if not (self.leaf_centers.dtype == torch.float32 and self.leaf_centers.is_contiguous()):
                self.leaf_centers = self.leaf_centers.to(torch.float32).contiguous()
            slot = self.leaf_centers[start_id:start_id + k2].clone()
            ids = lloyd(sub, slot, self.num_kmeans_iters, nchunks=1, k_active=k_active, id_offset=start_id)
            self.leaf_centers[start_id:start_id + k2] = slot
            self.leaf_cls_ids[selected_pts] = ids
            self.nn_index = self.leaf_cls_ids
        self.equalize_cluster_size(mode=mode)

    def forward(self, gaussian, iteration, assign=False, mode="root", selected_leaf=-1, pos_weight=1.0):
        if mode == "root":
            xyz_feat = gaussian._xyz.detach() * pos_weight
            feat = torch.cat((gaussian._ins_feat, xyz_feat), dim=1)    # [N, 9]
        elif mode == "leaf":
            feat = gaussian._ins_feat
        if assign:
            self.cluster_assign(feat, mode=mode, selected_leaf=selected_leaf)
        else:
            self.update_centers(feat, mode=mode, selected_leaf=selected_leaf)
        centers = self.centers if mode == "root" else self.leaf_centers
        gaussian._ins_feat_q = _GatherSTE.apply(gaussian._ins_feat, centers, self.nn_index, 6)
