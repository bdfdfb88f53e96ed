"""TEST INFRASTRUCTURE ONLY -- NumPy restatement of OpenGaussian's two-level k-means.

Follows /root/reference/scene/kmeans_quantize.py line by line in behaviour (not in code):
  * ``lloyd``            cluster_assign's chunked Lloyd loop + final re-assignment (:162-240)
  * ``KMeansOracle``     the stateful wrapper: root / leaf modes, centre-slot bookkeeping, the dummy id
                         k1*k2, and the straight-through value of forward() (:146-160,195-211,232-240,252-275)
PINNED: tests/test_oracle_kmeans.py checks it against tests/golden/kmeans_golden.npz, which was produced by
importing and running the reference module itself in the build container
(tests/golden/make_kmeans_golden.py).  Distances are evaluated directly as sum((x-c)^2) in float32; the
reference's torch.cdist goes through a matmul, so ids may differ on rows whose two best distances tie to
within rounding (the tests allow a 1e-3 fraction of such rows).
"""
from __future__ import annotations

import numpy as np

F = np.float32
CHUNK = 10000


def _argmin_sqdist(x, c):
    """First-minimum argmin over centres of squared Euclidean distance, float32."""
    # sequential fp32 accumulation over the feature axis, no FMA: the same operation order as the HIP kernel
    # (kmeans.hip is built with -ffp-contract=off), so both pick identical ids given identical centres
    d = np.zeros((x.shape[0], c.shape[0]), dtype=np.float32)
    for j in range(x.shape[1]):
        t = x[:, j][:, None] - c[:, j][None, :]
        d = d + t * t
    return np.argmin(d, axis=1)


def lloyd(feat, centers, iters, nchunks=None, k_active=None, id_offset=0):
    """Returns (centers [k,d] float32, ids [N] int64).

    counts start at 1e-6 and gain (n_chunk + 1e-6) per chunk (:167,186); after each iteration the sums are
    zeroed and counts are zeroed ONLY where > 0.1 (:213-214), so empty clusters keep a tiny growing count and
    their centre collapses to ~0 (:209).  `nchunks` = number of chunk-loop trips the reference makes
    (root: N // 10000 + 1 because of its `i*chunk > N` exit test, :193; leaf: 1)."""
    feat = np.ascontiguousarray(feat, dtype=np.float32)
    centers = np.array(centers, dtype=np.float32, copy=True)
    N, d = feat.shape
    k = centers.shape[0]
    ka = k if k_active is None else int(k_active)
    if nchunks is None:
        nchunks = N // CHUNK + 1
    bounds = [(i * CHUNK, min((i + 1) * CHUNK, N)) for i in range(nchunks)] if nchunks > 1 else [(0, N)]
    counts = np.zeros(k, dtype=np.float32) + F(1e-6)
    for _ in range(iters):
        sums = np.zeros((k, d), dtype=np.float32)
        for lo, hi in bounds:
            x = feat[lo:hi]
            ids = _argmin_sqdist(x, centers[:ka]) if hi > lo else np.zeros(0, dtype=np.int64)
            onehot = np.zeros((hi - lo, k), dtype=np.float32)
            onehot[np.arange(hi - lo), ids] = 1.0
            sums += onehot.T @ x
            counts = counts + (onehot.sum(axis=0, dtype=np.float32) + F(1e-6))
        centers = (sums / counts[:, None]).astype(np.float32)
        counts = np.where(counts > F(0.1), F(0.0), counts).astype(np.float32)
    out = np.zeros(N, dtype=np.int64)
    for lo, hi in bounds:
        if hi > lo:
            out[lo:hi] = _argmin_sqdist(feat[lo:hi], centers[:ka])
    return centers, out + int(id_offset)


class KMeansOracle:
    """State machine of Quantize_kMeans (attributes named as in the reference, :19-25)."""

    def __init__(self, num_clusters=64, num_leaf_clusters=10, num_iters=10):
        self.k1, self.k2, self.iters = num_clusters, num_leaf_clusters, num_iters
        self.centers = None
        self.leaf_centers = None
        self.iLeafSubNum = None
        self.cls_ids = None
        self.leaf_cls_ids = None
        self.nn_index = None

    def assign_root(self, feat9):
        self.centers, self.nn_index = lloyd(feat9, self.centers, self.iters)
        self.cls_ids = self.nn_index
        return self.nn_index

    def assign_leaf(self, feat6, selected_leaf):
        """Only the points of coarse cluster `selected_leaf` move; ids are offset by c*k2; all k2 rows of the
        slot are rewritten (:196-211,232-240)."""
        N = feat6.shape[0]
        if self.leaf_cls_ids is None:
            self.leaf_cls_ids = np.full(N, self.k1 * self.k2, dtype=np.int64)
        start = selected_leaf * self.k2
        sel = self.cls_ids == selected_leaf
        slot, ids = lloyd(feat6[sel], self.leaf_centers[start:start + self.k2], self.iters, nchunks=1,
                          k_active=int(self.iLeafSubNum[selected_leaf]), id_offset=start)
        self.leaf_centers = self.leaf_centers.copy()
        self.leaf_centers[start:start + self.k2] = slot
        self.leaf_cls_ids = self.leaf_cls_ids.copy()
        self.leaf_cls_ids[sel] = ids
        self.nn_index = self.leaf_cls_ids
        return self.nn_index

    def quantized(self, mode):
        """Forward value of _ins_feat_q: centres[nn_index][:, :6] (:273-275)."""
        c = self.centers if mode == "root" else self.leaf_centers
        return c[self.nn_index][:, :6]
