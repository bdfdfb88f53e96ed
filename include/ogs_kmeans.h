/*
 * ogs_kmeans.h -- C ABI of the MI355X-native Lloyd k-means used by OpenGaussian's two-level codebook.
 *
 * Replaces the PyTorch ops inside /root/reference/scene/kmeans_quantize.py:
 *   Quantize_kMeans.cluster_assign  (:146-241)  cdist -> argmin -> one_hot -> mask^T @ feat per 10k chunk
 *   Quantize_kMeans.get_dist        (:38-55)    pairwise Euclidean distance
 *   Quantize_kMeans.update_centers_ (:82-87)    mask^T @ feat
 *   the gather of Quantize_kMeans.forward (:273)
 * with: one fused pairwise-L2 + argmin + segmented-reduce kernel per Lloyd iteration (features read once,
 * centres in LDS, per-workgroup LDS accumulators, fixed-order cross-workgroup reduction) and a gather.
 * The Python class opengaussian_amd.kmeans.Quantize_kMeans keeps the reference's attributes and call
 * signature and binds these functions with ctypes.
 *
 * All pointers are DEVICE pointers; feat/centers are contiguous fp32, ids are int64 (torch.long, as the
 * reference's nn_index / cls_ids).  `stream` is a hipStream_t as void*.  Returns 0 or a negative
 * OGS_ERR_* code (see ogs_raster.h); ogs_last_error() describes it.
 */
#ifndef OGS_KMEANS_H
#define OGS_KMEANS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OGS_KMEANS_MAX_DIM 16
#define OGS_KMEANS_MAX_ACC 16384   /* k * (d + 1) accumulator floats per workgroup (64 KiB of LDS) */

size_t ogs_kmeans_tmp_bytes(int64_t N, int32_t d, int32_t k);

/* `iters` Lloyd iterations followed by the final re-assignment (kmeans_quantize.py:173-240).
 *   feat      [N,d]   points
 *   centers   [k,d]   in: initial centres, out: final centres
 *   k_active  argmin considers only the first k_active rows (leaf mode: iLeafSubNum[c], :172,200);
 *             all k rows are (re)written every iteration (empty rows collapse to ~0, :211)
 *   nchunks   number of 10k chunks the reference would have looped over (root: N/10000+1, leaf: 1):
 *             reproduces its `counts += n + 1e-6` per chunk and the counts>0.1 reset (:186,213-214)
 *   ids_out   [N] int64, final argmin + id_offset
 */
int ogs_kmeans_lloyd(const float* feat, int64_t N, int32_t d, float* centers, int32_t k, int32_t k_active,
                     int32_t iters, int32_t nchunks, int64_t* ids_out, int64_t id_offset, void* tmp, void* stream);

/* Sharded Lloyd (new capability, SURVEY.md section 8(e): "k-means shards by point range"): each rank holds a
 * contiguous range of the points and the (replicated) centres.  One iteration =
 *     ogs_kmeans_accumulate(my points)        -> table [k, d+1] = per-centre feature sums | point counts
 *     all-reduce(table, SUM) over the ranks      (k*(d+1) floats: 640 at k = 64, d = 9)
 *     ogs_kmeans_update(table, ...)            -> centres, with the reference's count rule (see nchunks above)
 * `tmp` as for ogs_kmeans_lloyd (ogs_kmeans_tmp_bytes(N, d, k)).  `counts_state` [k] is the reference's
 * persistent `counts` vector: set every entry to 1e-6 before the first iteration (:167) and pass it unchanged
 * afterwards; `nchunks` is computed from the GLOBAL point count. */
int ogs_kmeans_accumulate(const float* feat, int64_t N, int32_t d, const float* centers, int32_t k, int32_t k_active,
                          float* table, void* tmp, void* stream);
int ogs_kmeans_update(const float* table, int32_t k, int32_t d, int32_t nchunks, float* counts_state, float* centers,
                      void* stream);

/* ids_out[i] = argmin_j<k ||feat[i] - centers[j]||^2 (first minimum wins, as torch.argmin) + id_offset */
int ogs_kmeans_assign(const float* feat, int64_t N, int32_t d, const float* centers, int32_t k, int64_t* ids_out,
                      int64_t id_offset, void* stream);

/* out[i, 0:out_dim] = centers[ids[i], 0:out_dim]   (forward value of the straight-through estimator,
 * kmeans_quantize.py:273-275: x - x.detach() + centres == centres) */
int ogs_kmeans_gather(const float* centers, const int64_t* ids, int64_t N, int32_t vec_dim, int32_t out_dim,
                      float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* OGS_KMEANS_H */
