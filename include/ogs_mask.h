/*
 * ogs_mask.h -- C ABI of the image-space mask reductions of OpenGaussian's stage-1 / association losses
 * (SURVEY.md section 8 f3).
 *
 * Replaces the [num_mask, C, H, W] expansions inside
 *   /root/reference/utils/opengs_utlis.py:240-283   mask_feature_mean (+ its chunk helpers :195-238)
 *   /root/reference/train.py:102-122                cohesion_loss
 * with segmented reductions: the feature map [C,H,W] and the mask stack [N,H,W] are each read ONCE per call,
 * every wave skips the masks that have no pixel in its 256-pixel strip (one ballot per mask), and the
 * per-mask partial sums leave a wave as one contiguous float-atomic segment.  Nothing of size N*C*H*W is
 * ever materialised.  Masks may overlap (the reference's semantics are per mask).
 *
 * All pointers are DEVICE pointers; feat / weight / tables are contiguous fp32, masks are bytes (0 = outside,
 * anything else = inside).  HW = H*W.  C must be 3 or 6.  `stream` is a hipStream_t as void*.  Returns 0 or a
 * negative OGS_ERR_* code (ogs_raster.h); ogs_last_error() describes it.
 */
#ifndef OGS_MASK_H
#define OGS_MASK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Every per-mask output table below ([N, ...]) has rows OGS_MASK_TABLE_STRIDE floats apart (row n starts at
 * table + n * 16): one 64-byte line per mask, because float atomics from many waves serialise per cache line
 * (packed rows -- 8 masks per line for the 2-column cohesion table -- ran 2.5x slower). */
#define OGS_MASK_TABLE_STRIDE 16

/* table[n, 0:C]   = sum_pix mask[n,pix] * w[pix] * feat[:,pix]
 * table[n, C]     = sum_pix mask[n,pix] * w[pix]                      (w = 1 when weight == NULL)
 * with_squares != 0 (C == 6 or 3): additionally table[n, C+1 : 2C+1] = sum mask*w*feat^2
 * (variance path of mask_feature_mean(return_var=True), opengs_utlis.py:272-283).
 * The table is zeroed by the call.  mask_feature_mean = table[:, :C] / clamp(table[:, C], min=1). */
int ogs_mask_feature_sums(const float* feat, const uint8_t* masks, const float* weight, int32_t C, int32_t N,
                          int64_t HW, int32_t with_squares, float* table, void* stream);

/* Backward of the sums table, given g = dL/dtable ([N, C+1], host side: coef = g[:, :C] contiguous [N,C],
 * coef_cnt = g[:, C] contiguous [N]):
 *   dfeat[c,pix]  = w[pix] * sum_n mask[n,pix] * coef[n,c]                                  (always written)
 *   dweight[pix]  = sum_n mask[n,pix] * (sum_c coef[n,c] * feat[c,pix] + coef_cnt[n])       (when dweight != NULL;
 *                   the silhouette passed as image_mask is a rasterizer output, i.e. part of the autograd graph) */
int ogs_mask_feature_sums_backward(const uint8_t* masks, const float* weight, const float* coef, const float* feat,
                                   const float* coef_cnt, int32_t C, int32_t N, int64_t HW, float* dfeat,
                                   float* dweight, void* stream);

/* cohesion_loss pieces: table[n,0] = sum_pix mask[n,pix] * ||feat[:,pix] - mean[n,:]||_2, table[n,1] = pixel count
 * of mask n.  loss = mean_n(table[n,0] / clamp(table[n,1], 1)).  The [N,2] table is zeroed by the call. */
int ogs_mask_cohesion(const float* feat, const uint8_t* masks, const float* mean, int32_t C, int32_t N, int64_t HW,
                      float* table, void* stream);

/* Backward of cohesion_loss with gl[n] = dL/dloss_n / clamp(count_n, 1)  ([N], host side):
 *   dfeat[c,pix] = sum_n mask * gl[n] * (feat - mean[n]) / dist      (0 where dist == 0, as torch's norm)
 *   dmean[n,c]   = - sum_pix mask * gl[n] * (feat - mean[n]) / dist
 * Writes every element of dfeat [C,H,W]; dmean (rows OGS_MASK_TABLE_STRIDE apart) is zeroed by the call. */
int ogs_mask_cohesion_backward(const float* feat, const uint8_t* masks, const float* mean, const float* gl, int32_t C,
                               int32_t N, int64_t HW, float* dfeat, float* dmean, void* stream);

/* separation_loss (train.py:124-155), forward AND gradient in two small launches: means [N, C] (2 <= N <= 1024 masks,
 * C <= 16), late != 0 for iteration > 35 000 (weights below 0.9 become 0.1).  loss[0] = sum_ij inv_ij * w_ij / (N (N-1))
 * with inv_ij = 1 / (|m_i - m_j|^2 + 1), 0 on the diagonal, and w_ij = rank of inv_ij inside row i (ties by column, i.e.
 * a stable argsort().argsort()) / (N-1) * 0.9 + 0.1.  grad ([N, C], may be NULL) = dloss/dmeans; the rank weights carry no
 * gradient, as under autograd.  tmp: (N * N + N) floats of device scratch. */
int ogs_separation_loss(const float* means, int32_t N, int32_t C, int32_t late, float* loss, float* grad, float* tmp,
                        void* stream);

#ifdef __cplusplus
}
#endif
#endif /* OGS_MASK_H */
