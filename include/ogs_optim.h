/*
 * ogs_optim.h -- C ABI of the fused multi-tensor Adam step (first half).
 *
 * Replaces the optimizer step of /root/reference/scene/gaussian_model.py:216-230
 * (`torch.optim.Adam(l, lr=0.0, eps=1e-15)` over the 7 per-Gaussian parameter groups xyz, f_dc, f_rest,
 * opacity, scaling, rotation, ins_feat, stepped at train.py:594-611) by ONE launch over all groups: each
 * element of param / grad / exp_avg / exp_avg_sq is read once and param / exp_avg / exp_avg_sq written once
 * (28 B per element: HBM-bound), instead of one pass per elementwise op.
 *
 * Arithmetic follows torch's single-tensor Adam (no amsgrad, no weight decay, maximize = False) operation by
 * operation in fp32 with the scalar factors rounded from double exactly as torch does:
 *     m <- fma(1 - beta1, g - m, m)                                          (torch lerp_)
 *     v <- fma((1 - beta2) * g, g, v * beta2)                                (mul_ + addcmul_)
 *     p <- p + ((-lr / (1 - beta1^t)) * m) / (sqrt(v) / sqrt(1 - beta2^t) + eps)   (sqrt, div, add_, addcdiv_)
 *
 * All tensor pointers are DEVICE pointers to contiguous fp32; the descriptor array itself is HOST memory.
 */
#ifndef OGS_OPTIM_H
#define OGS_OPTIM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OGS_ADAM_MAX_TENSORS 16

typedef struct OgsAdamTensor {
    float* param;
    const float* grad;
    float* exp_avg;
    float* exp_avg_sq;
    int64_t numel;
    double lr;           /* the group's learning rate */
    int64_t step;        /* t >= 1: the step count AFTER this update (torch increments before using it) */
} OgsAdamTensor;

/* One Adam update of `count` (<= OGS_ADAM_MAX_TENSORS) tensors in a single launch.  Asynchronous on `stream`. */
int ogs_adam_step(const OgsAdamTensor* tensors, int32_t count, double beta1, double beta2, double eps, void* stream);

/* ---- densification bookkeeping (second half) -------------------------------------------
 *
 * Replaces the tensor surgery of /root/reference/scene/gaussian_model.py:357-510 -- replace_tensor_to_optimizer,
 * _prune_optimizer / prune_points, cat_tensors_to_optimizer / densification_postfix, densify_and_clone,
 * densify_and_split, densify_and_prune, add_densification_stats -- driven from train.py:594-611.  The reference
 * rebuilds the seven parameter tensors and their fourteen Adam moments with one boolean index or torch.cat per
 * tensor and per operation (clone, split, prune: ~100 passes over the state per densify_and_prune); here the
 * decisions of the whole operation are taken in one pass over the rows, turned into ONE row map, and every
 * tensor moves once, in one launch.  Host side: opengaussian_amd/densify.py. */

#define OGS_ROWS_MAX_TENSORS 32

typedef struct OgsRowTensor {
    const float* src;     /* [n_in, width]  */
    float* dst;           /* [n_out, width] */
    int32_t width;        /* floats per row (product of the trailing dimensions) */
    int32_t zero_new;     /* !=0: rows whose kind != 0 (not an old row) are written as zeros (Adam moments) */
} OgsRowTensor;

/* dst[r, :] = src[src_row[r], :] for every tensor, r in [0, n_out); src_row[r] < 0 -> zeros.  `kind` (uint8 [n_out],
 * may be NULL) marks new rows for the zero_new tensors.  One launch for <= OGS_ROWS_MAX_TENSORS tensors.
 * prune_points (:391-405) is this with src_row = the kept row numbers; cat_tensors_to_optimizer (:412-433) is it with
 * the extension rows appended to the map. */
int ogs_rows_gather(const OgsRowTensor* tensors, int32_t count, const int32_t* src_row, const uint8_t* kind, int64_t n_out,
                    void* stream);

typedef struct OgsDensifyArgs {
    int32_t N;                  /* Gaussians before the call */
    const float* grad_accum;    /* [N] xyz_gradient_accum                         (:489) */
    const float* denom;         /* [N]                                                     */
    const float* scaling;       /* [N,3] the raw (log) scaling parameter _scaling           */
    const float* opacity;       /* [N] the raw (logit) opacity parameter _opacity           */
    float max_grad;             /* densify_grad_threshold                         (train.py:602) */
    float min_opacity;          /* 0.005                                                  */
    float extent;               /* scene.cameras_extent                                   */
    float percent_dense;        /* GaussianModel.percent_dense                            */
    int32_t prune_world_size;   /* != 0 when max_screen_size is given (:498-501): also prune world size > 0.1 * extent */
} OgsDensifyArgs;

size_t ogs_densify_tmp_bytes(int32_t N);

/* Phase 1: per-row decisions of densify_and_prune + their prefix sums (kept in `tmp`), totals_host[4] (HOST memory)
 * = { surviving old rows, surviving clones, surviving split parents (each yields 2 children), selected split parents
 * S (= half the rows of the reference's `samples` draw) }.  Synchronises the stream (one 16-byte read-back; the
 * reference synchronises on every boolean index here). */
int ogs_densify_plan(const OgsDensifyArgs* args, void* tmp, uint32_t* totals_host, void* stream);

/* Phase 2: the row map of the result in the reference's final order [old | clones | first children | second children]:
 * src_row[n_out] (parent row), kind[n_out] (0 old, 1 clone, 2 / 3 first / second split child), sample_row[n_out] (row
 * of the child's draw in `samples` [2S,3], -1 otherwise); n_out = totals[0] + totals[1] + 2 * totals[2]. */
int ogs_densify_map(int32_t N, void* tmp, int32_t* src_row, uint8_t* kind, int32_t* sample_row, void* stream);

/* Phase 3 (after ogs_rows_gather): the split children's position and scaling (:446-448):
 * new_xyz[r] = R(q_parent / |q_parent|) . samples[sample_row[r]] + xyz_parent, new_scaling[r] = log(exp(s_parent) / 1.6).
 * xyz / scaling / rotation are the tensors BEFORE the call; new_xyz / new_scaling the gathered [n_out,3] ones. */
int ogs_densify_split_children(int64_t n_out, const int32_t* src_row, const uint8_t* kind, const int32_t* sample_row,
                               const float* xyz, const float* scaling, const float* rotation, const float* samples,
                               float* new_xyz, float* new_scaling, void* stream);

/* add_densification_stats (:512-514) + the max_radii2D update of train.py:597 in one pass:
 * where visible (uint8 [N], or radii > 0 when NULL): accum += ||grad_means2D[i, :2]||, denom += 1,
 * max_radii2D = max(max_radii2D, radii) (when both given).  grad_means2D is [N, stride] (stride 3 for means2D.grad). */
int ogs_densify_stats(int32_t N, const float* grad_means2D, int32_t stride, const uint8_t* visible, const int32_t* radii,
                      float* accum, float* denom, float* max_radii2D, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* OGS_OPTIM_H */
