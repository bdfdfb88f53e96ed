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

#ifdef __cplusplus
}
#endif
#endif /* OGS_OPTIM_H */
