"""Image-space mask reductions of OpenGaussian's stage-1 / association losses.

Same names, arguments and results as the reference's
  utils/opengs_utlis.py:90-123    calculate_iou
  utils/opengs_utlis.py:181-197   pair_mask_feature_mean
  utils/opengs_utlis.py:240-283   mask_feature_mean
  train.py:102-122                cohesion_loss
  train.py:124-155                separation_loss
but without the [num_mask, C, H, W] expansions: the two hot ones (mask_feature_mean, cohesion_loss, called every
stage-1 step, train.py:450-452) are segmented reductions in HIP (include/ogs_mask.h) with hand-written
backward passes; the others are small and stay plain torch on the GPU.  No CPU path: CPU tensors raise.
"""
from __future__ import annotations

import torch

from . import _lib
from ._lib import check, ptr


TABLE_STRIDE = 16      # OGS_MASK_TABLE_STRIDE


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _need_gpu(t: torch.Tensor, name: str):
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live on the GPU (got {t.device}); the MI355X mask reductions have no CPU path")


def _mask_bytes(masks: torch.Tensor) -> torch.Tensor:
    """[N,H,W] masks of any dtype / layout (bool stacks, the permuted int64 one-hot of get_SAM_mask_and_feat,
    opengs_utlis.py:147-149) -> contiguous uint8 0/1."""
    if masks.dtype == torch.bool and masks.is_contiguous():
        return masks.view(torch.uint8)
    if masks.dtype == torch.uint8 and masks.is_contiguous():
        return masks
    return (masks != 0).to(torch.uint8).contiguous()


class _MaskSums(torch.autograd.Function):
    """table [N, C+1] (or [N, 2C+1] with squares) of weighted per-mask feature sums | weighted counts."""

    @staticmethod
    def forward(ctx, feat_map, masks_u8, weight, with_squares):
        lib = _lib.lib()
        f = feat_map.detach().to(torch.float32).contiguous()
        C, H, W = (int(x) for x in f.shape)
        N = int(masks_u8.shape[0])
        w = None if weight is None else weight.detach().to(torch.float32).reshape(H, W).contiguous()
        width = (2 * C + 1) if with_squares else (C + 1)
        table = torch.empty(N, TABLE_STRIDE, dtype=torch.float32, device=f.device)   # 64-byte rows (ogs_mask.h)
        check(lib.ogs_mask_feature_sums(ptr(f), ptr(masks_u8), ptr(w), C, N, H * W, int(bool(with_squares)),
                                        ptr(table), _stream()), "ogs_mask_feature_sums")
        ctx.save_for_backward(masks_u8, w, f)
        ctx.shape = (C, H, W, N)
        ctx.weight_shape = None if weight is None else tuple(weight.shape)
        ctx.with_squares = bool(with_squares)
        return table[:, :width]

    @staticmethod
    def backward(ctx, g_table):
        if ctx.with_squares:
            raise RuntimeError("mask_feature_mean(return_var=True) is not differentiable here (the reference only "
                               "uses it under no_grad, train.py:689)")
        masks_u8, w, f = ctx.saved_tensors
        C, H, W, N = ctx.shape
        # d table[n, c] / d feat[c, pix] = mask * w ;  d table[n, c] / d w[pix] = mask * feat[c, pix] ;
        # d table[n, C] / d w[pix] = mask
        coef = g_table[:, :C].to(torch.float32).contiguous()
        need_w = w is not None and ctx.needs_input_grad[2]
        coef_cnt = g_table[:, C].to(torch.float32).contiguous() if need_w else None
        dfeat = torch.empty(C, H, W, dtype=torch.float32, device=g_table.device)
        dweight = torch.empty(H, W, dtype=torch.float32, device=g_table.device) if need_w else None
        check(_lib.lib().ogs_mask_feature_sums_backward(ptr(masks_u8), ptr(w), ptr(coef), ptr(f), ptr(coef_cnt), C, N,
                                                        H * W, ptr(dfeat), ptr(dweight), _stream()),
              "ogs_mask_feature_sums_backward")
        return (dfeat if ctx.needs_input_grad[0] else None, None,
                dweight.reshape(ctx.weight_shape) if need_w else None, None)


def mask_feature_mean(feat_map, gt_masks, image_mask=None, return_var=False):
    """Average instance feature inside each mask (utils/opengs_utlis.py:240-283).
    feat_map [C=3|6, H, W]; gt_masks [num_mask, H, W] (0/1 of any dtype); image_mask [1,H,W] / [H,W] weights
    (the rendered silhouette, train.py:450) or None.  Returns [num_mask, C]; with return_var=True
    (mean [N,C], variance [N], pixel count [N]) as the reference."""
    _need_gpu(feat_map, "feat_map")
    C, H, W = (int(x) for x in feat_map.shape)
    m = _mask_bytes(gt_masks.to(feat_map.device))
    if tuple(m.shape[1:]) != (H, W):
        raise RuntimeError(f"gt_masks must be [num_mask, {H}, {W}], got {tuple(gt_masks.shape)}")
    if image_mask is not None:
        if image_mask.numel() != H * W:
            raise RuntimeError(f"image_mask must have H*W = {H * W} elements, got {tuple(image_mask.shape)}")
        image_mask = image_mask.to(feat_map.device)
    table = _MaskSums.apply(feat_map, m, image_mask, bool(return_var))
    counts = table[:, C].clamp(min=1)
    mean = table[:, :C] / counts[:, None]
    if not return_var:
        return mean
    # sum mask*(f - mean)^2 = sum f^2 - 2 mean sum f + count mean^2   (masked_for_variance, :272-276)
    raw_count = table[:, C]
    var_c = (table[:, C + 1:] - 2.0 * mean * table[:, :C] + raw_count[:, None] * mean * mean) / counts[:, None]
    return mean, var_c.clamp_min(0).mean(dim=1), counts


class _Cohesion(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat_map, masks_u8, mean):
        lib = _lib.lib()
        f = feat_map.detach().to(torch.float32).contiguous()
        mu = mean.detach().to(torch.float32).contiguous()
        C, H, W = (int(x) for x in f.shape)
        N = int(masks_u8.shape[0])
        table = torch.empty(N, TABLE_STRIDE, dtype=torch.float32, device=f.device)
        check(lib.ogs_mask_cohesion(ptr(f), ptr(masks_u8), ptr(mu), C, N, H * W, ptr(table), _stream()),
              "ogs_mask_cohesion")
        cnt = table[:, 1].clamp(min=1)
        ctx.save_for_backward(f, masks_u8, mu, cnt)
        return (table[:, 0] / cnt).mean() if N > 0 else table.sum()

    @staticmethod
    def backward(ctx, g):
        f, masks_u8, mu, cnt = ctx.saved_tensors
        C, H, W = (int(x) for x in f.shape)
        N = int(masks_u8.shape[0])
        gl = (g.to(torch.float32) / (max(N, 1) * cnt)).contiguous()
        dfeat = torch.empty(C, H, W, dtype=torch.float32, device=f.device)
        dmean = torch.empty(N, TABLE_STRIDE, dtype=torch.float32, device=f.device)
        check(_lib.lib().ogs_mask_cohesion_backward(ptr(f), ptr(masks_u8), ptr(mu), ptr(gl), C, N, H * W, ptr(dfeat),
                                                    ptr(dmean), _stream()), "ogs_mask_cohesion_backward")
        return dfeat, None, dmean[:, :C]


def cohesion_loss(feat_map, gt_mask, feat_mean_stack):
    """Intra-mask smoothing loss, Eq. (1) of the paper (train.py:102-122): mean over masks of the mean L2 distance
    between the pixels of a mask and that mask's mean feature."""
    _need_gpu(feat_map, "feat_map")
    m = _mask_bytes(gt_mask.to(feat_map.device))
    return _Cohesion.apply(feat_map, m, feat_mean_stack)


class _Separation(torch.autograd.Function):
    """loss and dloss/dmeans from ogs_separation_loss (two launches); backward scales the stored gradient."""

    @staticmethod
    def forward(ctx, means, late: bool):
        N, Cc = means.shape
        m = means.detach().contiguous()
        dev = m.device
        loss = torch.empty((), dtype=torch.float32, device=dev)
        want_grad = means.requires_grad
        grad = torch.empty(N, Cc, dtype=torch.float32, device=dev) if want_grad else None
        tmp = torch.empty(N * N + N, dtype=torch.float32, device=dev)
        check(_lib.lib().ogs_separation_loss(ptr(m), N, Cc, int(bool(late)), ptr(loss), ptr(grad), ptr(tmp), _stream()),
              "ogs_separation_loss")
        ctx.grad = grad
        return loss

    @staticmethod
    def backward(ctx, g):
        return (None if ctx.grad is None else ctx.grad * g), None


def _separation_loss_torch(feat_mean_stack, iteration):
    N, _ = feat_mean_stack.shape
    diff_squared = (feat_mean_stack.unsqueeze(1) - feat_mean_stack.unsqueeze(0)).pow(2).sum(2)
    inverse_distance = 1.0 / (diff_squared + 1)
    eye = torch.eye(N, device=feat_mean_stack.device).bool()
    inverse_distance = inverse_distance.masked_fill(eye, 0)
    sorted_indices = inverse_distance.argsort().argsort()
    loss_weight = (sorted_indices.float() / (N - 1)) * (1.0 - 0.1) + 0.1
    if iteration > 35_000:
        loss_weight[loss_weight < 0.9] = 0.1
    return (inverse_distance * loss_weight).sum() / (N * (N - 1))


def separation_loss(feat_mean_stack, iteration):
    """Inter-mask contrastive loss, Eq. (2) (train.py:124-155).  On the GPU (fp32, 2 <= N <= 1024 masks, C <= 16): the
    whole [N, N] computation -- pairwise inverse distances, the rank of every element inside its row
    (argsort().argsort()), the rank weights, the sum AND the gradient -- in two small launches instead of ~30
    launch-bound torch kernels and two segmented sorts (0.3-0.45 ms -> 0.03 ms forward + backward).  Anything else (CPU
    tensors of the host-logic tests, other dtypes, N = 1) takes the literal torch formulation."""
    N, Cc = feat_mean_stack.shape
    if feat_mean_stack.is_cuda and feat_mean_stack.dtype is torch.float32 and 2 <= N <= 1024 and 1 <= Cc <= 16:
        return _Separation.apply(feat_mean_stack, iteration > 35_000)
    return _separation_loss_torch(feat_mean_stack, iteration)


def pair_mask_feature_mean(feat_map, masks):
    """Mean feature of N (map, mask) pairs (utils/opengs_utlis.py:181-197): feat_map [N,C,H,W], masks [N,H,W].
    One pass over data that already has the N*C*H*W shape -- nothing to fuse; torch on the GPU."""
    m = masks.unsqueeze(1).float()
    return (feat_map * m).sum(dim=[2, 3]) / (m.expand(-1, feat_map.shape[1], -1, -1).sum(dim=[2, 3]) + 1e-6)


def calculate_iou(masks1, masks2, base=None):
    """IoU matrix [m, n] of masks2 [m,H,W] against masks1 [n,H,W] (utils/opengs_utlis.py:90-123).  The
    intersection counts are one {0,1} GEMM over the pixels (fp32: sums of 0/1 products are exact below 2^24
    pixels) instead of an [m, n, H, W] boolean expansion."""
    _need_gpu(masks1, "masks1")
    a = (masks1 != 0).flatten(1)
    b = (masks2 != 0).flatten(1)
    if a.shape[1] >= (1 << 24):
        raise RuntimeError("calculate_iou: more than 2^24 pixels per mask")
    inter = b.float() @ a.float().t()                                            # [m, n]
    ca = a.sum(dim=1).float()[None, :]
    cb = b.sum(dim=1).float()[:, None]
    if base == "former":
        union = ca + 1e-6
    elif base == "later":
        union = cb + 1e-6
    else:
        union = ca + cb - inter + 1e-6
    return inter / union
