"""CPU restatement of the reference's image-space mask reductions -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product path
(opengaussian_amd/mask_ops.py -> include/ogs_mask.h) never does.

PINNED: tests/test_oracle_mask.py checks every function below against tests/golden/mask_golden.npz, produced by
tests/golden/make_mask_golden.py from the reference's own function bodies
(/root/reference/utils/opengs_utlis.py, /root/reference/train.py) run on the CPU in the build container.

The restatement is deliberately the segmented form (sum over the pixels of each mask), not a copy of the
reference's [num_mask, C, H, W] expansion, so that it scales to the test sizes; float64 accumulation is
available through `dtype`.
"""
import torch


def mask_feature_mean(feat_map, gt_masks, image_mask=None, return_var=False, dtype=torch.float32):
    """utils/opengs_utlis.py:240-283.  feat_map [C,H,W], gt_masks [N,H,W] 0/1, image_mask [1,H,W] or None."""
    C = feat_map.shape[0]
    f = feat_map.reshape(C, -1).to(dtype)                       # [C, HW]
    m = (gt_masks != 0).reshape(gt_masks.shape[0], -1).to(dtype)  # [N, HW]
    if image_mask is not None:
        w = image_mask.reshape(1, -1).to(dtype)                 # float weights, NOT binarised (:252-257)
        mw = m * w
        sums = mw @ f.t()                                       # sum feat*mask*image_mask          (:256,267)
        counts = mw.sum(dim=1)                                  # sum mask*image_mask               (:257)
    else:
        sums = m @ f.t()                                        # (:260,267)
        counts = m.sum(dim=1)                                   # (:261)
    counts = counts.clamp(min=1)                                # (:264)
    mean = sums / counts[:, None]                               # (:268)
    if not return_var:
        return mean
    # where(mask, feat*mask - mean, 0)^2 summed / counts, then the mean over channels   (:272-283)
    var_c = torch.stack([((f - mean[n][:, None]) ** 2 * m[n][None]).sum(dim=1) for n in range(m.shape[0])]) / counts[:, None]
    return mean, var_c.mean(dim=1), counts


def cohesion_loss(feat_map, gt_mask, feat_mean_stack, dtype=torch.float32):
    """train.py:102-122.  mean over masks of (sum over the mask's pixels of ||feat - mean||_2) / clamp(pixels, 1)."""
    C = feat_map.shape[0]
    f = feat_map.reshape(C, -1).to(dtype)
    m = (gt_mask != 0).reshape(gt_mask.shape[0], -1)
    losses = []
    for n in range(m.shape[0]):
        sel = f[:, m[n]]                                                    # pixels of mask n: masked_feat == feat there (:114)
        dist = (sel - feat_mean_stack[n].to(dtype)[:, None]).norm(p=2, dim=0)  # (:115)
        losses.append(dist.sum() / max(int(m[n].sum()), 1))                 # (:118-119)
    return torch.stack(losses).mean()                                       # (:121)


def calculate_iou(masks1, masks2, base=None):
    """utils/opengs_utlis.py:90-123 -> [m, n]."""
    a = (masks1 != 0).reshape(masks1.shape[0], -1).double()
    b = (masks2 != 0).reshape(masks2.shape[0], -1).double()
    inter = b @ a.t()
    if base == "former":
        union = a.sum(dim=1)[None, :] + 1e-6
    elif base == "later":
        union = b.sum(dim=1)[:, None] + 1e-6
    else:
        union = a.sum(dim=1)[None, :] + b.sum(dim=1)[:, None] - inter + 1e-6
    return (inter / union).float()
