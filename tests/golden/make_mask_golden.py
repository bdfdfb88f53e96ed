"""Generate tests/golden/mask_golden.npz by RUNNING THE REFERENCE's own function bodies on the CPU.

Run in the build container only (``python tests/golden/make_mask_golden.py``): /root/reference never travels,
only the vectors do.  utils/opengs_utlis.py cannot be imported as a module here (it imports `bitarray`, which
is not installed) and train.py pulls the whole training stack, so the pure-torch functions this path replaces
are compiled one by one from their source text (ast) into a namespace holding only torch / F:

    utils/opengs_utlis.py: calculate_iou, pair_mask_feature_mean, process_in_chunks,
                           calculate_variance_in_chunks, ele_multip_in_chunks, mask_feature_mean
    train.py:              cohesion_loss, separation_loss

Inputs are regenerated from the seed by ``case_inputs`` (shared with the tests); the file stores the outputs and
the autograd gradients of the stage-1 loss (train.py:450-456) w.r.t. the feature map and the silhouette.
"""
import ast
import os

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
REF_UTILS = "/root/reference/utils/opengs_utlis.py"
REF_TRAIN = "/root/reference/train.py"
WANT_UTILS = ["calculate_iou", "pair_mask_feature_mean", "process_in_chunks", "calculate_variance_in_chunks",
              "ele_multip_in_chunks", "mask_feature_mean"]
WANT_TRAIN = ["cohesion_loss", "separation_loss"]

# (seed, C, H, W, num_mask, overlap)
CASES = [(0, 6, 24, 36, 7, False), (1, 6, 31, 29, 12, False), (2, 3, 16, 20, 5, True), (3, 6, 40, 52, 20, False)]


def case_inputs(seed, C, H, W, N, overlap):
    g = torch.Generator().manual_seed(500 + seed)
    feat = torch.rand(C, H, W, generator=g)
    # a label image of blocky regions (like SAM masks), label 0 = invalid pixels -> mask stack excludes it
    coarse = torch.randint(0, N + 1, ((H + 3) // 4, (W + 3) // 4), generator=g)
    labels = coarse.repeat_interleave(4, 0).repeat_interleave(4, 1)[:H, :W]
    noise = torch.rand(H, W, generator=g) < 0.1
    labels = torch.where(noise, torch.randint(0, N + 1, (H, W), generator=g), labels)
    masks = torch.stack([labels == (n + 1) for n in range(N)])          # [N,H,W] bool, disjoint
    if overlap:
        masks = masks | (torch.rand(N, H, W, generator=g) < 0.15)
    masks[N - 1] = False                                                # an empty mask: counts clamp at 1
    sil = torch.rand(1, H, W, generator=g)                              # float silhouette weights
    masks2 = torch.rand(max(N // 2, 2), H, W, generator=g) < 0.3
    return feat, masks, sil, masks2


def load_functions(path, names):
    tree = ast.parse(open(path).read())
    ns = {"torch": torch, "F": F, "np": np}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            mod = ast.Module(body=[node], type_ignores=[])
            exec(compile(mod, path, "exec"), ns)
    missing = [n for n in names if n not in ns]
    assert not missing, missing
    return ns


def main():
    u = load_functions(REF_UTILS, WANT_UTILS)
    t = load_functions(REF_TRAIN, WANT_TRAIN)
    out = {}
    for (seed, C, H, W, N, overlap) in CASES:
        feat, masks, sil, masks2 = case_inputs(seed, C, H, W, N, overlap)
        k = f"s{seed}"
        fm = feat.clone().requires_grad_(True)
        sw = sil.clone().requires_grad_(True)        # the silhouette is a rasterizer output: part of the graph
        mean_w = u["mask_feature_mean"](fm, masks, image_mask=sw)
        coh = t["cohesion_loss"](fm, masks, mean_w)
        sep = t["separation_loss"](mean_w, 1000)
        loss = sep + 0.1 * coh                                           # train.py:456
        loss.backward()
        out[k + "_mean_w"] = mean_w.detach().numpy()
        out[k + "_cohesion"] = np.float32(coh.item())
        out[k + "_separation"] = np.float32(sep.item())
        out[k + "_dfeat"] = fm.grad.numpy()
        out[k + "_dsil"] = sw.grad.numpy()
        out[k + "_mean"] = u["mask_feature_mean"](feat, masks).numpy()
        mv, var, cnt = u["mask_feature_mean"](feat, masks, return_var=True)
        out[k + "_var"] = var.numpy(); out[k + "_cnt"] = cnt.numpy()
        # the one-hot int64 permuted layout get_SAM_mask_and_feat hands over (opengs_utlis.py:147-149)
        out[k + "_mean_int64"] = u["mask_feature_mean"](feat, masks.long(), image_mask=sil).numpy()
        for base in (None, "former", "later"):
            out[k + f"_iou_{base}"] = u["calculate_iou"](masks, masks2, base=base).numpy()
        pm = torch.rand(N, C, H, W, generator=torch.Generator().manual_seed(seed))
        out[k + "_pair"] = u["pair_mask_feature_mean"](pm, masks).numpy()
        out[k + "_sep_late"] = np.float32(t["separation_loss"](mean_w.detach(), 40000).item())
    np.savez_compressed(os.path.join(HERE, "mask_golden.npz"), **out)
    print("wrote", len(out), "arrays")


if __name__ == "__main__":
    main()
