"""Fused Adam for the per-Gaussian parameter groups (first half).

Drop-in for ``torch.optim.Adam(l, lr=0.0, eps=1e-15)`` as the reference builds it
(/root/reference/scene/gaussian_model.py:216-230): same constructor, same ``param_groups`` (the reference
rewrites ``group['lr']`` every iteration, :236-247) and the same per-parameter state dict
``{"step", "exp_avg", "exp_avg_sq"}``, so the densification code that reaches into ``optimizer.state``
(``replace_tensor_to_optimizer`` / ``_prune_optimizer`` / ``cat_tensors_to_optimizer``, :357-430) keeps working.
``step()`` is ONE HIP launch over all groups that share betas / eps -- all seven of the reference's -- (include/
ogs_optim.h; ``last_step_launches`` records the count) instead of one pass per elementwise op and group.  No CPU path: parameters must live on the GPU.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import OgsAdamTensor, check

MAX_TENSORS = 16      # OGS_ADAM_MAX_TENSORS


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        if lr < 0.0 or eps < 0.0 or not (0.0 <= betas[0] < 1.0) or not (0.0 <= betas[1] < 1.0):
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.lib()
        stream = torch.cuda.current_stream().cuda_stream
        keep = []                                   # contiguous gradient copies must outlive the launch call
        # descriptors of EVERY group, bucketed by (beta1, beta2, eps) -- launch-level constants of ogs_adam_step.  The
        # reference's seven groups share them (scene/gaussian_model.py:230), so a step is ONE launch.
        buckets = {}
        for group in self.param_groups:
            beta1, beta2 = group["betas"]
            descs = buckets.setdefault((float(beta1), float(beta2), float(group["eps"])), [])
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda:
                    raise RuntimeError("FusedAdam: parameters must live on the GPU (no CPU path)")
                if p.dtype != torch.float32 or not p.is_contiguous():
                    raise RuntimeError("FusedAdam: parameters must be contiguous fp32 tensors")
                if p.grad.is_sparse:
                    raise RuntimeError("FusedAdam does not support sparse gradients")
                g = p.grad
                if g.dtype != torch.float32 or not g.is_contiguous():
                    g = g.to(torch.float32).contiguous()
                    keep.append(g)
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                for k in ("exp_avg", "exp_avg_sq"):
                    # the reference replaces these tensors when it prunes / densifies: accept whatever it stored
                    if st[k].shape != p.shape or not st[k].is_contiguous() or st[k].dtype != torch.float32 or not st[k].is_cuda:
                        raise RuntimeError(f"FusedAdam: state['{k}'] must be a contiguous fp32 GPU tensor shaped like its parameter")
                st["step"] += 1
                d = OgsAdamTensor()
                d.param, d.grad = p.data_ptr(), g.data_ptr()
                d.exp_avg, d.exp_avg_sq = st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
                d.numel, d.lr, d.step = p.numel(), float(group["lr"]), int(st["step"].item())
                descs.append(d)
        self.last_step_launches = 0
        for (beta1, beta2, eps), descs in buckets.items():
            for i in range(0, len(descs), MAX_TENSORS):
                chunk = descs[i:i + MAX_TENSORS]
                arr = (OgsAdamTensor * len(chunk))(*chunk)
                check(lib.ogs_adam_step(arr, len(chunk), beta1, beta2, eps, stream), "ogs_adam_step")
                self.last_step_launches += 1
        return loss
