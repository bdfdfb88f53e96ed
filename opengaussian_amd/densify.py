"""Densification bookkeeping on the per-Gaussian parameters and their Adam state (second
half) -- host-side mirror of /root/reference/scene/gaussian_model.py:357-510 over the HIP kernels of
include/ogs_optim.h (csrc/densify.hip).

The reference's ``GaussianModel`` methods are kept by name and meaning, as functions over an optimizer whose param
groups are named like the reference's (``xyz, f_dc, f_rest, opacity, scaling, rotation, ins_feat``,
scene/gaussian_model.py:216-224; ``torch.optim.Adam`` or ``opengaussian_amd.optim.FusedAdam``):

    replace_tensor_to_optimizer(optimizer, tensor, name)          :357-372
    prune_optimizer(optimizer, mask) / prune_points(state, mask)  :374-405
    cat_tensors_to_optimizer(optimizer, tensors_dict)             :412-433
    densify_and_prune(state, max_grad, min_opacity, extent, max_screen_size)   :488-508 (clone + split + prune FUSED)
    add_densification_stats(state, viewspace_grad, update_filter, radii)       :512-514 (+ train.py:597)
    reset_opacity(state)                                           :300-303

``DensifyState`` bundles what those methods touch on the model: the optimizer, ``xyz_gradient_accum``, ``denom``,
``max_radii2D`` and ``percent_dense``.

Where the reference does one boolean index / torch.cat per tensor and per sub-operation (7 parameters + 14 moments
+ 3 statistics, three times per densify_and_prune), every operation here is: decide per row -> ONE row map -> ONE
launch that moves every tensor once.  Results are identical row for row (same order: surviving old rows, clones,
first split children, second split children); split children use the caller's ``samples`` exactly like the
reference's ``torch.normal(mean=0, std=stds)`` draw (:445).  No CPU path: tensors must live on the GPU.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Dict, Optional

import torch
from torch import nn

from . import _lib
from ._lib import OgsDensifyArgs, OgsRowTensor, check, ptr

MOMENTS = ("exp_avg", "exp_avg_sq")


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _row_width(t: torch.Tensor) -> int:
    return int(t[0].numel()) if t.shape[0] else int(torch.Size(t.shape[1:]).numel())


def _need_gpu(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise RuntimeError(f"{what} must live on the GPU (got {t.device}); the densification kernels have no CPU path")


def _f32c(t: torch.Tensor) -> torch.Tensor:
    return t.detach().to(torch.float32).contiguous()


def gather_rows(tensors, src_row: torch.Tensor, kind: Optional[torch.Tensor] = None, zero_new=()):
    """out[k][r] = tensors[k][src_row[r]] for a list of row-major fp32 tensors, in ONE launch per 32 tensors.
    ``zero_new``: indices of tensors whose rows with kind != 0 are written as zeros."""
    lib = _lib.lib()
    n_out = int(src_row.shape[0])
    srcs = [_f32c(t) for t in tensors]
    outs = [torch.empty((n_out,) + tuple(t.shape[1:]), dtype=torch.float32, device=src_row.device) for t in srcs]
    if n_out == 0:
        return outs
    descs = []
    for i, (s, o) in enumerate(zip(srcs, outs)):
        _need_gpu(s, "tensor")
        w = int(torch.Size(s.shape[1:]).numel())
        if w == 0:
            continue
        d = OgsRowTensor()
        d.src, d.dst, d.width, d.zero_new = ptr(s) if s.numel() else ptr(o), ptr(o), w, int(i in zero_new)
        descs.append(d)
    sr = src_row.to(torch.int32).contiguous()
    kd = None if kind is None else kind.to(torch.uint8).contiguous()
    for i in range(0, len(descs), 32):
        chunk = descs[i:i + 32]
        arr = (OgsRowTensor * len(chunk))(*chunk)
        check(lib.ogs_rows_gather(arr, len(chunk), ptr(sr), ptr(kd), n_out, _stream()), "ogs_rows_gather")
    return outs


def _groups(optimizer):
    for group in optimizer.param_groups:
        assert len(group["params"]) == 1, "one tensor per param group, as the reference builds them"
        yield group


def _swap_param(optimizer, group, new_tensor: torch.Tensor, new_state: Optional[dict]):
    """group['params'][0] <- nn.Parameter(new_tensor), optimizer.state re-keyed (the reference's idiom, :365-370)."""
    old = group["params"][0]
    optimizer.state.pop(old, None)
    p = nn.Parameter(new_tensor.requires_grad_(True))
    group["params"][0] = p
    if new_state is not None:
        optimizer.state[p] = new_state
    return p


def _is_sharded(optimizer) -> bool:
    """opengaussian_amd.dp.ShardedAdam: flat replicated parameters, moments only for the owned slice"""
    return hasattr(optimizer, "remap_rows")


def replace_tensor_to_optimizer(optimizer, tensor: torch.Tensor, name: str) -> Dict[str, nn.Parameter]:
    """scene/gaussian_model.py:357-372: swap the parameter of group `name`, zeroing its moments."""
    if _is_sharded(optimizer):
        return optimizer.replace_tensor(name, tensor)
    out = {}
    for group in _groups(optimizer):
        if group["name"] != name:
            continue
        st = optimizer.state.get(group["params"][0], None)
        if st is not None:
            st = dict(st)
            st["exp_avg"] = torch.zeros_like(tensor)
            st["exp_avg_sq"] = torch.zeros_like(tensor)
        out[name] = _swap_param(optimizer, group, tensor, st)
    return out


def _remap_optimizer(optimizer, src_row: torch.Tensor, kind: Optional[torch.Tensor], extras=()):
    """Move every group's parameter and moments (and the `extras` tensors) through the row map in one launch."""
    if _is_sharded(optimizer):
        return optimizer.remap_rows(src_row, kind, extras)
    groups = list(_groups(optimizer))
    tensors, zero_new, slots = [], set(), []
    for gi, group in enumerate(groups):
        p = group["params"][0]
        _need_gpu(p, f"parameter '{group.get('name', gi)}'")
        slots.append((gi, "param"))
        tensors.append(p.data)
        st = optimizer.state.get(p, None)
        if st is not None and "exp_avg" in st:
            for m in MOMENTS:
                zero_new.add(len(tensors))
                slots.append((gi, m))
                tensors.append(st[m])
    for e in extras:
        slots.append((-1, "extra"))
        tensors.append(e)
    outs = gather_rows(tensors, src_row, kind, zero_new)
    result, new_extras = {}, []
    per_group = {}
    for (gi, what), o in zip(slots, outs):
        if gi < 0:
            new_extras.append(o)
        else:
            per_group.setdefault(gi, {})[what] = o
    for gi, group in enumerate(groups):
        st = optimizer.state.get(group["params"][0], None)
        new_state = None
        if st is not None:
            new_state = dict(st)
            for m in MOMENTS:
                if m in per_group[gi]:
                    new_state[m] = per_group[gi][m]
        result[group.get("name", str(gi))] = _swap_param(optimizer, group, per_group[gi]["param"], new_state)
    return result, new_extras


def prune_optimizer(optimizer, mask: torch.Tensor, extras=()):
    """scene/gaussian_model.py:374-389 (`_prune_optimizer`): keep the rows where `mask` is True.  Returns
    ({name: new Parameter}, [extras pruned the same way])."""
    src_row = torch.nonzero(mask.reshape(-1)).flatten().to(torch.int32)
    return _remap_optimizer(optimizer, src_row, None, extras)


def cat_tensors_to_optimizer(optimizer, tensors_dict: Dict[str, torch.Tensor]) -> Dict[str, nn.Parameter]:
    """scene/gaussian_model.py:412-433: append `tensors_dict[name]` to every group, moments extended with zeros."""
    if _is_sharded(optimizer):
        # identity rows + appended rows (copied from row 0, flagged new -> zero moments), then the given values written over
        first = optimizer.names[0]
        n_old, n_new = int(optimizer.shapes[first][0]), int(tensors_dict[first].shape[0])
        dev = optimizer.flat.device
        src = torch.cat((torch.arange(n_old, dtype=torch.int32, device=dev), torch.zeros(n_new, dtype=torch.int32, device=dev)))
        kind = torch.cat((torch.zeros(n_old, dtype=torch.uint8, device=dev), torch.ones(n_new, dtype=torch.uint8, device=dev)))
        params, _ = optimizer.remap_rows(src, kind)
        with torch.no_grad():
            for name, p in params.items():
                p[n_old:].copy_(tensors_dict[name].detach())
        return params
    out = {}
    for group in _groups(optimizer):
        ext = tensors_dict[group["name"]]
        p = group["params"][0]
        st = optimizer.state.get(p, None)
        new_state = None
        if st is not None:
            new_state = dict(st)
            for m in MOMENTS:
                new_state[m] = torch.cat((st[m], torch.zeros_like(ext)), dim=0)
        out[group["name"]] = _swap_param(optimizer, group, torch.cat((p.data, ext.detach()), dim=0), new_state)
    return out


@dataclass
class DensifyState:
    """What the reference's densification methods touch on GaussianModel (scene/gaussian_model.py:88-100,211-214)."""
    optimizer: torch.optim.Optimizer
    xyz_gradient_accum: torch.Tensor       # [N,1]
    denom: torch.Tensor                    # [N,1]
    max_radii2D: torch.Tensor              # [N]
    percent_dense: float = 0.01
    last_plan: Optional[dict] = None       # counts + row map of the last densify_and_prune (diagnostics / tests)

    def params(self) -> Dict[str, nn.Parameter]:
        if _is_sharded(self.optimizer):
            return dict(self.optimizer.params)
        return {g["name"]: g["params"][0] for g in _groups(self.optimizer)}


def prune_points(state: DensifyState, mask: torch.Tensor) -> Dict[str, nn.Parameter]:
    """scene/gaussian_model.py:391-410: remove the rows where `mask` is True (parameters, moments, statistics)."""
    params, (acc, den, rad) = prune_optimizer(state.optimizer, ~mask.reshape(-1),
                                              extras=(state.xyz_gradient_accum, state.denom, state.max_radii2D.reshape(-1, 1)))
    state.xyz_gradient_accum, state.denom, state.max_radii2D = acc, den, rad.reshape(-1)
    return params


def add_densification_stats(state: DensifyState, viewspace_grad: torch.Tensor, update_filter: Optional[torch.Tensor] = None,
                            radii: Optional[torch.Tensor] = None):
    """scene/gaussian_model.py:512-514 and, when `radii` is given, the max_radii2D update of train.py:597 -- one pass.
    `update_filter` None: radii > 0 (what train.py passes as visibility_filter)."""
    if viewspace_grad is None:
        # render(..., viewspace_grad=None) stops producing dL/dmeans2D once every geometry tensor is detached (stage >= 1,
        # train.py:431-436): a schedule that still densifies there must ask for it explicitly
        raise RuntimeError("add_densification_stats: viewspace_points.grad is None -- the rasterizer ran its features-only "
                           "backward (all geometry tensors detached); call render(..., viewspace_grad=True) while densifying")
    _need_gpu(viewspace_grad, "viewspace_grad")
    g = _f32c(viewspace_grad)
    N = int(g.shape[0])
    vis = None if update_filter is None else update_filter.reshape(-1).to(torch.uint8).contiguous()
    rad = None if radii is None else radii.to(torch.int32).contiguous()
    if vis is None and rad is None:
        raise RuntimeError("add_densification_stats needs update_filter or radii")
    for t in (state.xyz_gradient_accum, state.denom):
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise RuntimeError("xyz_gradient_accum / denom must be contiguous fp32 GPU tensors (updated in place)")
    mr = state.max_radii2D if rad is not None else None
    check(_lib.lib().ogs_densify_stats(N, ptr(g), int(g.shape[1]), ptr(vis), ptr(rad), ptr(state.xyz_gradient_accum),
                                       ptr(state.denom), ptr(mr), _stream()), "ogs_densify_stats")


def reset_opacity(state: DensifyState) -> Dict[str, nn.Parameter]:
    """scene/gaussian_model.py:300-303: opacity <- inverse_sigmoid(min(sigmoid(opacity), 0.01)), moments zeroed."""
    op = state.params()["opacity"].data
    new = torch.min(torch.sigmoid(op), torch.ones_like(op) * 0.01)
    return replace_tensor_to_optimizer(state.optimizer, torch.log(new / (1 - new)), "opacity")


def densify_and_prune(state: DensifyState, max_grad: float, min_opacity: float, extent: float, max_screen_size,
                      samples: Optional[torch.Tensor] = None, generator: Optional[torch.Generator] = None):
    """scene/gaussian_model.py:488-508 -- densify_and_clone + densify_and_split (N = 2) + the final prune, fused.

    `samples` [2S,3]: the split children's offsets in the parent's frame, S = number of split parents, rows
    [0,S) for the first copies and [S,2S) for the second -- exactly the reference's
    ``torch.normal(mean=0, std=get_scaling[selected].repeat(2,1))`` (:443-445).  None: drawn here the same way.
    Returns {name: new Parameter}; `state`'s statistics are reset to zeros of the new size (:434-436)."""
    lib = _lib.lib()
    P = state.params()
    xyz = P["xyz"].data
    _need_gpu(xyz, "xyz")
    dev = xyz.device
    N = int(xyz.shape[0])
    scaling, rotation, opacity = _f32c(P["scaling"].data), _f32c(P["rotation"].data), _f32c(P["opacity"].data)
    a = OgsDensifyArgs()
    a.N = N
    acc, den = _f32c(state.xyz_gradient_accum), _f32c(state.denom)
    a.grad_accum, a.denom, a.scaling, a.opacity = ptr(acc), ptr(den), ptr(scaling), ptr(opacity)
    a.max_grad, a.min_opacity, a.extent, a.percent_dense = float(max_grad), float(min_opacity), float(extent), float(state.percent_dense)
    a.prune_world_size = int(bool(max_screen_size))
    tmp = torch.empty(int(lib.ogs_densify_tmp_bytes(N)), dtype=torch.uint8, device=dev)
    totals = (C.c_uint32 * 4)()
    check(lib.ogs_densify_plan(C.byref(a), ptr(tmp), totals, _stream()), "ogs_densify_plan")
    nA, nB, nC, S = (int(v) for v in totals)
    n_out = nA + nB + 2 * nC
    src_row = torch.empty(n_out, dtype=torch.int32, device=dev)
    kind = torch.empty(n_out, dtype=torch.uint8, device=dev)
    sample_row = torch.empty(n_out, dtype=torch.int32, device=dev)
    if n_out:
        check(lib.ogs_densify_map(N, ptr(tmp), ptr(src_row), ptr(kind), ptr(sample_row), _stream()), "ogs_densify_map")
    if samples is None:
        # the reference's draw (:443-445): torch.normal(mean=0, std=scaling of the selected parents, repeated for the
        # two copies).  Only the children that survive the final prune consume their draw, so only those rows are
        # drawn (row sample_row[r] of the [2S,3] table belongs to child r; the rows of pruned children stay zero)
        child = kind >= 2
        stds = torch.exp(scaling[src_row[child].long()])
        samples = torch.zeros(2 * S, 3, dtype=torch.float32, device=dev)
        samples[sample_row[child].long()] = torch.normal(mean=torch.zeros_like(stds), std=stds, generator=generator)
        if _is_sharded(state.optimizer) and state.optimizer.on:
            # every rank must create the SAME children: rank 0's draw is the one that counts
            import torch.distributed as dist
            dist.broadcast(samples, src=dist.get_global_rank(state.optimizer.group, 0) if state.optimizer.group is not None else 0,
                           group=state.optimizer.group)
    samples = _f32c(samples)
    if samples.shape != (2 * S, 3):
        raise RuntimeError(f"samples must be [2*S,3] with S={S} selected split parents, got {tuple(samples.shape)}")
    old_xyz = _f32c(xyz)
    params, _ = _remap_optimizer(state.optimizer, src_row, kind)
    if nC:
        check(lib.ogs_densify_split_children(n_out, ptr(src_row), ptr(kind), ptr(sample_row), ptr(old_xyz), ptr(scaling),
                                             ptr(rotation), ptr(samples), ptr(params["xyz"].data), ptr(params["scaling"].data),
                                             _stream()), "ogs_densify_split_children")
    state.xyz_gradient_accum = torch.zeros(n_out, 1, device=dev)
    state.denom = torch.zeros(n_out, 1, device=dev)
    state.max_radii2D = torch.zeros(n_out, device=dev)
    state.last_plan = {"kept": nA, "clones": nB, "split_children": 2 * nC, "split_parents_selected": S,
                       "src_row": src_row, "kind": kind}
    return params
