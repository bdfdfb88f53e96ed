"""One-view-per-GPU data parallelism for the rasterizer path (new capability; the reference is
single-process).

One process per GPU (torch.distributed, backend "nccl" == RCCL over xGMI on ROCm; "gloo" on CPU for the
tests).  Gaussians are replicated; rank r renders view r of the mini-batch; the only exchange per step is
  * SUM all-reduce of the flattened per-Gaussian gradients (one flat bucket -> few, large collectives,
    which is what point-to-point xGMI links want), issued asynchronously so it overlaps the next pass;
  * the non-linear densification statistics, which cannot be recovered from summed gradients
    (scene/gaussian_model.py:512-514, train.py:597): SUM of per-view ||grad means2D[:, :2]||, SUM of
    visibility counts, MAX of radii.
The partition has no data-path collective: views are independent (weak scaling).
"""
from __future__ import annotations

from typing import Iterable, List, Optional, Sequence

import torch
import torch.distributed as dist


import os as _os

# Rehearsal switch: run every collective even in a ONE-rank group (RCCL executes a 1-rank all-reduce / all-gather /
# reduce-scatter like any other), so the exact N > 1 code path -- nccl init with device_id, side streams, async
# work handles, reduce_scatter_tensor -- can be driven on a single-GPU box (tests/test_50_dp_gpu.py, bench.py under
# `torch.distributed.run --nproc-per-node 1`).  Never set for measurements.
FORCE_COLLECTIVES = _os.environ.get("OGS_DP_FORCE_COLLECTIVES", "0") == "1"


def collectives_on(group=None) -> bool:
    """True when the exchange must actually call the backend: more than one rank, or the rehearsal switch."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or FORCE_COLLECTIVES


def init_from_env(device_type: str = "cuda") -> tuple[int, int, int]:
    """(rank, world_size, local_rank) from torchrun's env; initialises the default group if needed."""
    import os
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or FORCE_COLLECTIVES) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        # "nccl" is RCCL on ROCm.  OGS_DIST_BACKEND=gloo lets several ranks share one GPU for functional
        # rehearsals of the N > 1 path on a single-GPU box (never used for measurements).
        backend = os.environ.get("OGS_DIST_BACKEND", "nccl" if device_type == "cuda" else "gloo")
        if device_type == "cuda":
            dev = local % max(torch.cuda.device_count(), 1)
            torch.cuda.set_device(dev)
            if backend == "nccl":
                dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", dev))
            else:
                dist.init_process_group(backend, rank=rank, world_size=world)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


class GradBucket:
    """Flat fp32 buffer holding a fixed list of per-Gaussian gradient tensors, all-reduced as ONE message."""

    def __init__(self, shapes: Sequence[Sequence[int]], device, group=None, average: bool = True):
        self.shapes = [tuple(s) for s in shapes]
        self.sizes = [int(torch.Size(s).numel()) for s in self.shapes]
        self.flat = torch.zeros(sum(self.sizes), dtype=torch.float32, device=device)
        self.group = group
        self.average = average
        self._work = None
        self._stream = torch.cuda.Stream(device) if torch.device(device).type == "cuda" else None

    def views(self) -> List[torch.Tensor]:
        out, off = [], 0
        for s, n in zip(self.shapes, self.sizes):
            out.append(self.flat[off:off + n].view(s))
            off += n
        return out

    def pack(self, grads: Iterable[Optional[torch.Tensor]]):
        for v, g in zip(self.views(), grads):
            if g is None:
                v.zero_()
            else:
                v.copy_(g.reshape(v.shape))

    def allreduce_async(self):
        """Start the SUM all-reduce on a side stream (overlaps whatever the caller enqueues next)."""
        if not collectives_on(self.group):
            return
        if self._stream is not None:
            self._stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._stream):
                self._work = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            self._work = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def wait(self) -> List[torch.Tensor]:
        if self._work is not None:
            self._work.wait()
            if self._stream is not None:
                torch.cuda.current_stream().wait_stream(self._stream)
            self._work = None
            if self.average:
                self.flat.div_(dist.get_world_size(self.group))
        return self.views()


class ShGradExchange:
    """Exact low-traffic replacement for all-reducing the dense SH gradient.

    For one view dL/dsh[p,m,c] = Y_m(dir(p)) * dL/dRGB[p,c] (rank 1), and every rank knows every view's camera
    centre, so the ranks all-gather the [P,3] factor (rasterizer `sh_rgb_sink`) and each rebuilds
    sum_v Y_m(dir_v(p)) * dL/dRGB_v[p,c] locally with ogs_sh_grad_from_views.  At degree 3 that puts 3*(N-1)
    floats per Gaussian on each GPU's xGMI links instead of the ~2*48*(N-1)/N of a ring all-reduce (21 vs 84 at
    N = 8), and the summation order over views is fixed, so the result does not depend on the collective's
    algorithm."""

    def __init__(self, P: int, sh_coeffs: int, device, group=None):
        self.P, self.M = int(P), int(sh_coeffs)
        self.group = group
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        self.rank = dist.get_rank(group) if self.world > 1 else 0
        self.on = collectives_on(group)
        self.gathered = torch.zeros(self.world, self.P, 3, dtype=torch.float32, device=device)
        self.local = torch.zeros(self.P, 3, dtype=torch.float32, device=device)     # send buffer (not aliased)
        self._work = None
        self._stream = torch.cuda.Stream(device) if torch.device(device).type == "cuda" else None

    def gather_async(self, dL_drgb: torch.Tensor):
        """Start the all-gather of this rank's [P,3] factor on a side stream."""
        if not self.on:
            self.gathered[0].copy_(dL_drgb)
            return
        local = self.local
        local.copy_(dL_drgb)

        def issue():
            return dist.all_gather_into_tensor(self.gathered.view(-1), local.view(-1), group=self.group, async_op=True)
        if self._stream is not None:
            self._stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._stream):
                self._work = issue()
        else:
            self._work = issue()

    def wait(self) -> torch.Tensor:
        """[V,P,3] factors of all views (view v = rank v), ready on the current stream."""
        if self._work is not None:
            self._work.wait()
            if self._stream is not None:
                torch.cuda.current_stream().wait_stream(self._stream)
            self._work = None
        return self.gathered

    def rebuild(self, means3D: torch.Tensor, campos_all: torch.Tensor, sh_degree: int,
                out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """sum over views of dL/dsh, [P,M,3] (HIP kernel; GPU only).  campos_all: [V,3] camera centres, row v =
        the view rank v rendered."""
        from . import _lib
        g = self.wait()
        if not g.is_cuda:
            raise RuntimeError("ShGradExchange.rebuild needs the HIP library (GPU tensors)")
        if out is None:
            out = torch.empty(self.P, self.M, 3, dtype=torch.float32, device=g.device)
        m3 = means3D.detach().to(torch.float32).contiguous()
        cp = campos_all.detach().to(device=g.device, dtype=torch.float32).contiguous()
        if cp.shape != (self.world, 3):
            raise RuntimeError(f"campos_all must be [{self.world},3]")
        _lib.check(_lib.lib().ogs_sh_grad_from_views(self.P, self.world, int(sh_degree), self.M, m3.data_ptr(),
                                                     cp.data_ptr(), g.data_ptr(), out.data_ptr(),
                                                     torch.cuda.current_stream().cuda_stream), "ogs_sh_grad_from_views")
        return out


def densification_stats(grad_means2D: torch.Tensor, radii: torch.Tensor) -> torch.Tensor:
    """This view's [2, P] SUM-reducible statistics: row 0 = ||grad_means2D[:, :2]|| on visible Gaussians, row 1 =
    visibility (0/1).  Append it to the gradient bucket so that it rides in the same all-reduce; MAX(radii) still
    needs its own (tiny) collective -- see reduce_max_radii."""
    vis = radii > 0
    return torch.stack([torch.norm(grad_means2D[:, :2], dim=-1) * vis, vis.to(torch.float32)])


def reduce_max_radii(radii: torch.Tensor, group=None, async_op: bool = False):
    """MAX over views of the screen-space radii (train.py:597).  Returns (tensor, work-or-None)."""
    rmax = radii.to(torch.int32).clone()
    work = None
    if collectives_on(group):
        work = dist.all_reduce(rmax, op=dist.ReduceOp.MAX, group=group, async_op=async_op)
    return rmax, work


def reduce_densification_stats(grad_means2D: torch.Tensor, radii: torch.Tensor, group=None):
    """Per-view statistics that must be reduced separately from the gradients.

    Returns (sum over views of ||grad_means2D[:, :2]|| * visible, visibility count, max radii), i.e. what
    add_densification_stats (scene/gaussian_model.py:512-514) and max_radii2D (train.py:597) would have
    accumulated had the views been processed one after another."""
    vis = radii > 0
    norm = torch.norm(grad_means2D[:, :2], dim=-1) * vis
    cnt = vis.to(torch.float32)
    rmax = radii.to(torch.int32).clone()
    if collectives_on(group):
        packed = torch.stack([norm, cnt])
        dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=group)
        norm, cnt = packed[0], packed[1]
        dist.all_reduce(rmax, op=dist.ReduceOp.MAX, group=group)
    return norm, cnt, rmax


def shard_views(num_views: int, rank: int, world: int) -> List[int]:
    """Views rendered by `rank` for a mini-batch of `num_views` (round-robin, one view per GPU per step
    when num_views == world)."""
    return [v for v in range(num_views) if v % world == rank]


class ShardedAdam:
    """ZeRO-1 style optimizer state sharding for the replicated-Gaussian data-parallel mode (SURVEY.md section 8(e)
    "alternative for >= 2 M points", f2): all parameters live in ONE flat fp32 buffer (the tensors handed out are
    views of it), rank r owns the r-th contiguous slice.  step():

        pack the per-view gradients into a flat buffer
        reduce-scatter (SUM)            -> this rank's slice of the summed gradient      (RCCL; gloo: all-reduce + slice)
        fused Adam on the slice         -> ogs_adam_step, one launch; moments exist for the slice only (memory / N)
        all-gather of the updated slice -> every rank holds the new parameters

    Same bytes on the wire as all-reduce + replicated Adam, 1/N of the optimizer time and state.  Per-parameter
    learning rates as in the reference's param groups (scene/gaussian_model.py:216-224).

    torch.optim.Adam semantics are kept per parameter: a parameter whose ``.grad`` is None on EVERY rank is skipped
    -- value and moments untouched, its own step counter not advanced -- which is what freezes xyz / SH / opacity
    / scaling / rotation from stage 1 on (train.py:431-436 detaches them, zero_grad(set_to_none=True) at
    train.py:610); a parameter that first receives a gradient late starts its bias correction at step 1.  The
    None-mask is agreed across ranks with one tiny MAX all-reduce (a rank whose view missed every Gaussian of a
    tensor contributes zeros)."""

    def __init__(self, named_shapes, lrs, device, group=None, betas=(0.9, 0.999), eps=1e-15):
        self.group = group
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        self.rank = dist.get_rank(group) if self.world > 1 else 0
        self.on = collectives_on(group)
        self.device = device
        self.names = [n for n, _ in named_shapes]
        self.lrs = dict(lrs)
        self.betas, self.eps = betas, eps
        self._layout({n: tuple(s) for n, s in named_shapes})
        self.exp_avg = torch.zeros(self.slice, dtype=torch.float32, device=device)
        self.exp_avg_sq = torch.zeros(self.slice, dtype=torch.float32, device=device)
        self.step_counts = {n: 0 for n in self.names}              # per parameter, like torch's state["step"]

    def _layout(self, shapes: dict):
        """(Re)build the flat layout for the per-tensor `shapes`: offsets, padded length, this rank's slice, the flat
        parameter / gradient buffers and the parameter views.  Moments are NOT touched (the caller installs them)."""
        device = self.device
        self.shapes = dict(shapes)
        self.offsets, off = {}, 0
        for n in self.names:
            self.offsets[n] = off
            off += int(torch.Size(self.shapes[n]).numel())
        self.total = off
        quantum = self.world * 4                                   # slices stay 16-byte aligned
        self.padded = (off + quantum - 1) // quantum * quantum
        self.slice = self.padded // self.world
        self.flat = torch.zeros(self.padded, dtype=torch.float32, device=device)
        self.grad = torch.zeros(self.padded, dtype=torch.float32, device=device)
        self.params = {n: self.flat[self.offsets[n]:self.offsets[n] + int(torch.Size(self.shapes[n]).numel())]
                       .view(self.shapes[n]).requires_grad_(True) for n in self.names}
        lo = self.rank * self.slice
        self.my = (lo, lo + self.slice)
        self.my_grad = torch.zeros(self.slice, dtype=torch.float32, device=device)
        self.my_param = torch.zeros(self.slice, dtype=torch.float32, device=device)

    # ---- densification on the sharded state (scene/gaussian_model.py:357-510) ---------------------
    def _full_moments(self):
        """Both moment vectors over the WHOLE flat layout, on every rank (one all-gather each: densification runs every
        ~100 iterations, train.py:594-605, so the 2 x 4 B x 65 x P bytes are amortised over a hundred steps)."""
        full = []
        for m in (self.exp_avg, self.exp_avg_sq):
            if self.on:
                out = torch.empty(self.padded, dtype=torch.float32, device=m.device)
                dist.all_gather_into_tensor(out, m.contiguous(), group=self.group)
            else:
                out = torch.zeros(self.padded, dtype=torch.float32, device=m.device)
                out[self.my[0]:self.my[1]] = m
            full.append(out)
        return full

    @torch.no_grad()
    def remap_rows(self, src_row: torch.Tensor, kind, extras=()):
        """Apply ONE row map to every parameter and to the optimizer state: new row r of every per-Gaussian tensor =
        old row src_row[r]; rows with kind != 0 are NEW (clones / split children / appended rows): their moments start at
        zero (scene/gaussian_model.py:412-433).  `extras` (statistics tensors) move through the same map.

        Every rank holds all parameters, so the parameters move locally; the moments exist per owned flat slice, so they
        are all-gathered once, moved through the same map and re-sliced at the NEW slice boundaries (the point count
        changed, so every boundary moved).  The map must be identical on all ranks -- it is when it derives from
        all-reduced statistics (dp.reduce_densification_stats) and broadcast samples (densify.densify_and_prune does
        both).  Returns ({name: new parameter view}, [new extras])."""
        from .densify import gather_rows
        m_full, v_full = self._full_moments()
        n_out = int(src_row.shape[0])
        tensors, zero_new = [], set()
        for n in self.names:
            a, cnt = self.offsets[n], int(torch.Size(self.shapes[n]).numel())
            tensors.append(self.flat[a:a + cnt].view(self.shapes[n]))
            for full in (m_full, v_full):
                zero_new.add(len(tensors))
                tensors.append(full[a:a + cnt].view(self.shapes[n]))
        tensors.extend(extras)
        outs = gather_rows(tensors, src_row, kind, zero_new)
        self._layout({n: (n_out,) + tuple(self.shapes[n][1:]) for n in self.names})
        new_m = torch.zeros(self.padded, dtype=torch.float32, device=self.flat.device)
        new_v = torch.zeros(self.padded, dtype=torch.float32, device=self.flat.device)
        for i, n in enumerate(self.names):
            a, cnt = self.offsets[n], int(torch.Size(self.shapes[n]).numel())
            self.flat[a:a + cnt].copy_(outs[3 * i].reshape(-1))
            new_m[a:a + cnt].copy_(outs[3 * i + 1].reshape(-1))
            new_v[a:a + cnt].copy_(outs[3 * i + 2].reshape(-1))
        lo, hi = self.my
        self.exp_avg, self.exp_avg_sq = new_m[lo:hi].clone(), new_v[lo:hi].clone()
        return dict(self.params), list(outs[3 * len(self.names):])

    @torch.no_grad()
    def replace_tensor(self, name: str, tensor: torch.Tensor):
        """scene/gaussian_model.py:357-372 (`replace_tensor_to_optimizer`, used by reset_opacity): new values for one
        parameter, its moments zeroed -- on the part of the owned slice that overlaps it."""
        self.params[name].copy_(tensor.reshape(self.shapes[name]))
        lo, hi = self.my
        a = max(self.offsets[name], lo)
        b = min(self.offsets[name] + self.params[name].numel(), hi)
        if b > a:
            self.exp_avg[a - lo:b - lo].zero_()
            self.exp_avg_sq[a - lo:b - lo].zero_()
        return {name: self.params[name]}

    def load(self, tensors: dict):
        """initial values (identical on every rank)"""
        with torch.no_grad():
            for n in self.names:
                self.params[n].copy_(tensors[n])

    @torch.no_grad()
    def step(self):
        from . import _lib
        from ._lib import OgsAdamTensor
        has = torch.tensor([0.0 if self.params[n].grad is None else 1.0 for n in self.names], dtype=torch.float32,
                           device=self.flat.device)
        if self.on:
            dist.all_reduce(has, op=dist.ReduceOp.MAX, group=self.group)
        active = {n for n, h in zip(self.names, has.tolist()) if h > 0}
        for n in self.names:
            g = self.params[n].grad
            seg = self.grad[self.offsets[n]:self.offsets[n] + self.params[n].numel()]
            if g is None:
                seg.zero_()
            else:
                seg.copy_(g.reshape(-1))
        if not active:
            return
        lo, hi = self.my
        if self.on:
            if dist.get_backend(self.group) == "nccl":
                dist.reduce_scatter_tensor(self.my_grad, self.grad, op=dist.ReduceOp.SUM, group=self.group)
            else:                                                   # gloo has no reduce-scatter
                dist.all_reduce(self.grad, op=dist.ReduceOp.SUM, group=self.group)
                self.my_grad.copy_(self.grad[lo:hi])
        else:
            self.my_grad.copy_(self.grad[lo:hi])
        self.my_param.copy_(self.flat[lo:hi])
        descs = []
        for n in self.names:                                        # this rank's part of every stepped parameter
            if n not in active:
                continue
            self.step_counts[n] += 1
            a = max(self.offsets[n], lo)
            b = min(self.offsets[n] + self.params[n].numel(), hi)
            if b <= a:
                continue
            d = OgsAdamTensor()
            d.param = self.my_param.data_ptr() + 4 * (a - lo)
            d.grad = self.my_grad.data_ptr() + 4 * (a - lo)
            d.exp_avg = self.exp_avg.data_ptr() + 4 * (a - lo)
            d.exp_avg_sq = self.exp_avg_sq.data_ptr() + 4 * (a - lo)
            d.numel, d.lr, d.step = b - a, float(self.lrs[n]), self.step_counts[n]
            descs.append(d)
        for i in range(0, len(descs), 16):                          # OGS_ADAM_MAX_TENSORS per launch
            chunk = descs[i:i + 16]
            arr = (OgsAdamTensor * len(chunk))(*chunk)
            _lib.check(_lib.lib().ogs_adam_step(arr, len(chunk), float(self.betas[0]), float(self.betas[1]), float(self.eps),
                                                torch.cuda.current_stream().cuda_stream), "ogs_adam_step")
        if self.on:
            dist.all_gather_into_tensor(self.flat, self.my_param, group=self.group)
        else:
            self.flat[lo:hi].copy_(self.my_param)
        for n in self.names:
            self.params[n].grad = None
