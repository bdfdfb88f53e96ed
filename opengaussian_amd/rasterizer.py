"""Drop-in Python surface of the reference's rasterizer package.

Mirrors ``ashawkey_diff_gaussian_rasterization`` as the reference uses it
(/root/reference/gaussian_renderer/__init__.py:15,55-70,104-112; utils/sam_refinement_utils.py:21,347-402):
``GaussianRasterizationSettings`` (12-field NamedTuple), ``GaussianRasterizer(raster_settings=...)`` called with
keyword args ``means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp`` and
returning ``(color[3,H,W], radii[P] int32, depth[1,H,W], alpha[1,H,W])``; ``markVisible(positions)``.

Underneath is the C ABI of include/ogs_raster.h (HIP kernels for gfx950) bound with ctypes; torch only
provides device memory, the current stream and autograd.  There is no CPU path: non-GPU tensors raise.

Extension beyond the reference API: ``colors_precomp`` may carry 3, 6, 9 or 12
channels; one pass then bins/sorts once and blends all channels (``color`` is [C,H,W]).
"""
from __future__ import annotations

import ctypes as C
import functools
import os
from typing import NamedTuple, Optional

import torch
from torch import nn

from . import _lib
from ._lib import OgsRasterBwdArgs, OgsRasterFwdArgs, check, ptr

SUPPORTED_CHANNELS = (3, 6, 9, 12)
BACKWARD_CHANNELS = (3, 6, 9)          # the gradient record holds C + 7 <= 16 slots
# Passes with at most this many Gaussians (ungrouped) take the two-launch tiny path (ogs_raster_forward_tiny): the
# SAM refiner's P = 1 footprint renders and other very small subsets.  0 disables it (the parity tests that export the
# binning state of small scenes do).  Never larger than ogs_raster_tiny_max_points().
TINY_MAX_P = 256
# diagnostics (scripts/diag_repeat.py): when set to a list, every backward appends its raw gradient-record buffer
_DEBUG_KEEP_BWD_TMP = None
# Capacity hint of the sync-free render phase: (P, W, H) -> num_rendered of the most recent passes at that size.
# The cameras of a training run take turns, so one pass's count does not predict the next one's: the
# capacity is 1.25x the MAXIMUM over the last _HINT_WINDOW passes -- after one sweep over the views it covers all of
# them; a scene that shrinks (pruning) lets the old maxima age out of the window.
_LAST_NUM_RENDERED: dict = {}
_HINT_WINDOW = 32
# how the render phase of every forward was sized: "blocking" (first pass at a size, grouped / debug passes: the
# reference's 4-byte read-back), "deferred" (sync-free, capacity from the hint), "overflow" (deferred result
# discarded, render phase redone with exact buffers)
PASS_STATS = {"blocking": 0, "deferred": 0, "overflow": 0, "tiny": 0, "tiny_rerendered_for_backward": 0, "reblend": 0}


def _hint_capacity(key):
    hist = _LAST_NUM_RENDERED.get(key)
    if not hist:
        return None
    return int(max(hist) * 1.25) + 4096


def _hint_record(key, D):
    from collections import deque
    if len(_LAST_NUM_RENDERED) > 256:      # subset renders come in many sizes: keep the hint table small
        _LAST_NUM_RENDERED.clear()
    hist = _LAST_NUM_RENDERED.get(key)
    if hist is None:
        hist = _LAST_NUM_RENDERED[key] = deque(maxlen=_HINT_WINDOW)
    hist.append(int(D))


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


_READBACK: dict = {}


def _readback_slot(dev):
    """One pinned int32 + one event per device, reused by every pass: the forward waits for the copy before it
    returns, so two passes never have a read-back in flight at the same time (pinned allocations are slow)."""
    slot = _READBACK.get(dev)
    if slot is None:
        slot = (torch.empty(1, dtype=torch.int32, pin_memory=True), torch.cuda.Event())
        _READBACK[dev] = slot
    return slot


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _f32c(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    """contiguous fp32 view/copy (inputs may be sliced / boolean-indexed views, :133,204-212).  Tensors that already
    are contiguous fp32 pass through untouched: a tiny pass is all host overhead, every torch call counts."""
    if t is None or t.numel() == 0:
        return None
    if t.dtype is torch.float32 and t.is_contiguous():
        return t.detach() if t.requires_grad else t
    return t.detach().to(torch.float32).contiguous()


_TINY_LIMIT = None
_TINY_SCRATCH: dict = {}


def _tiny_limit() -> int:
    global _TINY_LIMIT
    if _TINY_LIMIT is None:
        _TINY_LIMIT = int(_lib.lib().ogs_raster_tiny_max_points())
    return min(TINY_MAX_P, _TINY_LIMIT)


def _tiny_scratch(dev, stream: int, lib):
    """geom_buffer + geom_tmp of a tiny pass (<= 256 Gaussians, 12 channels: a few tens of KB), one pair per (device,
    stream): both are dead when the two launches of the pass have run, and passes on one stream are ordered."""
    key = (dev, stream)
    sc = _TINY_SCRATCH.get(key)
    if sc is None:
        n = int(lib.ogs_raster_tiny_max_points())
        sc = (torch.empty(int(lib.ogs_raster_geom_bytes(n, 12)), dtype=torch.uint8, device=dev),
              torch.empty(int(lib.ogs_raster_geom_tmp_bytes(n)), dtype=torch.uint8, device=dev))
        _TINY_SCRATCH[key] = sc
    return sc


_BG_TILED: dict = {}


def _tiled_bg(bg: torch.Tensor, Cn: int) -> torch.Tensor:
    """the 3-wide background tiled over a fused pass's channels; cached per (storage, version): training passes the same
    background tensor every iteration and a `repeat` is a launch plus an allocation on a path that is all host time"""
    key = (bg.data_ptr(), bg._version, Cn, bg.device)
    hit = _BG_TILED.get(key)
    if hit is None:
        if len(_BG_TILED) > 16:
            _BG_TILED.clear()
        hit = (bg, bg.repeat(Cn // 3))        # the source is held too: its address cannot be recycled while cached
        _BG_TILED[key] = hit
    return hit[1]


def _require_gpu(t: torch.Tensor, name: str):
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live on the GPU (got {t.device}); the MI355X rasterizer has no CPU path")


# Binning mode of the passes issued from this process (OgsRasterFwdArgs.full_binning, include/ogs_raster.h).  False (default): the
# (Gaussian, tile) pairs that cannot reach a pixel of their tile leave the list in the first pass of the tile sort.  True: the
# reference's full list is kept -- what tests/helpers.py::hip_export_binning compares with the oracle's binning entry by entry.
FULL_BINNING = os.environ.get("OGS_FULL_BINNING", "0") == "1"      # the environment variable is for A-B timing runs


class full_binning:
    """Context manager: passes issued inside keep the reference's full (Gaussian, tile) list (diagnostics / parity tests)."""

    def __enter__(self):
        global FULL_BINNING
        self._prev, FULL_BINNING = FULL_BINNING, True
        return self

    def __exit__(self, *exc):
        global FULL_BINNING
        FULL_BINNING = self._prev
        return False


def _fwd_args(rs, P, Cn, m3, shs, cols, opac, scl, rot, cov, bg, view, proj, campos, color, depth, alpha, radii, group_ids, G):
    a = OgsRasterFwdArgs()
    a.P, a.W, a.H, a.C = P, int(rs.image_width), int(rs.image_height), Cn
    a.sh_degree = int(rs.sh_degree)
    a.sh_coeffs = 0 if shs is None else int(shs.shape[1])
    a.tanfovx, a.tanfovy, a.scale_modifier = float(rs.tanfovx), float(rs.tanfovy), float(rs.scale_modifier)
    a.prefiltered, a.debug = int(bool(rs.prefiltered)), int(bool(rs.debug))
    a.bg, a.means3D, a.colors_precomp, a.shs, a.opacities = ptr(bg), ptr(m3), ptr(cols), ptr(shs), ptr(opac)
    a.scales, a.rotations, a.cov3D_precomp = ptr(scl), ptr(rot), ptr(cov)
    a.viewmatrix, a.projmatrix, a.campos = ptr(view), ptr(proj), ptr(campos)
    a.out_color, a.out_depth, a.out_alpha, a.radii = ptr(color), ptr(depth), ptr(alpha), ptr(radii)
    a.group_ids, a.num_groups = ptr(group_ids), G
    a.full_binning = int(FULL_BINNING)
    return a


# scratch sizes are pure functions of the shape: asked once per shape, not six ctypes calls per pass (a 100 k-Gaussian step is
# paced by the host)
@functools.lru_cache(maxsize=256)
def _geom_sizes(P, W, H, Cn, G):
    lib = _lib.lib()
    return (int(lib.ogs_raster_geom_bytes(P, Cn)), int(lib.ogs_raster_geom_tmp_bytes(P)),
            int(lib.ogs_raster_image_bytes_grouped(W, H, G)))


@functools.lru_cache(maxsize=1024)
def _render_sizes(count, W, H, Cn):
    lib = _lib.lib()
    return (int(lib.ogs_raster_binning_tmp_bytes(count, W, H)), int(lib.ogs_raster_sorted_bytes(count, Cn)),
            int(lib.ogs_raster_quad_list_bytes(count)))


def _streaming_render(a: OgsRasterFwdArgs, dev, lib, debug: bool):
    """The two-phase forward (geometry -> num_rendered -> render) on the buffers `a` already points to for inputs
    and outputs; allocates and returns (geom, image, point_list, sorted_rec, quad_list, num_rendered)."""
    P, W, H, Cn, G = int(a.P), int(a.W), int(a.H), int(a.C), max(int(a.num_groups), 1)
    u8 = lambda n: torch.empty(int(n), dtype=torch.uint8, device=dev)
    gb, gtb, ib = _geom_sizes(P, W, H, Cn, G)
    geom, geom_tmp, image = u8(gb), u8(gtb), u8(ib)
    a.geom_buffer, a.geom_tmp, a.image_buffer = ptr(geom), ptr(geom_tmp), ptr(image)
    stream = _stream()

    def alloc_render(count):
        btb, srb, qlb = _render_sizes(count, W, H, Cn)
        pl = torch.empty(max(count, 1), dtype=torch.int32, device=dev)
        bt, sr, ql = u8(btb), u8(srb), u8(qlb)
        a.point_list, a.binning_tmp, a.sorted_rec, a.quad_list = ptr(pl), ptr(bt), ptr(sr), ptr(ql)
        return pl, bt, sr, ql

    key = (P, W, H, G)
    # grouped passes: num_rendered follows the group ids of the call (cluster chunk, leaf range), which change
    # from call to call -> sized by the blocking read-back, like subset passes (their P is new every time)
    cap = None if (G > 1 or debug) else _hint_capacity(key)
    if cap is None:
        PASS_STATS["blocking"] += 1
        # first pass at this size: blocking 4-byte read-back of num_rendered (what the reference does every time)
        n = C.c_int64(0)
        check(lib.ogs_raster_forward_geometry(C.byref(a), stream, C.byref(n)), "ogs_raster_forward_geometry")
        D = int(n.value)
        point_list, bin_tmp, sorted_rec, quad_list = alloc_render(D)
        check(lib.ogs_raster_forward_render(C.byref(a), D, stream), "ogs_raster_forward_render")
    else:
        # steady state: no GPU idle gap.  The render phase is enqueued for a capacity derived from the recent
        # passes at this size; the true count arrives through an async pinned copy and is only WAITED for after
        # everything is queued.  Overflow (scene changed a lot) -> redo the render phase with exact buffers.
        check(lib.ogs_raster_forward_geometry(C.byref(a), stream, None), "ogs_raster_forward_geometry")
        pinned, ev = _readback_slot(dev)
        check(lib.ogs_raster_read_num_rendered_async(C.byref(a), stream, pinned.data_ptr()),
              "ogs_raster_read_num_rendered_async")
        ev.record()
        PASS_STATS["deferred"] += 1
        point_list, bin_tmp, sorted_rec, quad_list = alloc_render(cap)
        check(lib.ogs_raster_forward_render_deferred(C.byref(a), cap, stream), "ogs_raster_forward_render_deferred")
        ev.synchronize()
        # the geometry phase of THIS pass (its depth sort) and every earlier launch have finished: anything a kernel reported
        # through the sticky status word (a one-launch radix pass whose bounded look-back wait ran out) raises here
        check(lib.ogs_check_async_status(), "ogs_check_async_status")
        D = int(pinned.item()) & 0xFFFFFFFF
        if D > cap:
            PASS_STATS["overflow"] += 1
            point_list, bin_tmp, sorted_rec, quad_list = alloc_render(D)
            check(lib.ogs_raster_forward_render(C.byref(a), D, stream), "ogs_raster_forward_render")
    _hint_record(key, D)
    return geom, image, point_list, sorted_rec, quad_list, D


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                raster_settings: GaussianRasterizationSettings, geom_channels: int = 0, sh_rgb_sink=None,
                group_ids=None, num_groups: int = 1, keep_sink=None):
        rs = raster_settings
        ctx.geom_channels = int(geom_channels)
        ctx.sh_rgb_sink = sh_rgb_sink
        # outputs no loss touches (depth in every reference loss, alpha without RGBA data) must reach backward as
        # None, not as zero images: the kernels compile the corresponding terms out
        ctx.set_materialize_grads(False)
        G = max(int(num_groups), 1)
        ctx.num_groups = G
        if G > 1:
            if group_ids is None or group_ids.shape[0] != means3D.shape[0]:
                raise RuntimeError("a grouped pass needs group_ids [P]")
            group_ids = group_ids.detach().to(device=means3D.device, dtype=torch.int32).contiguous()
        else:
            group_ids = None
        _require_gpu(means3D, "means3D")
        dev = means3D.device
        lib = _lib.lib()
        P = int(means3D.shape[0])
        H, W = int(rs.image_height), int(rs.image_width)
        if means3D.dim() != 2 or means3D.shape[1] != 3:
            raise RuntimeError("means3D must have dimensions (num_points, 3)")

        m3 = _f32c(means3D)
        shs = _f32c(sh)
        cols = _f32c(colors_precomp)
        opac = _f32c(opacities)
        scl = _f32c(scales)
        rot = _f32c(rotations)
        cov = _f32c(cov3Ds_precomp)
        if shs is not None and cols is not None:
            Cn = 3 + int(cols.shape[1])        # fused pass: SH -> channels 0..2, colors_precomp -> the rest
        else:
            Cn = 3 if cols is None else int(cols.shape[1])
        if Cn not in SUPPORTED_CHANNELS:
            raise RuntimeError(f"the blended channel count must be 3, 6, 9 or 12, got {Cn}")
        if Cn not in BACKWARD_CHANNELS and any(ctx.needs_input_grad[:8]):
            # fail here, not in the middle of loss.backward()
            raise RuntimeError(f"a {Cn}-channel pass is forward-only (backward supports {BACKWARD_CHANNELS} channels): "
                               "detach the inputs or split the pass")
        bg = _f32c(rs.bg.to(dev))
        if bg is None or bg.numel() != Cn:
            if bg is not None and bg.numel() == 3 and Cn > 3:
                # the facade's bg is 3-wide and the reference applies it to EVERY 3-channel pass (RGB, feat[:, :3],
                # feat[:, 3:6], gaussian_renderer/__init__.py:55-70,129-151): tile it over the fused channels
                bg = _tiled_bg(bg, Cn)
            else:
                raise RuntimeError(f"bg must have {Cn} entries")
        view = _f32c(rs.viewmatrix.to(dev))
        proj = _f32c(rs.projmatrix.to(dev))
        campos = _f32c(rs.campos.to(dev))

        # P > 0: the kernels write every pixel of every tile and every radius, so no fill pass is needed
        alloc = torch.zeros if P == 0 else torch.empty
        lead = (G,) if G > 1 else ()           # grouped pass: one image per group
        # three separate tensors (not views of one allocation): callers may modify an output in place
        color = alloc(*lead, Cn, H, W, dtype=torch.float32, device=dev)
        depth = alloc(*lead, 1, H, W, dtype=torch.float32, device=dev)
        alpha = alloc(*lead, 1, H, W, dtype=torch.float32, device=dev)
        radii = alloc(P, dtype=torch.int32, device=dev)
        ctx.raster_settings = rs
        ctx.P, ctx.Cn, ctx.num_rendered, ctx.tiny = P, Cn, 0, False
        ctx.full_binning = bool(FULL_BINNING)
        if P == 0:
            # reference behaviour: zero images, nothing launched (SURVEY.md section 8(b) "Errors")
            ctx.save_for_backward()
            ctx.mark_non_differentiable(radii)
            return color, radii, depth, alpha

        a = _fwd_args(rs, P, Cn, m3, shs, cols, opac, scl, rot, cov, bg, view, proj, campos, color, depth, alpha, radii,
                      group_ids, G)

        if G == 1 and not rs.debug and P <= _tiny_limit():
            # tiny pass: two launches, no read-back, nothing kept -- backward() re-renders through the streaming path
            stream = _stream()
            geom, geom_tmp = _tiny_scratch(dev, stream, lib)
            a.geom_buffer, a.geom_tmp = geom.data_ptr(), geom_tmp.data_ptr()
            check(lib.ogs_raster_forward_tiny(C.byref(a), stream), "ogs_raster_forward_tiny")
            PASS_STATS["tiny"] += 1
            ctx.tiny, ctx.num_rendered = True, -1
            ctx.save_for_backward(m3, shs, cols, opac, scl, rot, cov, bg, view, proj, campos, radii, alpha)
            ctx.mark_non_differentiable(radii)
            return color, radii, depth, alpha

        geom, image, point_list, sorted_rec, quad_list, D = _streaming_render(a, dev, lib, rs.debug)
        ctx.num_rendered = D
        ctx.save_for_backward(m3, shs, cols, opac, scl, rot, cov, bg, view, proj, campos, radii, alpha, geom, image,
                              point_list, sorted_rec, quad_list)
        ctx.mark_non_differentiable(radii)
        if keep_sink is not None and G == 1 and D > 0:
            # frozen-geometry cache (KeptPasses below): what a re-blend of this pass needs
            keep_sink.append({"P": P, "W": W, "H": H, "Cn": Cn, "fused": shs is not None and cols is not None, "D": D,
                              "image": image, "sorted_rec": sorted_rec, "quad_list": quad_list, "radii": radii,
                              "full_binning": bool(FULL_BINNING)})
        return color, radii, depth, alpha

    @staticmethod
    def backward(ctx, grad_color, grad_radii, grad_depth, grad_alpha):
        rs = ctx.raster_settings
        P, Cn = ctx.P, ctx.Cn
        if P == 0:
            return (None,) * 14
        lib = _lib.lib()
        H, W = int(rs.image_height), int(rs.image_width)
        scratch = None
        if ctx.tiny:
            # a tiny pass keeps nothing: re-render through the streaming path (scratch outputs) to obtain the binning
            # and blend state the backward kernels read.  Rare: the refiner's footprint renders never call backward.
            (m3, shs, cols, opac, scl, rot, cov, bg, view, proj, campos, radii, alpha) = ctx.saved_tensors
            dev = m3.device
            e = lambda *sh: torch.empty(*sh, dtype=torch.float32, device=dev)
            # scratch outputs: held in `scratch` until the re-render has been enqueued AND everything else of this
            # backward has been allocated -- `a` only carries raw pointers, a tensor freed here would be handed out
            # again by the caching allocator while the kernels still write through the old pointer
            scratch = (e(Cn, H, W), e(1, H, W), e(1, H, W), torch.empty(P, dtype=torch.int32, device=dev))
            a = _fwd_args(rs, P, Cn, m3, shs, cols, opac, scl, rot, cov, bg, view, proj, campos, *scratch, None, 1)
            geom, image, point_list, sorted_rec, quad_list, D = _streaming_render(a, dev, lib, False)
            ctx.num_rendered = D
            PASS_STATS["tiny_rerendered_for_backward"] += 1
        else:
            (m3, shs, cols, opac, scl, rot, cov, bg, view, proj, campos, radii, alpha, geom, image,
             point_list, sorted_rec, quad_list) = ctx.saved_tensors
        dev = m3.device
        need = ctx.needs_input_grad   # means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3D
        z = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)

        g_m3 = z(P, 3) if need[0] else None
        g_m2 = z(P, 3) if need[1] else None
        # data-parallel mode (dp.ShGradExchange): hand out the rank-1 factor [P,3] instead of the dense dL/dsh
        sink = ctx.sh_rgb_sink if (need[2] and shs is not None) else None
        g_sh = z(*shs.shape) if (need[2] and shs is not None and sink is None) else None
        g_sh_rgb = z(P, 3) if sink is not None else None
        g_col = z(*cols.shape) if (need[3] and cols is not None) else None
        g_op = z(P, 1) if need[4] else None
        g_scl = z(P, 3) if (need[5] and scl is not None) else None
        g_rot = z(P, 4) if (need[6] and rot is not None) else None
        g_cov = z(P, 6) if (need[7] and cov is not None) else None

        gc = _f32c(grad_color)
        if gc is None:
            gc = torch.zeros(*((ctx.num_groups,) if ctx.num_groups > 1 else ()), Cn, H, W, dtype=torch.float32, device=dev)
        gd = _f32c(grad_depth)
        ga = _f32c(grad_alpha)
        bwd_tmp = torch.empty(int(lib.ogs_raster_backward_tmp_bytes(P)), dtype=torch.uint8, device=dev)

        b = OgsRasterBwdArgs()
        b.P, b.W, b.H, b.C = P, W, H, Cn
        b.sh_degree = int(rs.sh_degree)
        b.sh_coeffs = 0 if shs is None else int(shs.shape[1])
        b.tanfovx, b.tanfovy, b.scale_modifier = float(rs.tanfovx), float(rs.tanfovy), float(rs.scale_modifier)
        b.debug = int(bool(rs.debug))
        b.num_rendered = int(ctx.num_rendered)
        b.geom_channels = int(ctx.geom_channels)
        b.num_groups = int(ctx.num_groups)
        b.sorted_rec, b.quad_list = ptr(sorted_rec), ptr(quad_list)
        b.bg, b.means3D, b.colors_precomp, b.shs, b.opacities = ptr(bg), ptr(m3), ptr(cols), ptr(shs), ptr(opac)
        b.scales, b.rotations, b.cov3D_precomp = ptr(scl), ptr(rot), ptr(cov)
        b.viewmatrix, b.projmatrix, b.campos = ptr(view), ptr(proj), ptr(campos)
        b.radii, b.out_alpha = ptr(radii), ptr(alpha)
        b.dL_dcolor, b.dL_ddepth, b.dL_dalpha = ptr(gc), ptr(gd), ptr(ga)
        b.geom_buffer, b.image_buffer, b.point_list, b.bwd_tmp = ptr(geom), ptr(image), ptr(point_list), ptr(bwd_tmp)
        b.dL_dmeans2D, b.dL_dcolors, b.dL_dopacity, b.dL_dmeans3D = ptr(g_m2), ptr(g_col), ptr(g_op), ptr(g_m3)
        b.dL_dcov3D, b.dL_dsh, b.dL_dscales, b.dL_drotations = ptr(g_cov), ptr(g_sh), ptr(g_scl), ptr(g_rot)
        b.dL_dsh_rgb = ptr(g_sh_rgb)
        check(lib.ogs_raster_backward(C.byref(b), _stream()), "ogs_raster_backward")
        scratch = None          # (tiny pass re-render) stream-ordered: safe to recycle once the launches are queued
        if _DEBUG_KEEP_BWD_TMP is not None:
            _DEBUG_KEEP_BWD_TMP.append(bwd_tmp)
        if sink is not None:
            sink.append(g_sh_rgb)
        return g_m3, g_m2, g_sh, g_col, g_op, g_scl, g_rot, g_cov, None, None, None, None, None, None


# ---- kept passes: the frozen-geometry cache ------------------------------------------------------------------------------
# From stage 1 on the reference trains `_ins_feat` alone: train.py:431-436 detaches every other Gaussian parameter, and the
# stage-1 calls render without the random footprint rescale (train.py:346-350).  For a given camera every such call rebuilds,
# entry for entry, the binning state of the previous one -- preprocess, both sorts, the duplication and pack are spent on
# reproducing bytes that already exist.  With 288 GB of HBM they can simply stay: a finished pass whose caller vouches for its
# geometry (`frozen_key`, renderer.py builds it from the parameters' storages and version counters) leaves image_buffer, the
# packed records, their quadrant streams and the radii behind (0.35 GB for a 1080p view of 1 M Gaussians, 0.08 GB for a
# ScanNet-class view whose tiles stop at 15 % of their lists: compacted to what the tiles packed), and the next pass with the same
# key is ogs_raster_forward_reblend: ONE launch, the forward blend over the kept streams with the records' feature channels taken
# from the caller's current tensor (the kept records are never written).
# Images, depth, alpha, radii and the feature gradients are bit for bit those of a full pass (tests/test_14_kept_pass_gpu.py).
class _KeptPass:
    __slots__ = ("key", "generation", "holds", "P", "W", "H", "Cn", "E", "fused", "D", "image", "sorted_rec", "quad_list", "radii",
                 "nbytes", "hits", "full_binning")


class KeptPasses:
    """One kept pass per SLOT (a camera); a slot whose key changed (parameters stepped, replaced, another image size ...) is
    dropped on lookup.  Admission stops at the byte budget -- nothing is evicted to make room: the training loop draws its
    cameras from a shuffled stack (train.py:296-299), the access pattern on which least-recently-used eviction hits nothing
    once the working set exceeds the budget, while a fixed resident subset keeps hitting for its share of the views."""

    def __init__(self, budget_bytes=None):
        self.budget_bytes = budget_bytes          # None: OGS_KEPT_PASSES_GB, default min(device / 4, free / 2) at first use
        self.slots: dict = {}
        self.nbytes = 0
        self.stats = {"hits": 0, "misses": 0, "stale": 0, "admitted": 0, "rejected_budget": 0, "dropped_old_generation": 0,
                      "rejected_out_of_memory": 0}

    def _budget(self, dev) -> int:
        if self.budget_bytes is None:
            env = os.environ.get("OGS_KEPT_PASSES_GB")
            if env is not None:
                self.budget_bytes = int(float(env) * (1 << 30))
            else:
                # a quarter of the device, but never more than half of what is free when the first pass asks: the cache must not
                # be what pushes the training process into an out-of-memory error
                free, total = torch.cuda.mem_get_info(dev)
                self.budget_bytes = int(min(total // 4, free // 2))
        return self.budget_bytes

    def enabled(self, dev) -> bool:
        return self._budget(dev) > 0

    def lookup(self, slot, key):
        e = self.slots.get(slot)
        if e is None:
            self.stats["misses"] += 1
            return None
        if e.key != key:
            self.drop(slot)
            self.stats["stale"] += 1
            self.stats["misses"] += 1
            return None
        e.hits += 1
        self.stats["hits"] += 1
        return e

    def drop(self, slot):
        e = self.slots.pop(slot, None)
        if e is not None:
            self.nbytes -= e.nbytes

    def clear(self):
        self.slots.clear()
        self.nbytes = 0

    def admit(self, slot, key, holds, kept: dict, generation=None) -> bool:
        """kept: what _RasterizeGaussians.forward left in its keep_sink.  The record array and the quadrant streams of the pass
        were sized for a CAPACITY (1.25 x the recent maximum of num_rendered, and the reachable pairs are about half of it) and
        laid out by the sorted list's tile ranges: the records and stream entries the tiles actually PACKED are moved into
        exact-size buffers (ogs_raster_compact_kept) -- one 8-byte read-back per admitted view.
        generation: what all passes over ONE state of the model share (renderer.py: the parameters' part of the key).  Entries
        of another generation can never hit again -- the model moved on, or another scene was loaded -- but cameras that are
        not revisited would hold their bytes (and the old parameters) forever: when the budget is short they go first."""
        dev = kept["image"].device
        lib = _lib.lib()
        W, H, Cn = kept["W"], kept["H"], kept["Cn"]
        tiles = ((W + 15) // 16) * ((H + 15) // 16)
        image = kept["image"]
        # image_buffer layout (include/ogs_raster.h, ogs_raster_compact_kept): ranges | n_contrib | counters | final_T | tile_order
        up = lambda n: (n + 255) // 256 * 256
        o_cnt = up(tiles * 8) + up(W * H * 4)
        packed = image[o_cnt:o_cnt + tiles * 20].view(torch.int32).view(tiles, 5)[:, 4].to(torch.int64)   # records per tile
        ends = torch.cumsum(packed, 0)
        used = int(ends[-1].item())                           # the one read-back of an admission
        if used <= 0:
            return False
        srb, qlb = int(lib.ogs_raster_sorted_bytes(used, Cn)), int(lib.ogs_raster_quad_list_bytes(used))
        nbytes = srb + qlb + image.numel() + kept["radii"].numel() * 4
        if self.nbytes + nbytes > self._budget(dev) and generation is not None:
            for old_slot in [sl for sl, en in self.slots.items() if en.generation != generation]:
                self.drop(old_slot)
                self.stats["dropped_old_generation"] += 1
        if self.nbytes + nbytes > self._budget(dev):
            self.stats["rejected_budget"] += 1
            return False
        e = _KeptPass()
        e.key, e.holds, e.generation = key, holds, generation
        e.P, e.W, e.H, e.Cn, e.fused, e.D = kept["P"], W, H, Cn, bool(kept["fused"]), used
        e.E = Cn - 3 if e.fused else Cn
        e.full_binning = kept["full_binning"]
        e.radii = kept["radii"]
        try:
            # the pass laid its records and quadrant streams out by the ranges of the sorted list, sized for a capacity; what is
            # kept is what the tiles PACKED (all a re-blend reads), re-laid out by the exclusive scan of those counts
            e.image = image.clone()
            new_ranges = torch.stack((ends - packed, ends), 1).to(torch.int32).contiguous()
            e.image[: tiles * 8] = new_ranges.view(-1).view(torch.uint8)
            e.sorted_rec = torch.empty(srb, dtype=torch.uint8, device=dev)
            e.quad_list = torch.empty(qlb, dtype=torch.uint8, device=dev)
        except torch.cuda.OutOfMemoryError:
            # the budget is a share of the device, not of what the training process left free: keeping a pass is optional
            self.stats["rejected_out_of_memory"] += 1
            return False
        check(lib.ogs_raster_compact_kept(W, H, Cn, ptr(image), ptr(kept["sorted_rec"]), ptr(kept["quad_list"]), ptr(e.image),
                                          ptr(e.sorted_rec), ptr(e.quad_list), _stream()), "ogs_raster_compact_kept")
        e.nbytes, e.hits = nbytes, 0
        self.drop(slot)
        self.slots[slot] = e
        self.nbytes += nbytes
        self.stats["admitted"] += 1
        return True


KEPT_PASSES = KeptPasses()


class KeptImages:
    """Outputs of a pass in which NOTHING trains: stage >= 1 renders RGB from frozen parameters on every call (the unfused RGB pass
    of the rescaled stage-2.1 calls, gaussian_renderer/__init__.py:104-112, and every stage-2.2 call, train.py:339-341) -- for a
    given camera, model state and background tensor the same image, depth, alpha and radii, bit for bit, every time.  45 MB per
    1080p view.  Same slots / keys / generations / budget rules as KeptPasses; a hit hands out CLONES (callers may modify what
    they get)."""

    def __init__(self, budget_bytes=None):
        self.budget_bytes = budget_bytes          # None: the budget of KEPT_PASSES (counted separately)
        self.slots: dict = {}
        self.nbytes = 0
        self.stats = {"hits": 0, "misses": 0, "stale": 0, "admitted": 0, "rejected_budget": 0}

    def _budget(self, dev) -> int:
        return KEPT_PASSES._budget(dev) if self.budget_bytes is None else self.budget_bytes

    def lookup(self, slot, key):
        e = self.slots.get(slot)
        if e is None or e[0] != key:
            if e is not None:
                self.drop(slot)
                self.stats["stale"] += 1
            self.stats["misses"] += 1
            return None
        self.stats["hits"] += 1
        return tuple(t.clone() for t in e[3])

    def drop(self, slot):
        e = self.slots.pop(slot, None)
        if e is not None:
            self.nbytes -= e[4]

    def clear(self):
        self.slots.clear()
        self.nbytes = 0

    def admit(self, slot, key, holds, outputs, generation=None) -> bool:
        dev = outputs[0].device
        nbytes = sum(t.numel() * t.element_size() for t in outputs)
        if self.nbytes + nbytes > self._budget(dev) and generation is not None:
            for old_slot in [sl for sl, en in self.slots.items() if en[1] != generation]:
                self.drop(old_slot)
        if self.nbytes + nbytes > self._budget(dev):
            self.stats["rejected_budget"] += 1
            return False
        try:
            kept = tuple(t.detach().clone() for t in outputs)
        except torch.cuda.OutOfMemoryError:
            return False
        self.drop(slot)
        self.slots[slot] = (key, generation, holds, kept, nbytes)
        self.nbytes += nbytes
        self.stats["admitted"] += 1
        return True


KEPT_IMAGES = KeptImages()


def clear_kept():
    """Forget every kept pass and kept image (after parameter values were written behind torch's back: `.data`, raw pointers)."""
    KEPT_PASSES.clear()
    KEPT_IMAGES.clear()


class _ReblendKept(torch.autograd.Function):
    """forward: ogs_raster_forward_reblend on a kept pass; backward: the features-only ogs_raster_backward on the same state."""

    @staticmethod
    def forward(ctx, extra_feats, entry: _KeptPass, raster_settings: GaussianRasterizationSettings):
        rs = raster_settings
        ctx.set_materialize_grads(False)
        _require_gpu(extra_feats, "colors_precomp")
        dev = extra_feats.device
        lib = _lib.lib()
        P, W, H, Cn = entry.P, entry.W, entry.H, entry.Cn
        cols = _f32c(extra_feats)
        if cols is None or tuple(cols.shape) != (P, entry.E):
            raise RuntimeError(f"the kept pass blends {entry.E} caller channels of {P} Gaussians, got {tuple(extra_feats.shape)}")
        bg = _f32c(rs.bg.to(dev))
        if bg is None or bg.numel() != Cn:
            if bg is not None and bg.numel() == 3 and Cn > 3:
                bg = _tiled_bg(bg, Cn)
            else:
                raise RuntimeError(f"bg must have {Cn} entries")
        color = torch.empty(Cn, H, W, dtype=torch.float32, device=dev)
        depth = torch.empty(1, H, W, dtype=torch.float32, device=dev)
        alpha = torch.empty(1, H, W, dtype=torch.float32, device=dev)
        a = OgsRasterFwdArgs()
        a.P, a.W, a.H, a.C = P, W, H, Cn
        a.sh_coeffs = 1 if entry.fused else 0          # flag only: channels 0..2 of the kept records stay (header)
        a.debug = int(bool(rs.debug))
        a.bg, a.colors_precomp = ptr(bg), ptr(cols)
        a.out_color, a.out_depth, a.out_alpha = ptr(color), ptr(depth), ptr(alpha)
        a.image_buffer, a.sorted_rec, a.quad_list = ptr(entry.image), ptr(entry.sorted_rec), ptr(entry.quad_list)
        check(lib.ogs_raster_forward_reblend(C.byref(a), _stream()), "ogs_raster_forward_reblend")
        PASS_STATS["reblend"] += 1
        ctx.entry, ctx.raster_settings = entry, rs
        ctx.save_for_backward(cols)
        radii = entry.radii.detach()                   # an alias: the pass' radii are those of the kept pass
        ctx.mark_non_differentiable(radii)
        return color, radii, depth, alpha

    @staticmethod
    def backward(ctx, grad_color, grad_radii, grad_depth, grad_alpha):
        if not ctx.needs_input_grad[0]:
            return None, None, None
        e, rs = ctx.entry, ctx.raster_settings
        (cols,) = ctx.saved_tensors
        dev = cols.device
        lib = _lib.lib()
        gc = _f32c(grad_color)
        if gc is None:
            return torch.zeros_like(cols), None, None
        g_col = torch.empty(e.P, e.E, dtype=torch.float32, device=dev)
        bwd_tmp = torch.empty(int(lib.ogs_raster_backward_tmp_bytes(e.P)), dtype=torch.uint8, device=dev)
        b = OgsRasterBwdArgs()
        b.P, b.W, b.H, b.C = e.P, e.W, e.H, e.Cn
        b.debug = int(bool(rs.debug))
        b.num_rendered, b.num_groups = int(e.D), 1
        b.colors_precomp = ptr(cols)
        b.shs = ptr(cols) if e.fused else None         # tested against NULL only in the features-only pass (header)
        b.radii, b.dL_dcolor = ptr(e.radii), ptr(gc)
        b.image_buffer, b.sorted_rec, b.quad_list, b.bwd_tmp = ptr(e.image), ptr(e.sorted_rec), ptr(e.quad_list), ptr(bwd_tmp)
        b.dL_dcolors = ptr(g_col)
        check(lib.ogs_raster_backward(C.byref(b), _stream()), "ogs_raster_backward")
        return g_col, None, None


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings, sh_rgb_sink=None):
    """sh_rgb_sink (extension, data parallelism): a list.  When given, backward does NOT produce the dense
    dL/dsh (shs.grad stays None); it appends the [P,3] clamp-masked gradient of the SH-evaluated RGB instead, from
    which dp.ShGradExchange rebuilds the sum over views of dL/dsh (see include/ogs_raster.h, dL_dsh_rgb)."""
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, raster_settings, 0, sh_rgb_sink)


def rasterize_fused(means3D, means2D, opacities, shs, extra_feats, raster_settings, scales=None, rotations=None,
                    cov3D_precomp=None, detach_extra_from_geometry=True, sh_rgb_sink=None, frozen_key=None):
    """ONE pass for what the reference renders in several (gaussian_renderer/__init__.py:104-163): RGB from SH in
    channels 0..2 plus `extra_feats` [P, 3|6|9] (e.g. the 6-D ins_feat) in the following channels -- one
    preprocess / sort / blend for all of them.  Returns (color [3+E,H,W], radii, depth, alpha).

    detach_extra_from_geometry=True reproduces the stage-1/2 training graph (train.py:431-436): the loss on the
    extra channels reaches only `extra_feats`; geometry, opacity and means2D receive gradient from the RGB /
    depth / alpha outputs alone -- bit-for-bit what two separate reference passes (RGB with all gradients, feature
    pass with everything else detached) would accumulate.

    frozen_key (extension): ``(slot, key, holds[, generation])`` -- the caller vouches that every input but `extra_feats` (and the
    background) is the same whenever `key` is the same (renderer.py: storages + version counters of the model's parameters,
    the camera, the image size).  When none of those inputs requires grad, the first pass of a slot is kept (KEPT_PASSES,
    budget OGS_KEPT_PASSES_GB, 0 = off) and later passes with an equal key re-blend it: one launch instead of the
    whole binning pipeline, same bits.  `holds` (any object) is kept alive with the entry so that the addresses in `key`
    cannot be recycled while it exists; `generation` (optional, hashable) names the model state the key belongs to: entries
    of other generations are dropped first when the budget is short."""
    empty = torch.Tensor([])
    args = (means3D, means2D, shs, extra_feats, opacities, empty if scales is None else scales,
            empty if rotations is None else rotations, empty if cov3D_precomp is None else cov3D_precomp, raster_settings,
            3 if detach_extra_from_geometry else 0, sh_rgb_sink)
    if frozen_key is None or not means3D.is_cuda or not KEPT_PASSES.enabled(means3D.device) or raster_settings.debug or \
            (torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in
                                             (means3D, means2D, opacities, shs, scales, rotations, cov3D_precomp))):
        return _RasterizeGaussians.apply(*args)
    slot, key, holds = frozen_key[:3]
    generation = frozen_key[3] if len(frozen_key) > 3 else None
    key = (key, bool(FULL_BINNING), int(extra_feats.shape[-1]))
    entry = KEPT_PASSES.lookup(slot, key)
    if entry is not None:
        return _ReblendKept.apply(extra_feats, entry, raster_settings)
    sink: list = []
    out = _RasterizeGaussians.apply(*args, None, 1, sink)
    if sink:
        KEPT_PASSES.admit(slot, key, holds, sink[0], generation)
    return out


def rasterize_groups(means3D, means2D, opacities, group_ids, num_groups, raster_settings, shs=None,
                     colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None):
    """Batched subset rendering: ONE pass returns, for every g in [0, num_groups), the
    images the reference obtains by calling the rasterizer on the boolean-indexed subset ``group_ids == g``
    (its per-cluster loops, gaussian_renderer/__init__.py:203-225,327-345): one preprocess, one sort and one
    launch sequence instead of num_groups of each, and no index copies of the inputs.

    Returns (color [G,C,H,W], radii [P], depth [G,1,H,W], alpha [G,1,H,W]); Gaussians whose id is outside
    [0, num_groups) are not rendered (radii 0).  Gradients flow to the full-size inputs."""
    empty = torch.Tensor([])
    if (shs is None) == (colors_precomp is None):
        raise Exception('Please provide excatly one of either SHs or precomputed colors!')
    return _RasterizeGaussians.apply(means3D, means2D, empty if shs is None else shs,
                                     empty if colors_precomp is None else colors_precomp, opacities,
                                     empty if scales is None else scales, empty if rotations is None else rotations,
                                     empty if cov3D_precomp is None else cov3D_precomp, raster_settings, 0, None,
                                     group_ids, int(num_groups))


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings: GaussianRasterizationSettings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions: torch.Tensor) -> torch.Tensor:
        """bool[P]: near-plane test (upstream `mark_visible`; unused by the reference's own code)."""
        _require_gpu(positions, "positions")
        rs = self.raster_settings
        with torch.no_grad():
            pos = _f32c(positions)
            P = 0 if pos is None else int(pos.shape[0])
            out = torch.zeros(P, dtype=torch.uint8, device=positions.device)
            if P:
                view = _f32c(rs.viewmatrix.to(positions.device))
                proj = _f32c(rs.projmatrix.to(positions.device))
                check(_lib.lib().ogs_mark_visible(P, ptr(pos), ptr(view), ptr(proj), ptr(out), _stream()),
                      "ogs_mark_visible")
        return out.bool()

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        rs = self.raster_settings
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        empty = torch.Tensor([])
        shs = empty if shs is None else shs
        colors_precomp = empty if colors_precomp is None else colors_precomp
        scales = empty if scales is None else scales
        rotations = empty if rotations is None else rotations
        cov3D_precomp = empty if cov3D_precomp is None else cov3D_precomp
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
                                   cov3D_precomp, rs)
