"""TEST INFRASTRUCTURE ONLY -- CPU restatement of one pass of the tile rasterizer.

PARITY UNPINNED: the reference's rasterizer is the pip package
``ashawkey_diff_gaussian_rasterization`` (import at
/root/reference/gaussian_renderer/__init__.py:15), whose C++/CUDA source is NOT in
/root/reference (.MISSING_LARGE_BLOBS:1 lists the un-vendored
submodules/ashawkey-diff-gaussian-rasterization.zip; version unpinned, README.md:34-37).
The reference holds no tests, golden vectors or fixtures for this path, so this oracle
restates the *published algorithm* (3DGS tile rasterizer + the ashawkey depth/alpha
outputs) as specified in SURVEY.md Appendix A, and is anchored on what the reference
does hold:
  * the call sites and I/O contract at gaussian_renderer/__init__.py:55-70,104-112;
  * the in-reference restatements of its sub-formulas, which DO run here and pin the
    fragments below through tests/golden/ref_fragments.npz
    (SH polynomial utils/sh_utils.py:57-112 with the +0.5 / clamp_min(0) of
    gaussian_renderer/__init__.py:96-97; quaternion->R and R S S^T R^T with packing order
    xx,xy,xz,yy,yz,zz utils/general_utils.py:64-110 + scene/gaussian_model.py:41-45;
    projection/transposition conventions utils/graphics_utils.py:54-74 +
    scene/cameras.py:71-78; the 1/(w+1e-7) homogeneous divide utils/graphics_utils.py:22-29).
Everything else is checked by self-consistency (tests/test_oracle_raster.py): shs path ==
colors_precomp path, scales/rotations path == cov3D_precomp path, analytic HIP backward vs
float64 autograd through this restatement.

Two layers:
  * ``preprocess`` / ``bin_tiles``: NumPy float32 with an explicit, documented operation
    order and NO fused multiply-add.  The HIP preprocess kernel is compiled with
    -ffp-contract=off and follows the same order, so radii, tile rects, depth bits and
    hence the (tile<<32 | depth_bits) keys are compared BIT-EXACT.
  * ``blend`` : per-tile vectorised PyTorch (float32 for forward parity and as the
    "pure-PyTorch CPU alpha-blend" baseline; float64 + autograd as the gradient oracle).
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np
import torch

F = np.float32
BLOCK = 16

SH_C0 = F(0.28209479177387814)
SH_C1 = F(0.4886025119029199)
SH_C2 = [F(v) for v in (1.0925484305920792, -1.0925484305920792, 0.31539156525252005,
                        -1.0925484305920792, 0.5462742152960396)]
SH_C3 = [F(v) for v in (-0.5900435899266435, 2.890611442640554, -0.4570457994644658,
                        0.3731763325901154, -0.4570457994644658, 1.445305721320277,
                        -0.5900435899266435)]


@dataclass
class Geom:
    """Per-Gaussian forward state."""
    depth: np.ndarray        # [P] f32   p_view.z
    radii: np.ndarray        # [P] i32
    xy: np.ndarray           # [P,2] f32 pixel centre
    conic: np.ndarray        # [P,3] f32 (A, B, C)
    opacity: np.ndarray      # [P] f32
    rgb: np.ndarray          # [P,C] f32 blended features
    clamped: np.ndarray      # [P,3] bool (SH path only)
    rect_min: np.ndarray     # [P,2] i32
    rect_max: np.ndarray     # [P,2] i32
    tiles_touched: np.ndarray  # [P] u32
    cov3D: np.ndarray        # [P,6] f32


def _as32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def quat_to_rot(rot):
    """R(q), q=(r,x,y,z) used AS GIVEN (A.1 step 3; same matrix as utils/general_utils.py:78-99
    minus its normalisation, which the caller does at scene/gaussian_model.py:131-132)."""
    r, x, y, z = rot[:, 0], rot[:, 1], rot[:, 2], rot[:, 3]
    one, two = F(1.0), F(2.0)
    R = np.empty((rot.shape[0], 3, 3), dtype=np.float32)
    R[:, 0, 0] = one - two * (y * y + z * z)
    R[:, 0, 1] = two * (x * y - r * z)
    R[:, 0, 2] = two * (x * z + r * y)
    R[:, 1, 0] = two * (x * y + r * z)
    R[:, 1, 1] = one - two * (x * x + z * z)
    R[:, 1, 2] = two * (y * z - r * x)
    R[:, 2, 0] = two * (x * z - r * y)
    R[:, 2, 1] = two * (y * z + r * x)
    R[:, 2, 2] = one - two * (x * x + y * y)
    return R


def cov3d_from_scale_rot(scales, rotations, scale_modifier):
    """Sigma = R S S^T R^T packed (xx,xy,xz,yy,yz,zz) -- scene/gaussian_model.py:41-45,
    utils/general_utils.py:64-73."""
    s = F(scale_modifier) * scales
    R = quat_to_rot(rotations)
    M = R * s[:, None, :]          # M_ik = R_ik * s_k
    def dot(i, j):
        return M[:, i, 0] * M[:, j, 0] + M[:, i, 1] * M[:, j, 1] + M[:, i, 2] * M[:, j, 2]
    return np.stack([dot(0, 0), dot(0, 1), dot(0, 2), dot(1, 1), dot(1, 2), dot(2, 2)], axis=1)


def eval_sh_rgb(sh_degree, shs, means3D, campos):
    """SH -> RGB, +0.5, clamp>=0 (A.1 step 9; polynomial of utils/sh_utils.py:74-100;
    coefficient layout shs[P,K,3], scene/gaussian_model.py:151-155)."""
    d = means3D - campos[None, :]
    ln = np.sqrt(d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2])
    x, y, z = d[:, 0] / ln, d[:, 1] / ln, d[:, 2] / ln
    x, y, z = x[:, None], y[:, None], z[:, None]
    sh = shs
    res = SH_C0 * sh[:, 0]
    if sh_degree > 0:
        res = res - SH_C1 * y * sh[:, 1] + SH_C1 * z * sh[:, 2] - SH_C1 * x * sh[:, 3]
        if sh_degree > 1:
            xx, yy, zz = x * x, y * y, z * z
            xy, yz, xz = x * y, y * z, x * z
            res = (res + SH_C2[0] * xy * sh[:, 4] + SH_C2[1] * yz * sh[:, 5]
                   + SH_C2[2] * (F(2.0) * zz - xx - yy) * sh[:, 6]
                   + SH_C2[3] * xz * sh[:, 7] + SH_C2[4] * (xx - yy) * sh[:, 8])
            if sh_degree > 2:
                res = (res + SH_C3[0] * y * (F(3.0) * xx - yy) * sh[:, 9]
                       + SH_C3[1] * xy * z * sh[:, 10]
                       + SH_C3[2] * y * (F(4.0) * zz - xx - yy) * sh[:, 11]
                       + SH_C3[3] * z * (F(2.0) * zz - F(3.0) * xx - F(3.0) * yy) * sh[:, 12]
                       + SH_C3[4] * x * (F(4.0) * zz - xx - yy) * sh[:, 13]
                       + SH_C3[5] * z * (xx - yy) * sh[:, 14]
                       + SH_C3[6] * x * (xx - F(3.0) * yy) * sh[:, 15])
    res = res + F(0.5)
    clamped = res < 0
    return np.maximum(res, F(0.0)).astype(np.float32), clamped


def preprocess(means3D, opacities, viewmatrix, projmatrix, campos, W, H, tanfovx, tanfovy,
               scales=None, rotations=None, cov3D_precomp=None, shs=None, colors_precomp=None,
               scale_modifier=1.0, sh_degree=0) -> Geom:
    """SURVEY.md Appendix A.1.  float32, left-to-right evaluation, no FMA."""
    means3D = _as32(means3D)
    P = means3D.shape[0]
    V = _as32(viewmatrix).reshape(16)
    M = _as32(projmatrix).reshape(16)
    campos = _as32(campos).reshape(3)
    opac = _as32(opacities).reshape(P)
    tfx, tfy = F(tanfovx), F(tanfovy)
    fx = F(W) / (F(2.0) * tfx)
    fy = F(H) / (F(2.0) * tfy)
    gx, gy = (W + BLOCK - 1) // BLOCK, (H + BLOCK - 1) // BLOCK
    x, y, z = means3D[:, 0], means3D[:, 1], means3D[:, 2]

    with np.errstate(all="ignore"):
        # 1. view space, near cull
        pvx = V[0] * x + V[4] * y + V[8] * z + V[12]
        pvy = V[1] * x + V[5] * y + V[9] * z + V[13]
        pvz = V[2] * x + V[6] * y + V[10] * z + V[14]
        ok = pvz > F(0.2)
        # 2. clip space, homogeneous divide
        hx = M[0] * x + M[4] * y + M[8] * z + M[12]
        hy = M[1] * x + M[5] * y + M[9] * z + M[13]
        hw = M[3] * x + M[7] * y + M[11] * z + M[15]
        p_w = F(1.0) / (hw + F(1e-7))
        projx, projy = hx * p_w, hy * p_w
        # 3. 3D covariance
        if cov3D_precomp is not None:
            cov3D = _as32(cov3D_precomp)
        else:
            cov3D = cov3d_from_scale_rot(_as32(scales), _as32(rotations), scale_modifier)
        Sxx, Sxy, Sxz, Syy, Syz, Szz = (cov3D[:, i] for i in range(6))
        # 4. EWA splat
        limx, limy = F(1.3) * tfx, F(1.3) * tfy
        txtz, tytz = pvx / pvz, pvy / pvz
        tx = np.minimum(limx, np.maximum(-limx, txtz)) * pvz
        ty = np.minimum(limy, np.maximum(-limy, tytz)) * pvz
        tz = pvz
        J00 = fx / tz
        J02 = -(fx * tx) / (tz * tz)
        J11 = fy / tz
        J12 = -(fy * ty) / (tz * tz)
        # Wr[i][j] = V[4*j+i]  (rotation part of world->view, column-vector convention)
        T00 = J00 * V[0] + J02 * V[2]
        T01 = J00 * V[4] + J02 * V[6]
        T02 = J00 * V[8] + J02 * V[10]
        T10 = J11 * V[1] + J12 * V[2]
        T11 = J11 * V[5] + J12 * V[6]
        T12 = J11 * V[9] + J12 * V[10]
        a0 = T00 * Sxx + T01 * Sxy + T02 * Sxz
        a1 = T00 * Sxy + T01 * Syy + T02 * Syz
        a2 = T00 * Sxz + T01 * Syz + T02 * Szz
        b0 = T10 * Sxx + T11 * Sxy + T12 * Sxz
        b1 = T10 * Sxy + T11 * Syy + T12 * Syz
        b2 = T10 * Sxz + T11 * Syz + T12 * Szz
        ca = a0 * T00 + a1 * T01 + a2 * T02 + F(0.3)
        cb = a0 * T10 + a1 * T11 + a2 * T12
        cc = b0 * T10 + b1 * T11 + b2 * T12 + F(0.3)
        # 5. conic
        det = ca * cc - cb * cb
        ok &= det != 0
        det_inv = F(1.0) / det
        conic = np.stack([cc * det_inv, -cb * det_inv, ca * det_inv], axis=1)
        # 6. radius
        mid = F(0.5) * (ca + cc)
        sq = np.sqrt(np.maximum(F(0.1), mid * mid - det))
        lam = np.maximum(mid + sq, mid - sq)
        rad_f = np.ceil(F(3.0) * np.sqrt(lam))
        rad_f = np.where(np.isfinite(rad_f), rad_f, F(0.0))
        radius = rad_f.astype(np.int32)
        ok &= radius > 0
        # 7. pixel centre
        px = ((projx + F(1.0)) * F(W) - F(1.0)) * F(0.5)
        py = ((projy + F(1.0)) * F(H) - F(1.0)) * F(0.5)
        # 8. tile rect ((int) truncation toward zero, then clamp to the grid)
        rf = radius.astype(np.float32)

        def tr(v):
            v = np.where(np.isfinite(v), v, F(0.0))
            return np.trunc(v).astype(np.int64)
        rminx = np.clip(tr((px - rf) / F(BLOCK)), 0, gx)
        rminy = np.clip(tr((py - rf) / F(BLOCK)), 0, gy)
        rmaxx = np.clip(tr((px + rf + F(BLOCK) - F(1.0)) / F(BLOCK)), 0, gx)
        rmaxy = np.clip(tr((py + rf + F(BLOCK) - F(1.0)) / F(BLOCK)), 0, gy)
        area = (rmaxx - rminx) * (rmaxy - rminy)
        ok &= area != 0
        # 9. colour
        if colors_precomp is not None:
            rgb = _as32(colors_precomp)
            clamped = np.zeros((P, 3), dtype=bool)
        else:
            rgb, clamped = eval_sh_rgb(sh_degree, _as32(shs), means3D, campos)

    z32 = np.zeros(P, dtype=np.int32)
    return Geom(
        depth=np.where(ok, pvz, F(0.0)).astype(np.float32),
        radii=np.where(ok, radius, z32).astype(np.int32),
        xy=np.where(ok[:, None], np.stack([px, py], axis=1), F(0.0)).astype(np.float32),
        conic=np.where(ok[:, None], conic, F(0.0)).astype(np.float32),
        opacity=opac,
        rgb=rgb,
        clamped=clamped & ok[:, None],
        rect_min=np.stack([rminx, rminy], axis=1).astype(np.int32),
        rect_max=np.stack([rmaxx, rmaxy], axis=1).astype(np.int32),
        tiles_touched=np.where(ok, area, 0).astype(np.uint32),
        cov3D=cov3D,
    )


@dataclass
class Binning:
    num_rendered: int
    keys_sorted: np.ndarray     # [D] u64  (tile << 32) | float_bits(depth)
    point_list: np.ndarray      # [D] u32  Gaussian index per sorted entry
    ranges: np.ndarray          # [T,2] u32 [start,end) per tile, (0,0) if empty
    offsets: np.ndarray         # [P] u32 inclusive scan of tiles_touched


def bin_tiles(g: Geom, W, H) -> Binning:
    """duplicate with keys, stable sort, tile ranges."""
    gx, gy = (W + BLOCK - 1) // BLOCK, (H + BLOCK - 1) // BLOCK
    tt = g.tiles_touched.astype(np.int64)
    offsets = np.cumsum(tt)
    D = int(offsets[-1]) if len(offsets) else 0
    vis = np.nonzero(g.radii > 0)[0]
    depth_bits = g.depth.view(np.uint32).astype(np.uint64)
    # emission order: Gaussian index, then y, then x (row-major inside the rect)
    w = (g.rect_max[vis, 0] - g.rect_min[vis, 0]).astype(np.int64)
    cnt = tt[vis]
    gid = np.repeat(vis, cnt)
    start = np.repeat(offsets[vis] - cnt, cnt)
    local = np.arange(D, dtype=np.int64) - start
    wrep = np.repeat(w, cnt)
    ty = np.repeat(g.rect_min[vis, 1].astype(np.int64), cnt) + local // np.maximum(wrep, 1)
    tx = np.repeat(g.rect_min[vis, 0].astype(np.int64), cnt) + local % np.maximum(wrep, 1)
    tile = (ty * gx + tx).astype(np.uint64)
    keys = (tile << np.uint64(32)) | depth_bits[gid]
    order = np.argsort(keys, kind="stable")
    keys_sorted = keys[order]
    point_list = gid[order].astype(np.uint32)
    ranges = np.zeros((gx * gy, 2), dtype=np.uint32)
    if D:
        tile_sorted = (keys_sorted >> np.uint64(32)).astype(np.int64)
        first = np.nonzero(np.r_[True, tile_sorted[1:] != tile_sorted[:-1]])[0]
        last = np.r_[first[1:], D]
        ranges[tile_sorted[first], 0] = first
        ranges[tile_sorted[first], 1] = last
    return Binning(D, keys_sorted, point_list, ranges, offsets.astype(np.uint32))


def _tile_pixels(tx, ty, dtype):
    ax = torch.arange(BLOCK, dtype=dtype)
    X = (tx * BLOCK + ax)[None, :].expand(BLOCK, BLOCK).reshape(-1)
    Y = (ty * BLOCK + ax)[:, None].expand(BLOCK, BLOCK).reshape(-1)
    return X, Y


def blend(xy, conic, opacity, feats, depth, ranges, point_list, W, H, bg, tiles=None):
    """SURVEY.md Appendix A.3, vectorised per 16x16 tile; differentiable (torch autograd).

    xy [P,2], conic [P,3], opacity [P], feats [P,C], depth [P]: torch tensors of one dtype
    (float32 -> forward parity / CPU baseline, float64 -> gradient oracle).
    ranges [T,2], point_list [D]: integer arrays from ``bin_tiles``.
    tiles: optional iterable of tile ids to render (others are left at zero) -- used for the
    bounded CPU-baseline sample.
    Returns color [C,H,W] (+T*bg), depth [1,H,W], alpha [1,H,W], n_contrib [H,W] (int32).
    The min(0.99, .) clamp passes gradient straight through, as the reference backward does
    (A.4: dL/dG = opacity * dL/dalpha regardless of the clamp).
    """
    dtype = xy.dtype
    C = feats.shape[1]
    gx, gy = (W + BLOCK - 1) // BLOCK, (H + BLOCK - 1) // BLOCK
    Hp, Wp = gy * BLOCK, gx * BLOCK
    ranges = np.asarray(ranges).astype(np.int64)
    pl = torch.as_tensor(np.asarray(point_list).astype(np.int64))
    bg = bg.to(dtype)
    tile_ids = range(gx * gy) if tiles is None else tiles
    col_tiles, dep_tiles, alp_tiles, nc_tiles, where = [], [], [], [], []
    for t in tile_ids:
        s, e = int(ranges[t, 0]), int(ranges[t, 1])
        ty, tx = divmod(t, gx)
        if e <= s:
            continue
        ids = pl[s:e]
        X, Y = _tile_pixels(tx, ty, dtype)
        dx = xy[ids, 0][:, None] - X[None, :]
        dy = xy[ids, 1][:, None] - Y[None, :]
        A, B, Cc = conic[ids, 0][:, None], conic[ids, 1][:, None], conic[ids, 2][:, None]
        power = -0.5 * (A * dx * dx + Cc * dy * dy) - B * dx * dy
        # exp of a clamped exponent: below -80 the result is a denormal (fp32) that can never reach alpha >= 1/255
        # (that needs power >= ln(1/255) = -5.5), and denormal arithmetic is 10-50x slower on the host; the
        # clamp changes no valid contribution and no gradient (clamped entries are masked out by `valid`)
        G = torch.exp(torch.clamp(power, min=-80.0))
        a_raw = opacity[ids][:, None] * G
        alpha = a_raw + (torch.clamp(a_raw, max=0.99) - a_raw).detach()
        valid = (power <= 0) & (alpha >= 1.0 / 255.0)
        a = torch.where(valid, alpha, torch.zeros_like(alpha))
        one_m = 1.0 - a
        Tincl = torch.cumprod(one_m, dim=0)
        stop = valid & (Tincl < 1e-4)
        stopped = torch.cummax(stop.to(torch.int8), dim=0)[0].bool()
        active = valid & ~stopped
        Texcl = torch.cat([torch.ones_like(Tincl[:1]), Tincl[:-1]], dim=0)
        w = torch.where(active, a * Texcl, torch.zeros_like(a))
        Tfinal = torch.prod(torch.where(active, one_m, torch.ones_like(one_m)), dim=0)
        col = w.t() @ feats[ids] + Tfinal[:, None] * bg[None, :]
        dep = w.t() @ depth[ids]
        alp = w.sum(dim=0)
        idx1 = torch.arange(1, e - s + 1, dtype=torch.int32)[:, None]
        nc = (idx1 * active.to(torch.int32)).max(dim=0)[0]
        col_tiles.append(col); dep_tiles.append(dep); alp_tiles.append(alp); nc_tiles.append(nc)
        where.append((ty, tx))
    color = bg[:, None, None].expand(C, Hp, Wp).clone() if tiles is None else torch.zeros(C, Hp, Wp, dtype=dtype)
    dmap = torch.zeros(Hp, Wp, dtype=dtype)
    amap = torch.zeros(Hp, Wp, dtype=dtype)
    nmap = torch.zeros(Hp, Wp, dtype=torch.int32)
    if where:
        tys = torch.tensor([w_[0] for w_ in where]); txs = torch.tensor([w_[1] for w_ in where])
        # scatter whole tiles (one index_put per output keeps autograd cheap)
        colT = torch.stack(col_tiles)      # [n,256,C]
        n = colT.shape[0]
        ax = torch.arange(BLOCK)
        yy = (tys[:, None, None] * BLOCK + ax[None, :, None]).expand(n, BLOCK, BLOCK).reshape(n, -1)
        xx = (txs[:, None, None] * BLOCK + ax[None, None, :]).expand(n, BLOCK, BLOCK).reshape(n, -1)
        flat = (yy * Wp + xx).reshape(-1)
        color = color.reshape(C, -1).index_copy(1, flat, colT.reshape(-1, C).t()).reshape(C, Hp, Wp)
        dmap = dmap.reshape(-1).index_copy(0, flat, torch.stack(dep_tiles).reshape(-1)).reshape(Hp, Wp)
        amap = amap.reshape(-1).index_copy(0, flat, torch.stack(alp_tiles).reshape(-1)).reshape(Hp, Wp)
        nmap = nmap.reshape(-1).index_copy(0, flat, torch.stack(nc_tiles).reshape(-1)).reshape(Hp, Wp)
    return color[:, :H, :W], dmap[None, :H, :W], amap[None, :H, :W], nmap[:H, :W]


# ----------------------------------------------------------------------------------------
# Differentiable (torch) restatement of A.1 -- same formulas as ``preprocess`` -- used with
# a FIXED binning to obtain float64 autograd gradients for every input (A.4 + A.5 oracle).
# ----------------------------------------------------------------------------------------

def _quat_to_rot_t(q):
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    return torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=1).reshape(-1, 3, 3)


def eval_sh_rgb_t(sh_degree, shs, means3D, campos):
    d = means3D - campos[None, :]
    d = d / d.norm(dim=1, keepdim=True)
    x, y, z = d[:, 0:1], d[:, 1:2], d[:, 2:3]
    sh = shs
    c0, c1 = float(SH_C0), float(SH_C1)
    c2 = [float(v) for v in SH_C2]; c3 = [float(v) for v in SH_C3]
    res = c0 * sh[:, 0]
    if sh_degree > 0:
        res = res - c1 * y * sh[:, 1] + c1 * z * sh[:, 2] - c1 * x * sh[:, 3]
        if sh_degree > 1:
            xx, yy, zz = x * x, y * y, z * z
            xy, yz, xz = x * y, y * z, x * z
            res = (res + c2[0] * xy * sh[:, 4] + c2[1] * yz * sh[:, 5]
                   + c2[2] * (2 * zz - xx - yy) * sh[:, 6] + c2[3] * xz * sh[:, 7]
                   + c2[4] * (xx - yy) * sh[:, 8])
            if sh_degree > 2:
                res = (res + c3[0] * y * (3 * xx - yy) * sh[:, 9] + c3[1] * xy * z * sh[:, 10]
                       + c3[2] * y * (4 * zz - xx - yy) * sh[:, 11]
                       + c3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[:, 12]
                       + c3[4] * x * (4 * zz - xx - yy) * sh[:, 13]
                       + c3[5] * z * (xx - yy) * sh[:, 14] + c3[6] * x * (xx - 3 * yy) * sh[:, 15])
    return torch.clamp_min(res + 0.5, 0.0)


def preprocess_t(means3D, means2D, opacities, viewmatrix, projmatrix, campos, W, H, tanfovx, tanfovy,
                 scales=None, rotations=None, cov3D_precomp=None, shs=None, colors_precomp=None,
                 scale_modifier=1.0, sh_degree=0):
    """Differentiable A.1 (torch, any float dtype).  ``means2D`` is the zero [P,3] gradient sink of
    gaussian_renderer/__init__.py:45: it displaces the NDC position, so its gradient is the
    pixel-space gradient scaled by (0.5*W, 0.5*H) (A.4).
    Returns xy [P,2], conic [P,3], opacity [P], rgb [P,C], depth [P]."""
    V = viewmatrix.reshape(16); M = projmatrix.reshape(16)
    x, y, z = means3D[:, 0], means3D[:, 1], means3D[:, 2]
    pvx = V[0] * x + V[4] * y + V[8] * z + V[12]
    pvy = V[1] * x + V[5] * y + V[9] * z + V[13]
    pvz = V[2] * x + V[6] * y + V[10] * z + V[14]
    hx = M[0] * x + M[4] * y + M[8] * z + M[12]
    hy = M[1] * x + M[5] * y + M[9] * z + M[13]
    hw = M[3] * x + M[7] * y + M[11] * z + M[15]
    p_w = 1.0 / (hw + 1e-7)
    projx = hx * p_w + means2D[:, 0]
    projy = hy * p_w + means2D[:, 1]
    if cov3D_precomp is not None:
        cov = cov3D_precomp
    else:
        R = _quat_to_rot_t(rotations)
        Mm = R * (scale_modifier * scales)[:, None, :]
        S = Mm @ Mm.transpose(1, 2)
        cov = torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], dim=1)
    Sxx, Sxy, Sxz, Syy, Syz, Szz = (cov[:, i] for i in range(6))
    fx = W / (2.0 * tanfovx); fy = H / (2.0 * tanfovy)
    limx, limy = 1.3 * tanfovx, 1.3 * tanfovy
    # A.5: where the +-1.3*tanfov clamp is active the reference zeroes the x / y gradient path and does
    # NOT propagate through the clamped value's dependence on z -> detach the clamped branch.
    txtz, tytz = pvx / pvz, pvy / pvz
    tx = torch.where(txtz.abs() <= limx, pvx, (torch.clamp(txtz, -limx, limx) * pvz).detach())
    ty = torch.where(tytz.abs() <= limy, pvy, (torch.clamp(tytz, -limy, limy) * pvz).detach())
    tz = pvz
    J00 = fx / tz; J02 = -(fx * tx) / (tz * tz); J11 = fy / tz; J12 = -(fy * ty) / (tz * tz)
    T00 = J00 * V[0] + J02 * V[2]; T01 = J00 * V[4] + J02 * V[6]; T02 = J00 * V[8] + J02 * V[10]
    T10 = J11 * V[1] + J12 * V[2]; T11 = J11 * V[5] + J12 * V[6]; T12 = J11 * V[9] + J12 * V[10]
    a0 = T00 * Sxx + T01 * Sxy + T02 * Sxz
    a1 = T00 * Sxy + T01 * Syy + T02 * Syz
    a2 = T00 * Sxz + T01 * Syz + T02 * Szz
    b0 = T10 * Sxx + T11 * Sxy + T12 * Sxz
    b1 = T10 * Sxy + T11 * Syy + T12 * Syz
    b2 = T10 * Sxz + T11 * Syz + T12 * Szz
    ca = a0 * T00 + a1 * T01 + a2 * T02 + 0.3
    cb = a0 * T10 + a1 * T11 + a2 * T12
    cc = b0 * T10 + b1 * T11 + b2 * T12 + 0.3
    det = ca * cc - cb * cb
    conic = torch.stack([cc / det, -cb / det, ca / det], dim=1)
    px = ((projx + 1.0) * W - 1.0) * 0.5
    py = ((projy + 1.0) * H - 1.0) * 0.5
    if colors_precomp is not None:
        rgb = colors_precomp
    else:
        rgb = eval_sh_rgb_t(sh_degree, shs, means3D, campos)
    return torch.stack([px, py], dim=1), conic, opacities.reshape(-1), rgb, pvz


def render_forward(means3D, opacities, viewmatrix, projmatrix, campos, W, H, tanfovx, tanfovy, bg,
                   scales=None, rotations=None, cov3D_precomp=None, shs=None, colors_precomp=None,
                   scale_modifier=1.0, sh_degree=0):
    """One full float32 pass (A.1-A.3).  Inputs: array-likes.  Returns dict with every intermediate."""
    g = preprocess(means3D, opacities, viewmatrix, projmatrix, campos, W, H, tanfovx, tanfovy,
                   scales, rotations, cov3D_precomp, shs, colors_precomp, scale_modifier, sh_degree)
    b = bin_tiles(g, W, H)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    color, depth, alpha, n_contrib = blend(t(g.xy), t(g.conic), t(g.opacity), t(g.rgb), t(g.depth),
                                           b.ranges, b.point_list, W, H, t(_as32(bg)))
    return dict(geom=g, binning=b, color=color.numpy(), depth=depth.numpy(), alpha=alpha.numpy(),
                n_contrib=n_contrib.numpy())


def render_backward_f64(inputs: dict, binning: Binning, W, H, tanfovx, tanfovy, bg,
                        dL_dcolor, dL_ddepth, dL_dalpha, scale_modifier=1.0, sh_degree=0, tiles=None,
                        dtype=torch.float64):
    """float64 autograd through preprocess_t + blend with the float32 binning held fixed.
    ``inputs``: dict of numpy arrays with keys means3D, opacities, viewmatrix, projmatrix, campos and
    scales+rotations | cov3D_precomp, shs | colors_precomp.  Returns dict of gradients (numpy f64),
    including 'means2D' ([P,3], z = 0).
    ``tiles``: optional list of tile ids -- only those tiles are blended (a bounded sample of a full-size
    scene: the caller passes upstream gradients that are zero outside them, so the result is the exact gradient
    of that loss).
    ``dtype``: torch.float64 is the gradient oracle.  torch.float32 evaluates the SAME graph in fp32 -- what any fp32
    implementation of the algorithm computes, skip decisions (alpha >= 1/255, T < 1e-4) included: the tests use it to
    attribute the rare Gaussian whose gradient differs from the float64 one because a decision flips in fp32."""
    t64 = lambda a: torch.tensor(np.asarray(a), dtype=dtype)
    # Only Gaussians that appear in a tile list receive gradient; restrict the graph to them so
    # culled rows (pvz<=0.2, det==0) cannot inject 0*inf NaNs, then scatter back.
    P_all = np.asarray(inputs["means3D"]).shape[0]
    if tiles is None:
        used = np.unique(np.asarray(binning.point_list).astype(np.int64))
    else:
        rg = np.asarray(binning.ranges).astype(np.int64)
        pl_all = np.asarray(binning.point_list).astype(np.int64)
        parts = [pl_all[rg[t, 0]:rg[t, 1]] for t in tiles]
        used = np.unique(np.concatenate(parts)) if parts else np.zeros(0, np.int64)
    remap = np.full(P_all, -1, dtype=np.int64); remap[used] = np.arange(len(used))
    binning = Binning(binning.num_rendered, binning.keys_sorted,
                      remap[np.asarray(binning.point_list).astype(np.int64)], binning.ranges, binning.offsets)
    leaves = {}
    for k in ("means3D", "opacities", "scales", "rotations", "cov3D_precomp", "shs", "colors_precomp"):
        if inputs.get(k) is not None:
            leaves[k] = t64(np.asarray(inputs[k])[used]).requires_grad_(True)
    P = leaves["means3D"].shape[0]
    leaves["means2D"] = torch.zeros(P, 3, dtype=dtype, requires_grad=True)
    xy, conic, op, rgb, depth = preprocess_t(
        leaves["means3D"], leaves["means2D"], leaves["opacities"], t64(inputs["viewmatrix"]),
        t64(inputs["projmatrix"]), t64(inputs["campos"]).reshape(3), W, H, float(tanfovx), float(tanfovy),
        leaves.get("scales"), leaves.get("rotations"), leaves.get("cov3D_precomp"), leaves.get("shs"),
        leaves.get("colors_precomp"), scale_modifier, sh_degree)
    color, dmap, amap, _ = blend(xy, conic, op, rgb, depth, binning.ranges, binning.point_list, W, H, t64(bg),
                                 tiles=tiles)
    loss = (color * t64(dL_dcolor)).sum() + (dmap * t64(dL_ddepth)).sum() + (amap * t64(dL_dalpha)).sum()
    names = list(leaves)
    grads = torch.autograd.grad(loss, [leaves[n] for n in names], allow_unused=True)
    out = {}
    for n, g in zip(names, grads):
        if g is None:
            out[n] = None
            continue
        full = np.zeros((P_all,) + tuple(g.shape[1:]), dtype=np.float64)
        full[used] = g.double().numpy()
        out[n] = full
    return out
