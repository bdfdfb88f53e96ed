"""``render()`` -- drop-in counterpart of /root/reference/gaussian_renderer/__init__.py:22-373.

Same signature, same flags, same 14-key result dict, same CPU-RNG draws (``torch.rand(1)`` for the rescale
coin and factor, :121-124) -- but the rasterizer passes are fused wherever the reference re-runs
preprocess + sort + blend on identical geometry:

  reference (4 passes, :104-163)                    here
  RGB | feat[:, :3] | feat[:, 3:6] | silhouette  -> ONE 9-channel pass when no rescale is drawn
                                                    (RGB + 6 feat; alpha is shared), otherwise RGB pass +
                                                    ONE 6-channel pass whose alpha IS the silhouette
  per coarse / fine cluster: 2 passes (:203-225,   -> ONE 6-channel pass per cluster
  :327-345)

The silhouette pass of the reference only keeps alpha (:153), which does not depend on colour, so taking it
from the feature pass is exact.  Gradients: every blended channel feeds the geometry gradients exactly as in
separate passes (a sum over channels); whatever the caller detached (train.py:431-436) stays detached.
"""
from __future__ import annotations

import math

import torch

from . import rasterizer as _rasterizer
from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer, rasterize_fused, rasterize_groups

# a grouped pass materialises [G, C+2, H, W] fp32: bound G per pass (the loop over passes keeps the order)
GROUPS_PER_PASS = 64
BATCH_SUBSETS = True     # False: per-subset calls exactly as the reference loops (kept for the equivalence test)

C0 = 0.28209479177387814
C1 = 0.4886025119029199
C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
      1.445305721320277, -0.5900435899266435]


def eval_sh(deg, sh, dirs):
    """SH polynomial of utils/sh_utils.py:57-112 (degrees 0-3); sh [..., C, K], dirs [..., 3]."""
    result = C0 * sh[..., 0]
    if deg > 0:
        x, y, z = dirs[..., 0:1], dirs[..., 1:2], dirs[..., 2:3]
        result = result - C1 * y * sh[..., 1] + C1 * z * sh[..., 2] - C1 * x * sh[..., 3]
        if deg > 1:
            xx, yy, zz = x * x, y * y, z * z
            xy, yz, xz = x * y, y * z, x * z
            result = (result + C2[0] * xy * sh[..., 4] + C2[1] * yz * sh[..., 5] +
                      C2[2] * (2.0 * zz - xx - yy) * sh[..., 6] + C2[3] * xz * sh[..., 7] + C2[4] * (xx - yy) * sh[..., 8])
            if deg > 2:
                result = (result + C3[0] * y * (3 * xx - yy) * sh[..., 9] + C3[1] * xy * z * sh[..., 10] +
                          C3[2] * y * (4 * zz - xx - yy) * sh[..., 11] + C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[..., 12] +
                          C3[4] * x * (4 * zz - xx - yy) * sh[..., 13] + C3[5] * z * (xx - yy) * sh[..., 14] +
                          C3[6] * x * (xx - 3 * yy) * sh[..., 15])
    return result


def _knn_mean_filter(points: torch.Tensor) -> torch.Tensor:
    """Outlier mask of the `post_process` branch (:293-309): mean squared distance to the K = sqrt(n) nearest
    neighbours (self included, as pytorch3d.ops.knn_points(x, x) returns it) below mean + std."""
    n = points.shape[0]
    K = max(1, int(n ** 0.5))
    # direct differences, as pytorch3d's kernel computes them (the default cdist goes through a matmul for > 25 rows and
    # loses ~1e-4 absolute on squared distances of ~1e-3 by cancellation)
    d2 = torch.cdist(points, points, compute_mode="donot_use_mm_for_euclid_dist") ** 2
    knn = torch.topk(d2, K, dim=1, largest=False).values
    return knn.mean(dim=-1) < knn.mean() + knn.std()


def _render_subsets(group_of_point: torch.Tensor, num_groups: int, min_points: int, raster_settings, means3D,
                    means2D, opacity, scales, rotations, cov3D_precomp, colors=None, shs=None, groups_per_pass=None):
    """Images of the subsets {p : group_of_point[p] == g} with at least `min_points` members, in ONE grouped
    rasterizer pass per GROUPS_PER_PASS subsets instead of one boolean-indexed call each (the reference's
    per-cluster loops, :203-225,327-345).  Returns (kept group numbers, colour list [C,H,W], alpha list [1,H,W])."""
    valid = group_of_point >= 0
    counts = torch.bincount(group_of_point[valid], minlength=num_groups)[:num_groups]
    kept = torch.nonzero(counts >= min_points).flatten()
    kept_list = kept.tolist()
    imgs, sils = [], []
    step = int(groups_per_pass or GROUPS_PER_PASS)
    for c0 in range(0, len(kept_list), step):
        chunk = kept[c0:c0 + step]
        remap = torch.full((num_groups + 1,), -1, dtype=torch.int32, device=group_of_point.device)
        remap[chunk] = torch.arange(chunk.numel(), dtype=torch.int32, device=group_of_point.device)
        local = remap[torch.where(valid, group_of_point, torch.full_like(group_of_point, num_groups))]
        if chunk.numel() == 1:
            # a single subset: the plain pass on the indexed subset (what the reference does)
            m = local == 0
            color, _, _, alpha = GaussianRasterizer(raster_settings)(
                means3D=means3D[m], means2D=means2D[m], shs=None if shs is None else shs[m],
                colors_precomp=None if colors is None else colors[m], opacities=opacity[m],
                scales=None if scales is None else scales[m], rotations=None if rotations is None else rotations[m],
                cov3D_precomp=None if cov3D_precomp is None else cov3D_precomp[m])
            imgs.append(color); sils.append(alpha)
            continue
        color, _, _, alpha = rasterize_groups(means3D, means2D, opacity, local, int(chunk.numel()), raster_settings,
                                              shs=shs, colors_precomp=colors, scales=scales, rotations=rotations,
                                              cov3D_precomp=cov3D_precomp)
        imgs.extend(color.unbind(0)); sils.extend(alpha.unbind(0))
    return kept_list, imgs, sils


_MODEL_PARAMS = ("_xyz", "_scaling", "_rotation", "_opacity", "_features_dc", "_features_rest")


def _frozen_geometry_key(viewpoint_camera, pc, pipe, xyz, scaling_modifier):
    """``(slot, key, holds)`` for rasterizer.rasterize_fused(frozen_key=...) or None.

    Stage >= 1 of the reference's schedule: the six geometry / appearance parameters of the model are detached
    (train.py:431-436) and nothing steps them any more, so the binning state of a camera's pass can be kept.  What vouches for
    "nothing changed": the storage address and torch's version counter of each parameter (every in-place update -- an optimizer
    step, `reset_opacity`, densification's `cat` -- bumps the counter or moves the storage; `.detach()`, which train.py repeats
    every iteration, does neither), the camera's matrices likewise, and the scalar settings of the pass.  The tensors are held
    by the entry, so an address cannot be handed out again while its key is in use.  NOT covered: writes through `.data` or
    raw pointers (they leave the counter alone) -- call ``rasterizer.clear_kept()`` after such a write, or set
    OGS_KEPT_PASSES_GB=0.  Returns None when the model does not look like the reference's GaussianModel (no such attributes,
    `get_xyz` not the `_xyz` parameter itself), when anything still requires grad, or under `pipe.debug`."""
    if getattr(pipe, "debug", False) or getattr(pipe, "compute_cov3D_python", False) or getattr(pipe, "convert_SHs_python", False):
        return None
    params = []
    for name in _MODEL_PARAMS:
        t = getattr(pc, name, None)
        if not isinstance(t, torch.Tensor) or t.requires_grad or not t.is_cuda:
            return None
        params.append(t)
    if params[0].data_ptr() != xyz.data_ptr():
        return None
    cam = (viewpoint_camera.world_view_transform, viewpoint_camera.full_proj_transform, viewpoint_camera.camera_center)
    if not all(isinstance(t, torch.Tensor) for t in cam):
        return None
    ident = lambda t: (t.data_ptr(), t._version, tuple(t.shape))
    key = (tuple(ident(t) for t in params), tuple(ident(t) for t in cam), int(viewpoint_camera.image_height),
           int(viewpoint_camera.image_width), float(viewpoint_camera.FoVx), float(viewpoint_camera.FoVy),
           float(scaling_modifier), int(pc.active_sh_degree))
    slot = (cam[0].data_ptr(), int(viewpoint_camera.image_height), int(viewpoint_camera.image_width))
    return slot, key, (tuple(params), cam), key[0]          # generation = the parameters' identities: one state of the model


def render(viewpoint_camera, pc, pipe, bg_color: torch.Tensor, iteration,
           scaling_modifier=1.0, override_color=None, visible_mask=None, mask_num=0,
           cluster_idx=None, leaf_cluster_idx=None, rescale=True, origin_feat=False,
           render_feat_map=True, render_color=True, render_cluster=False, better_vis=False,
           selected_root_id=None, selected_leaf_id=None, pre_mask=None, seg_rgb=False,
           post_process=False, root_num=64, leaf_num=10, viewspace_grad=None):
    """Render the scene.  Background tensor (bg_color) must be on GPU!

    viewspace_grad (extension; the reference has no such flag): whether ``viewspace_points.grad`` (dL/dmeans2D) is
    wanted.  Its only consumer is densification (train.py:594-598, stage 0).  None = automatic: True while any
    geometry tensor requires grad, False once they are all detached (train.py:431-436) -- the rasterizer then
    runs its features-only backward (only dL/d ins_feat, no alpha-gradient recursion); ``viewspace_points`` is
    still returned, its ``.grad`` stays None.  Pass True to force the reference's behaviour."""
    xyz = pc.get_xyz
    dev = xyz.device
    screenspace_points = torch.zeros_like(xyz, dtype=xyz.dtype, requires_grad=True, device=dev) + 0
    try:
        screenspace_points.retain_grad()
    except Exception:
        pass

    tanfovx = math.tan(viewpoint_camera.FoVx * 0.5)
    tanfovy = math.tan(viewpoint_camera.FoVy * 0.5)
    raster_settings = GaussianRasterizationSettings(
        image_height=int(viewpoint_camera.image_height), image_width=int(viewpoint_camera.image_width),
        tanfovx=tanfovx, tanfovy=tanfovy, bg=bg_color, scale_modifier=scaling_modifier,
        viewmatrix=viewpoint_camera.world_view_transform, projmatrix=viewpoint_camera.full_proj_transform,
        sh_degree=pc.active_sh_degree, campos=viewpoint_camera.camera_center, prefiltered=False, debug=pipe.debug)
    rasterizer = GaussianRasterizer(raster_settings=raster_settings)

    means3D = xyz
    means2D = screenspace_points
    opacity = pc.get_opacity

    scales = rotations = cov3D_precomp = None
    if pipe.compute_cov3D_python:
        cov3D_precomp = pc.get_covariance(scaling_modifier)
    else:
        scales = pc.get_scaling
        rotations = pc.get_rotation

    if viewspace_grad is None:
        viewspace_grad = any(t is not None and t.requires_grad for t in (means3D, opacity, scales, rotations, cov3D_precomp))
    if not viewspace_grad:
        means2D = screenspace_points.detach()

    shs = colors_precomp = None
    if override_color is None:
        if pipe.convert_SHs_python:
            shs_view = pc.get_features.transpose(1, 2).view(-1, 3, (pc.max_sh_degree + 1) ** 2)
            dir_pp = pc.get_xyz - viewpoint_camera.camera_center.repeat(pc.get_features.shape[0], 1)
            dir_pp_normalized = dir_pp / dir_pp.norm(dim=1, keepdim=True)
            colors_precomp = torch.clamp_min(eval_sh(pc.active_sh_degree, shs_view, dir_pp_normalized) + 0.5, 0.0)
        else:
            shs = pc.get_features
    else:
        colors_precomp = override_color

    # same CPU RNG draws, in the same order, as the reference (:121-124)
    prob = torch.rand(1)
    rescale_factor = torch.tensor(1.0, dtype=torch.float32, device=dev)
    rescaled = bool(prob > 0.5 and rescale)
    if rescaled:
        rescale_factor = torch.rand(1).to(dev)

    rendered_image = radii = rendered_depth = rendered_alpha = None
    rendered_ins_feat = silhouette = None
    ins_feat = None
    if render_feat_map:
        ins_feat = (pc.get_ins_feat(origin=origin_feat) + 1) / 2
    # a 12-channel pass has no backward (gradient record: C + 7 <= 16 slots): only fuse / widen that far without grad
    differentiable = torch.is_grad_enabled() and any(
        t is not None and t.requires_grad for t in (means3D, means2D, opacity, scales, rotations, cov3D_precomp, shs,
                                                    colors_precomp, ins_feat))
    max_pass_channels = 9 if differentiable else 12
    can_fuse = (render_color and render_feat_map and not rescaled and ins_feat.shape[-1] in (3, 6, 9)
                and ins_feat.shape[-1] + 3 <= max_pass_channels)
    if can_fuse:
        # RGB + features + silhouette on identical geometry: one bin / sort / blend for everything
        if shs is not None:
            # stage >= 1 (everything but ins_feat frozen, no rescale draw): the camera's first pass is kept, later ones re-blend it
            frozen = None if (viewspace_grad or scales is None) else _frozen_geometry_key(viewpoint_camera, pc, pipe, xyz,
                                                                                          scaling_modifier)
            out, radii, rendered_depth, rendered_alpha = rasterize_fused(
                means3D, means2D, opacity, shs, ins_feat, raster_settings, scales=scales * rescale_factor,
                rotations=rotations, cov3D_precomp=cov3D_precomp, detach_extra_from_geometry=False, frozen_key=frozen)
        else:
            out, radii, rendered_depth, rendered_alpha = rasterizer(
                means3D=means3D, means2D=means2D, shs=None, colors_precomp=torch.cat((colors_precomp, ins_feat), dim=1),
                opacities=opacity, scales=scales * rescale_factor, rotations=rotations, cov3D_precomp=cov3D_precomp)
        rendered_image, rendered_ins_feat = out[:3], out[3:]
        silhouette = rendered_alpha
    else:
        if render_color:
            # nothing of this pass trains once the model is frozen (the unfused RGB pass of a rescaled stage-2.1 call, every
            # stage-2.2 call): for a camera, a model state and a background tensor its outputs are the same every time
            frozen = None
            if shs is not None and not viewspace_grad and scales is not None and xyz.is_cuda and _rasterizer.KEPT_PASSES.enabled(dev):
                frozen = _frozen_geometry_key(viewpoint_camera, pc, pipe, xyz, scaling_modifier)
            hit = None
            if frozen is not None:
                img_key = (frozen[1], bg_color.data_ptr(), bg_color._version, tuple(bg_color.shape))
                hit = _rasterizer.KEPT_IMAGES.lookup(frozen[0], img_key)
            if hit is not None:
                rendered_image, radii, rendered_depth, rendered_alpha = hit
            else:
                rendered_image, radii, rendered_depth, rendered_alpha = rasterizer(
                    means3D=means3D, means2D=means2D, shs=shs, colors_precomp=colors_precomp, opacities=opacity,
                    scales=scales, rotations=rotations, cov3D_precomp=cov3D_precomp)
                if frozen is not None and not rendered_image.requires_grad:
                    _rasterizer.KEPT_IMAGES.admit(frozen[0], img_key, (frozen[2], bg_color),
                                      (rendered_image, radii, rendered_depth, rendered_alpha), frozen[3])
        if render_feat_map:
            # `scales * rescale_factor` with scales=None (compute_cov3D_python) raises, as in the reference (:135)
            if ins_feat.shape[-1] in (3, 6, 9, 12) and ins_feat.shape[-1] <= max_pass_channels:
                rendered_ins_feat, _, _, silhouette = rasterizer(
                    means3D=means3D, means2D=means2D, shs=None, colors_precomp=ins_feat, opacities=opacity,
                    scales=scales * rescale_factor, rotations=rotations, cov3D_precomp=cov3D_precomp)
            else:
                parts = []
                for lo in range(0, ins_feat.shape[-1], 3):
                    part, _, _, silhouette = rasterizer(
                        means3D=means3D, means2D=means2D, shs=None, colors_precomp=ins_feat[:, lo:lo + 3],
                        opacities=opacity, scales=scales * rescale_factor, rotations=rotations,
                        cov3D_precomp=cov3D_precomp)
                    parts.append(part)
                rendered_ins_feat = torch.cat(parts, dim=0)

    # ---- [Stage 2.2 preprocessing] coarse cluster feature maps (:168-236) ---------------------------------
    # Both cluster blocks below reduce to "label every Gaussian with the subset it is rendered in (-1: none), then
    # render all subsets": _render_subsets does that in grouped passes (BATCH_SUBSETS) or, with one group per pass,
    # as the reference's one rasterizer call per boolean-indexed subset.
    per_pass = GROUPS_PER_PASS if BATCH_SUBSETS else 1
    none_of = lambda t: torch.full_like(t, -1)
    viewed_pts = radii > 0
    if cluster_idx is not None:
        num_cluster = cluster_idx.max() + 1
        cluster_occur = torch.zeros(num_cluster).to(torch.bool)
    else:
        cluster_occur = None
    if render_cluster and cluster_idx is not None and viewed_pts.sum() != 0:
        ins_feat = (pc.get_ins_feat(origin=origin_feat) + 1) / 2
        rendered_clusters = []
        rendered_cluster_silhouettes = []
        gid = torch.where(viewed_pts, cluster_idx.to(torch.int64), none_of(cluster_idx).to(torch.int64))
        if better_vis:
            # every coarse cluster, small Gaussians only, at least 100 of them (:186-189)
            gid = torch.where((scales < 0.5).all(dim=1), gid, none_of(gid))
        elif selected_root_id is None:
            gid = none_of(gid)          # `idx != None` holds for every idx: the reference skips every cluster (:180-181)
        else:
            gid = torch.where(gid == selected_root_id, gid, none_of(gid))       # the selected one only (:180-181)
        if viewpoint_camera.bClusterOccur is not None:                           # clusters this camera never sees (:182-183)
            occ = torch.as_tensor(viewpoint_camera.bClusterOccur).to(device=gid.device, dtype=torch.bool)
            gid = torch.where(occ[gid.clamp_min(0)], gid, none_of(gid))
        kept, imgs, sils = _render_subsets(gid, int(num_cluster), 100 if better_vis else 1, raster_settings, means3D,
                                           means2D, opacity, None if scales is None else scales * rescale_factor,
                                           rotations, cov3D_precomp, colors=ins_feat, groups_per_pass=per_pass)
        if sils:
            seen = (torch.stack([s_.max() for s_ in sils]) > 0.8).tolist()      # one read-back for all clusters
            for idx, img, sil, ok in zip(kept, imgs, sils, seen):
                if ok:
                    cluster_occur[idx] = True
                    rendered_clusters.append(img)
                    rendered_cluster_silhouettes.append(sil)
        if len(rendered_cluster_silhouettes) != 0:
            rendered_cluster_silhouettes = torch.vstack(rendered_cluster_silhouettes)
    else:
        rendered_clusters, rendered_cluster_silhouettes = None, None

    # ---- [Stage 2.2 & 3] fine cluster feature maps (:239-356) --------------------------------------------------
    if leaf_cluster_idx is not None and leaf_cluster_idx.numel() > 0:
        ins_feat = (pc.get_ins_feat(origin=origin_feat) + 1) / 2
        rendered_leaf_clusters = []
        rendered_leaf_cluster_silhouettes = []
        occured_leaf_id = []
        lid = leaf_cluster_idx.to(torch.int64)
        if selected_leaf_id is not None:
            # the listed leaves rendered TOGETHER as one subset, reported under the first listed id (:287-289,317-318)
            first_leaf = int(selected_leaf_id.flatten()[0])
            n_sub, sub_base = 1, first_leaf
            gid = torch.where((lid.unsqueeze(1) == selected_leaf_id.to(lid.device)).any(dim=1), torch.zeros_like(lid), none_of(lid))
        else:
            start_leaf = selected_root_id * leaf_num if selected_root_id is not None else 0
            n_sub = leaf_num if selected_root_id is not None else root_num * leaf_num
            sub_base = start_leaf
            gid = torch.where((lid >= start_leaf) & (lid < start_leaf + n_sub), lid - start_leaf, none_of(lid))
        occ = viewpoint_camera.bClusterOccur
        if occ is not None and occ[selected_root_id] == False:                   # noqa: E712  (:284-285; indexing with
            gid = none_of(gid)                                                   # None keeps the reference's behaviour)
        if pre_mask is not None:
            gid = torch.where(pre_mask, gid, none_of(gid))
        gid = torch.where(viewed_pts, gid, none_of(gid))
        if better_vis:                                                           # small Gaussians, >= 100 per leaf (:293-296)
            gid = torch.where((scales < 0.1).all(dim=1), gid, none_of(gid))
            cnt = torch.bincount(gid[gid >= 0], minlength=n_sub)
            gid = torch.where((cnt >= 100)[gid.clamp_min(0)], gid, none_of(gid))
        if post_process:                                                         # per-leaf kNN outlier filter (:297-309)
            for g_ in torch.unique(gid[gid >= 0]).tolist():
                rows = torch.nonzero(gid == g_).flatten()
                gid[rows[~_knn_mean_filter(means3D[rows])]] = -1
        kept, imgs, sils = _render_subsets(gid, n_sub, 10, raster_settings, means3D, means2D, opacity, scales, rotations,
                                           cov3D_precomp, colors=None if seg_rgb else ins_feat,
                                           shs=shs if seg_rgb else None, groups_per_pass=per_pass)
        for k_, img, sil in zip(kept, imgs, sils):
            occured_leaf_id.append(sub_base + k_)
            if seg_rgb and ins_feat.shape[-1] > 3:      # the reference renders the same RGB twice and stacks it (:336-346)
                img = torch.cat((img, img), dim=0)
            rendered_leaf_clusters.append(img)
            rendered_leaf_cluster_silhouettes.append(sil)
        if len(rendered_leaf_cluster_silhouettes) != 0:
            rendered_leaf_cluster_silhouettes = torch.vstack(rendered_leaf_cluster_silhouettes)
    else:
        rendered_leaf_clusters = None
        rendered_leaf_cluster_silhouettes = None
        occured_leaf_id = None

    return {"render": rendered_image,
            "alpha": rendered_alpha,
            "depth": rendered_depth,
            "silhouette": silhouette,
            "ins_feat": rendered_ins_feat,
            "cluster_imgs": rendered_clusters,
            "cluster_silhouettes": rendered_cluster_silhouettes,
            "leaf_clusters_imgs": rendered_leaf_clusters,
            "leaf_cluster_silhouettes": rendered_leaf_cluster_silhouettes,
            "occured_leaf_id": occured_leaf_id,
            "cluster_occur": cluster_occur,
            "viewspace_points": screenspace_points,
            "visibility_filter": radii > 0,
            "radii": radii}
