"""Seeded inputs of the render() fixtures -- shared by tests/golden/make_render_golden.py (which runs the REFERENCE's
gaussian_renderer.render() over the CPU oracle, build container only) and tests/test_11_render_gpu.py (which runs
opengaussian_amd.renderer.render() on the GPU and compares with the stored results).  Nothing here touches
/root/reference: the file travels to the GPU box.

One case = a tiny scene (raw GaussianModel parameters), a camera, the flags of one render() call as train.py /
render_lerf_by_text.py issue it, the global-RNG seed in front of the call (the reference draws its rescale coin and factor
from torch's global CPU generator, gaussian_renderer/__init__.py:121-124) and the weights of a fixed linear loss over
every tensor of the result dict.
"""
from __future__ import annotations

import types
import zlib

import torch

from opengaussian_amd.synthetic import make_camera, make_scene

PARAM_NAMES = ("_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity", "_ins_feat")
TENSOR_KEYS = ("render", "alpha", "depth", "silhouette", "ins_feat", "cluster_silhouettes", "leaf_cluster_silhouettes")
LIST_KEYS = ("cluster_imgs", "leaf_clusters_imgs")

# name -> definition.  `rng_seed` is chosen so that the rescale coin lands as `expect_rescale` says
# (torch.manual_seed(s); torch.rand(1): s=0 -> 0.4963, s=1 -> 0.7576).
CASES = {
    # stage 1 / 2.1 training call (train.py:352-358 with the default flags): RGB, 6-D feature map, silhouette
    "stage1_norescale": dict(scene_seed=31, P=700, W=72, H=40, f=55.0, rng_seed=0, expect_rescale=False, grads=True,
                             kwargs=dict()),
    "stage1_rescale": dict(scene_seed=32, P=700, W=72, H=40, f=55.0, rng_seed=1, expect_rescale=True, grads=True,
                           pipe=dict(convert_SHs_python=True), kwargs=dict()),
    # rescale=False although the coin says "rescale" (render.py:61 passes rescale=False)
    "eval_rescale_off": dict(scene_seed=33, P=500, W=56, H=40, f=50.0, rng_seed=1, expect_rescale=False, grads=False,
                             kwargs=dict(rescale=False)),
    # stage 2.2 training call: the selected coarse cluster + its leaves (train.py:335-358)
    "stage22_selected_root": dict(scene_seed=34, P=900, W=72, H=40, f=55.0, rng_seed=1, expect_rescale=True, grads=True,
                                  quantized=True, clusters=dict(k1=4, k2=3), kwargs=dict(render_feat_map=False, render_cluster=True,
                                                                         selected_root_id=2, root_num=4, leaf_num=3)),
    # pseudo-label build (train.py:758-760, no_grad): every coarse cluster with the better_vis filters, raw (not
    # quantised) features, a camera that never sees cluster 1
    "better_vis_all_clusters": dict(scene_seed=35, P=3000, W=72, H=40, f=55.0, rng_seed=0, expect_rescale=False,
                                    grads=False, quantized=True,
                                    clusters=dict(k1=4, k2=3, no_leaf=True, occur=[True, False, True, True]),
                                    kwargs=dict(render_cluster=True, better_vis=True, render_feat_map=False, origin_feat=True,
                                                root_num=4, leaf_num=3, rescale=False)),
    # stage-3 language association (train.py:852-855): the leaves of one coarse cluster, which this camera has seen ...
    "lang_leaves_of_root": dict(scene_seed=39, P=1500, W=72, H=40, f=55.0, rng_seed=1, expect_rescale=False, grads=False,
                                quantized=True, clusters=dict(k1=4, k2=3, leaf_only=True, occur=[True, True, False, True]),
                                kwargs=dict(rescale=False, render_feat_map=False, render_cluster=True, origin_feat=True,
                                            better_vis=False, selected_root_id=1, root_num=4, leaf_num=3)),
    # ... and of one it has not (:270-271: every leaf skipped, empty lists)
    "lang_leaves_root_unseen": dict(scene_seed=39, P=1500, W=72, H=40, f=55.0, rng_seed=1, expect_rescale=False, grads=False,
                                    clusters=dict(k1=4, k2=3, leaf_only=True, occur=[True, True, False, True]),
                                    kwargs=dict(rescale=False, render_feat_map=False, render_cluster=True, origin_feat=True,
                                                better_vis=False, selected_root_id=2, root_num=4, leaf_num=3)),
    # text-query path (render_lerf_by_text.py:158-168): a union of leaves, pre-mask, RGB instead of features, kNN filter
    "selected_leaf_union_seg_rgb": dict(scene_seed=36, P=3000, W=72, H=40, f=55.0, rng_seed=1, expect_rescale=False,
                                        grads=False, clusters=dict(k1=4, k2=3, leaf_only=True), pre_mask=True,
                                        kwargs=dict(selected_leaf_id=[4, 7], render_feat_map=False, render_cluster=False,
                                                    better_vis=True, seg_rgb=True, post_process=True, rescale=False,
                                                    root_num=4, leaf_num=3)),
    # render_cluster with neither better_vis nor a selected root: the reference skips every cluster
    "cluster_none_selected": dict(scene_seed=37, P=400, W=40, H=40, f=40.0, rng_seed=0, expect_rescale=False, grads=False,
                                  clusters=dict(k1=3, k2=2, no_leaf=True), kwargs=dict(render_cluster=True, rescale=False)),
    # RGB only through a precomputed covariance (pipe.compute_cov3D_python; the feature map would raise, :135)
    "rgb_only_cov3d_python": dict(scene_seed=38, P=400, W=40, H=40, f=40.0, rng_seed=0, expect_rescale=False, grads=True,
                                  pipe=dict(compute_cov3D_python=True), kwargs=dict(render_feat_map=False)),
}


def _strip_symmetric_cov(scaling, scaling_modifier, rotation):
    """build_covariance_from_scaling_rotation of scene/gaussian_model.py:41-45 (R S S^T R^T, packed xx xy xz yy yz zz)."""
    q = rotation / rotation.norm(dim=1, keepdim=True)
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                     2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                     2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=1).reshape(-1, 3, 3)
    L = R * (scaling_modifier * scaling)[:, None, :]
    S = L @ L.transpose(1, 2)
    return torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], dim=1)


def attach_quantized(model, params: dict, device="cpu"):
    """Stage 2: `_ins_feat_q` as Quantize_kMeans.forward leaves it (scene/kmeans_quantize.py:273-275) -- the straight-through
    estimator `_ins_feat - _ins_feat.detach() + sampled` around a (here: seeded random) code-book sample."""
    if "sampled" in params:
        model._ins_feat_q = model._ins_feat - model._ins_feat.detach() + params["sampled"].to(device)


class TinyModel:
    """What render() reads of scene/gaussian_model.py:GaussianModel (:122-172), with the reference's activations
    (exp / sigmoid / normalize, :47-62) over RAW parameters, so gradients run back through them."""

    def __init__(self, params: dict, device="cpu", requires_grad=True):
        for n in PARAM_NAMES:
            setattr(self, n, params[n].detach().clone().to(device).requires_grad_(requires_grad))
        self._ins_feat_q = torch.empty(0)
        attach_quantized(self, params, device)
        self.active_sh_degree = 3
        self.max_sh_degree = 3

    get_xyz = property(lambda s: s._xyz)
    get_scaling = property(lambda s: torch.exp(s._scaling))
    get_rotation = property(lambda s: torch.nn.functional.normalize(s._rotation))
    get_opacity = property(lambda s: torch.sigmoid(s._opacity))
    get_features = property(lambda s: torch.cat((s._features_dc, s._features_rest), dim=1))

    def get_ins_feat(self, origin=False):
        f = self._ins_feat if (len(self._ins_feat_q) == 0 or origin) else self._ins_feat_q
        return torch.nn.functional.normalize(f, dim=1)

    def get_covariance(self, scaling_modifier=1):
        return _strip_symmetric_cov(self.get_scaling, scaling_modifier, self._rotation)

    def leaves(self):
        return [getattr(self, n) for n in PARAM_NAMES]


def build(name: str):
    """-> dict(params, cam, pipe, bg, kwargs, rng_seed, grads, expect_rescale) -- CPU tensors."""
    c = CASES[name]
    P, W, H, f = c["P"], c["W"], c["H"], c["f"]
    sc = make_scene(P, W, H, f, f, seed=c["scene_seed"], log_scale_mean=-3.0)
    g = torch.Generator().manual_seed(9000 + c["scene_seed"])
    op = sc.opacities.clamp(1e-4, 1 - 1e-4)
    params = {
        "_xyz": sc.means3D.clone(),
        "_features_dc": sc.shs[:, :1].clone(),
        "_features_rest": sc.shs[:, 1:].clone(),
        "_scaling": torch.log(sc.scales),
        "_rotation": sc.rotations * (0.5 + torch.rand(P, 1, generator=g)),         # not unit: the getter normalises
        "_opacity": torch.log(op / (1 - op)),
        "_ins_feat": torch.randn(P, 6, generator=g),
    }
    if c.get("quantized"):
        params["sampled"] = torch.randn(P, 6, generator=g)
    cam = make_camera(W, H, f, f)
    kwargs = dict(c["kwargs"])
    cl = c.get("clusters")
    if cl is not None:
        k1, k2 = cl["k1"], cl["k2"]
        # spatially coherent coarse clusters (vertical bands by x / z) so that each is seen as a solid region
        band = ((sc.means3D[:, 0] / sc.means3D[:, 2].clamp_min(0.3) / (W / (2 * f)) + 1.1) / 2.2 * k1).floor().clamp(0, k1 - 1)
        cluster_idx = band.to(torch.int64)
        leaf = cluster_idx * k2 + torch.randint(0, k2, (P,), generator=g)
        leaf[torch.rand(P, generator=g) < 0.05] = k1 * k2                             # the dummy id of never-assigned points
        if not cl.get("leaf_only"):
            kwargs["cluster_idx"] = cluster_idx
        if not cl.get("no_leaf"):
            kwargs["leaf_cluster_idx"] = leaf
        if cl.get("occur") is not None:
            cam.bClusterOccur = torch.tensor(cl["occur"])
    if c.get("pre_mask"):
        kwargs["pre_mask"] = torch.rand(P, generator=g) < 0.8
    if "selected_leaf_id" in kwargs:
        kwargs["selected_leaf_id"] = torch.tensor(kwargs["selected_leaf_id"])
    pipe = types.SimpleNamespace(debug=False, compute_cov3D_python=False, convert_SHs_python=False)
    for k, v in c.get("pipe", {}).items():
        setattr(pipe, k, v)
    bg = torch.tensor([0.2, 0.1, 0.3])
    return dict(params=params, cam=cam, pipe=pipe, bg=bg, kwargs=kwargs, rng_seed=c["rng_seed"], grads=c["grads"],
                expect_rescale=c["expect_rescale"])


def move_kwargs(kwargs: dict, device):
    return {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in kwargs.items()}


def _weights(key: str, index: int, shape, device):
    g = torch.Generator().manual_seed(zlib.crc32(f"{key}/{index}".encode()) & 0x7FFFFFFF)
    return torch.randn(tuple(shape), generator=g).to(device)


def fixed_loss(out: dict):
    """A fixed linear functional of every image the call returned (weights seeded by key and list position; the
    un-normalised depth map gets a small weight).  None when nothing differentiable came back."""
    loss = None
    for k in TENSOR_KEYS:
        v = out.get(k)
        if torch.is_tensor(v) and v.dtype.is_floating_point and v.requires_grad:
            term = (v * _weights(k, 0, v.shape, v.device)).sum() * (0.05 if k == "depth" else 1.0)
            loss = term if loss is None else loss + term
    for k in LIST_KEYS:
        for i, v in enumerate(out.get(k) or []):
            if v.requires_grad:
                term = (v * _weights(k, i, v.shape, v.device)).sum()
                loss = term if loss is None else loss + term
    return loss
