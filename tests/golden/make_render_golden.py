"""Generate tests/golden/render_golden.npz by RUNNING THE REFERENCE's own gaussian_renderer.render() on the CPU.

Run in the build container only (``python tests/golden/make_render_golden.py``): /root/reference never travels, only
the vectors do.  /root/reference/gaussian_renderer/__init__.py:22-373 is pure Python; it fails to import here only
because of what it imports (:15-20).  This script loads that very file (importlib, by path) with ``sys.modules``
stand-ins for exactly those imports:

    ashawkey_diff_gaussian_rasterization  -> the NamedTuple of 12 fields render() fills (:55-68) and a CPU
                                             ``GaussianRasterizer`` backed by oracle/raster_oracle.py: fp32 NumPy
                                             preprocess + binning + fp32 per-tile blend forward, float64 autograd through
                                             the oracle's differentiable restatement backward (fixed binning)
    scene.gaussian_model                  -> a module with an empty ``GaussianModel`` (only a type annotation, :22)
    utils.sh_utils                        -> the REFERENCE's own file, loaded by path (eval_sh, :92-97)
    utils.opengs_utlis                    -> the REFERENCE's own file, loaded by path with an empty ``bitarray`` stand-in
                                             (render() uses nothing of it; the star import must resolve, :18)
    pytorch3d.ops.knn_points              -> squared distances to the K nearest neighbours, ascending (cdist + topk),
                                             what knn_points(x, x, K=K).dists holds (:299-304)

and with device "cuda" mapped to the CPU (``Tensor.cuda`` = identity, ``zeros_like(device="cuda")``).  The model handed to
render() is the REFERENCE's own ``GaussianModel`` getters (scene/gaussian_model.py:47-62,122-172, class body compiled
from source text because the module imports `plyfile`); the camera is a plain attribute holder.

WHAT THIS PINS: render()'s orchestration -- which passes are issued on which subsets with which scales, the RNG draws,
the (ins_feat + 1) / 2 mapping, the cluster / leaf loops with their filters, the kNN filter, the 14-key dict -- and the
gradients of a fixed loss through all of it.  The rasterizer arithmetic underneath is the oracle's on both sides and
stays PARITY UNPINNED (the reference's CUDA rasterizer is not in /root/reference).

Stored per case: every entry of the result dict, and (training cases) the gradient of
tests/golden/render_cases.fixed_loss w.r.t. the seven raw parameter tensors and ``viewspace_points``.
"""
import ast
import collections
import importlib.util
import os
import sys
import types
from typing import NamedTuple

import numpy as np
import torch
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import raster_oracle as ro           # noqa: E402  (test infrastructure: the checker)
from tests.golden import render_cases as rc      # noqa: E402

REF = "/root/reference"


# ---------------------------------------------------------------------------------------------------------------------
# stand-in for the un-vendored rasterizer package: same Python surface, oracle inside
# ---------------------------------------------------------------------------------------------------------------------
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


class _OracleRasterize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, opacities, shs, colors_precomp, scales, rotations, cov3D_precomp, rs):
        n = lambda t: None if t is None else t.detach().to(torch.float32).contiguous().numpy()
        W, H = int(rs.image_width), int(rs.image_height)
        inp = dict(means3D=n(means3D), opacities=n(opacities), viewmatrix=n(rs.viewmatrix), projmatrix=n(rs.projmatrix),
                   campos=n(rs.campos), scales=n(scales), rotations=n(rotations), cov3D_precomp=n(cov3D_precomp), shs=n(shs),
                   colors_precomp=n(colors_precomp))
        P = inp["means3D"].shape[0]
        C = 3 if colors_precomp is None else colors_precomp.shape[1]
        ctx.rs, ctx.inp, ctx.P = rs, inp, P
        ctx.present = [t is not None for t in (means3D, means2D, opacities, shs, colors_precomp, scales, rotations, cov3D_precomp)]
        if P == 0:
            ctx.binning = None
            z = lambda c: torch.zeros(c, H, W)
            return z(C), torch.zeros(0, dtype=torch.int32), z(1), z(1)
        ref = ro.render_forward(W=W, H=H, tanfovx=rs.tanfovx, tanfovy=rs.tanfovy, bg=n(rs.bg), scale_modifier=rs.scale_modifier,
                                sh_degree=rs.sh_degree, **inp)
        ctx.binning = ref["binning"]
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
        radii = t(ref["geom"].radii.astype(np.int32))
        ctx.mark_non_differentiable(radii)
        return t(ref["color"]), radii, t(ref["depth"]), t(ref["alpha"])

    @staticmethod
    def backward(ctx, gC, gR, gD, gA):
        rs = ctx.rs
        if ctx.binning is None or ctx.binning.num_rendered == 0:
            grads = {}
        else:
            W, H = int(rs.image_width), int(rs.image_height)
            z = lambda g, c: np.zeros((c, H, W)) if g is None else g.double().numpy()
            C = 3 if ctx.inp["colors_precomp"] is None else ctx.inp["colors_precomp"].shape[1]
            with torch.enable_grad():
                grads = ro.render_backward_f64({k: v for k, v in ctx.inp.items() if v is not None}, ctx.binning, W, H,
                                               rs.tanfovx, rs.tanfovy, rs.bg.detach().double().numpy(), z(gC, C), z(gD, 1),
                                               z(gA, 1), scale_modifier=rs.scale_modifier, sh_degree=rs.sh_degree)
        names = ["means3D", "means2D", "opacities", "shs", "colors_precomp", "scales", "rotations", "cov3D_precomp"]
        shapes = {"means2D": (ctx.P, 3)}
        out = []
        for name, present in zip(names, ctx.present):
            if not present:
                out.append(None)
                continue
            g = grads.get(name)
            if g is None:
                shape = shapes.get(name) or ctx.inp[name].shape
                out.append(torch.zeros(shape, dtype=torch.float32))
            else:
                ref_shape = shapes.get(name) or ctx.inp[name].shape
                out.append(torch.from_numpy(g.astype(np.float32)).reshape(ref_shape))
        return (*out, None)


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        return _OracleRasterize.apply(means3D, means2D, opacities, shs, colors_precomp, scales, rotations, cov3D_precomp,
                                      self.raster_settings)


_KNN = collections.namedtuple("KNN", "dists idx knn")


def knn_points(p1, p2, K=1, **_):
    d2 = ((p1[:, :, None, :] - p2[:, None, :, :]) ** 2).sum(-1)          # direct differences, fp32, as pytorch3d's kernel
    vals, idx = torch.topk(d2, K, dim=-1, largest=False, sorted=True)
    return _KNN(vals, idx, None)


# ---------------------------------------------------------------------------------------------------------------------
def _load_by_path(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


class _CudaToCpu:
    """device "cuda" -> cpu while the reference runs (no GPU in the build container)."""
    NAMES = ("zeros", "ones", "empty", "full", "tensor", "zeros_like", "ones_like", "rand")

    def __enter__(self):
        self.saved_cuda = torch.Tensor.cuda
        torch.Tensor.cuda = lambda self_, *a, **k: self_
        self.saved = {n: getattr(torch, n) for n in self.NAMES}
        for n, f in self.saved.items():
            setattr(torch, n, (lambda f: lambda *a, **k: f(*a, **({**k, "device": "cpu"} if k.get("device") == "cuda" else k)))(f))
        return self

    def __exit__(self, *exc):
        torch.Tensor.cuda = self.saved_cuda
        for n, f in self.saved.items():
            setattr(torch, n, f)


def load_reference_render():
    mods = {}
    m = types.ModuleType("ashawkey_diff_gaussian_rasterization")
    m.GaussianRasterizationSettings, m.GaussianRasterizer = GaussianRasterizationSettings, GaussianRasterizer
    mods["ashawkey_diff_gaussian_rasterization"] = m
    scene = types.ModuleType("scene"); scene.__path__ = []
    gm = types.ModuleType("scene.gaussian_model"); gm.GaussianModel = type("GaussianModel", (), {})
    mods["scene"], mods["scene.gaussian_model"] = scene, gm
    utils = types.ModuleType("utils"); utils.__path__ = []
    mods["utils"] = utils
    ba = types.ModuleType("bitarray"); ba.bitarray = type("bitarray", (), {})
    mods["bitarray"] = ba
    p3d = types.ModuleType("pytorch3d"); p3d.__path__ = []
    ops = types.ModuleType("pytorch3d.ops"); ops.knn_points = knn_points
    p3d.ops = ops
    mods["pytorch3d"], mods["pytorch3d.ops"] = p3d, ops
    sys.modules.update(mods)
    _load_by_path("utils.sh_utils", os.path.join(REF, "utils", "sh_utils.py"))
    _load_by_path("utils.opengs_utlis", os.path.join(REF, "utils", "opengs_utlis.py"))
    ref = _load_by_path("ref_gaussian_renderer", os.path.join(REF, "gaussian_renderer", "__init__.py"))
    return ref.render


MODEL_METHODS = ["setup_functions", "__init__", "get_scaling", "get_rotation", "get_xyz", "get_features", "get_opacity",
                 "get_ins_feat", "get_covariance"]


def load_reference_model():
    """The reference's GaussianModel reduced to its constructor + the getters render() reads (source text, ast)."""
    ns = {"torch": torch, "nn": nn, "np": np}
    tree = ast.parse(open(os.path.join(REF, "utils", "general_utils.py")).read())
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in ("build_rotation", "build_scaling_rotation", "strip_lowerdiag",
                                                               "strip_symmetric", "inverse_sigmoid"):
            exec(compile(ast.Module(body=[node], type_ignores=[]), "general_utils.py", "exec"), ns)
    tree = ast.parse(open(os.path.join(REF, "scene", "gaussian_model.py")).read())
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "GaussianModel"][0]
    cls.body = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name in MODEL_METHODS]
    assert sorted(n.name for n in cls.body) == sorted(MODEL_METHODS), [n.name for n in cls.body]
    exec(compile(ast.Module(body=[cls], type_ignores=[]), "gaussian_model.py", "exec"), ns)
    return ns["GaussianModel"]


def reference_model(GaussianModel, params):
    m = GaussianModel(3)
    for n in rc.PARAM_NAMES:
        setattr(m, n, nn.Parameter(params[n].clone().requires_grad_(True)))
    m.active_sh_degree = 3
    rc.attach_quantized(m, params)
    return m


def np_of(v):
    return v.detach().cpu().numpy()


def store_result(store, k, out):
    for key in rc.TENSOR_KEYS + ("radii", "visibility_filter", "cluster_occur"):
        v = out[key]
        if v is None:
            store[f"{k}/{key}/none"] = np.array(1)
        elif isinstance(v, list):                        # the reference leaves an EMPTY LIST when nothing was kept (:233,355)
            assert len(v) == 0
            store[f"{k}/{key}/emptylist"] = np.array(1)
        else:
            store[f"{k}/{key}"] = np_of(v)
    for key in rc.LIST_KEYS:
        v = out[key]
        if v is None:
            store[f"{k}/{key}/none"] = np.array(1)
        else:
            store[f"{k}/{key}/len"] = np.array(len(v))
            for i, img in enumerate(v):
                store[f"{k}/{key}/{i}"] = np_of(img)
    v = out["occured_leaf_id"]
    if v is None:
        store[f"{k}/occured_leaf_id/none"] = np.array(1)
    else:
        store[f"{k}/occured_leaf_id"] = np.array([int(x) for x in v], dtype=np.int64)
    vp = out["viewspace_points"]
    store[f"{k}/viewspace_points_shape"] = np.array(vp.shape)


def main():
    render = load_reference_render()
    GaussianModel = load_reference_model()
    store = {}
    for name in rc.CASES:
        case = rc.build(name)
        with _CudaToCpu():
            pc = reference_model(GaussianModel, case["params"])
            # our stand-in model must present render() with the same activated values as the reference's getters
            tiny = rc.TinyModel(case["params"])
            for getter in ("get_xyz", "get_scaling", "get_rotation", "get_opacity", "get_features"):
                assert torch.equal(getattr(pc, getter), getattr(tiny, getter)), getter
            assert torch.equal(pc.get_ins_feat(), tiny.get_ins_feat()) and torch.equal(pc.get_ins_feat(origin=True), tiny.get_ins_feat(origin=True))
            assert torch.allclose(pc.get_covariance(1.0), tiny.get_covariance(1.0), atol=1e-7)
            store[f"{name}/inputs_checksum"] = np.array([float(case["params"][n].double().sum()) for n in sorted(case["params"])])
            torch.manual_seed(case["rng_seed"])
            coin = float(torch.rand(1))
            assert (coin > 0.5 and case["kwargs"].get("rescale", True)) == case["expect_rescale"], (name, coin)
            torch.manual_seed(case["rng_seed"])
            grad_ctx = torch.enable_grad() if case["grads"] else torch.no_grad()
            with grad_ctx:
                out = render(case["cam"], pc, case["pipe"], case["bg"], 1, **case["kwargs"])
                store_result(store, name, out)
                if case["grads"]:
                    loss = rc.fixed_loss(out)
                    loss.backward()
                    store[f"{name}/loss"] = np.array(float(loss.detach()))
                    for n in rc.PARAM_NAMES:
                        g = getattr(pc, n).grad
                        store[f"{name}/grad/{n}"] = np.zeros_like(np_of(getattr(pc, n))) if g is None else np_of(g)
                    vg = out["viewspace_points"].grad
                    store[f"{name}/grad/viewspace_points"] = np_of(vg)
        desc = {k: (tuple(v.shape) if torch.is_tensor(v) else (len(v) if isinstance(v, list) else v)) for k, v in out.items()}
        print(name, "coin %.4f" % coin, desc)
    np.savez_compressed(os.path.join(HERE, "render_golden.npz"), **store)
    print("wrote", len(store), "arrays,", os.path.getsize(os.path.join(HERE, "render_golden.npz")), "bytes")


if __name__ == "__main__":
    main()
