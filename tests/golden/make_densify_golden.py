"""Generate tests/golden/densify_golden.npz by RUNNING THE REFERENCE's own GaussianModel methods on the CPU.

Run in the build container only (``python tests/golden/make_densify_golden.py``): /root/reference never travels, only
the vectors do.  scene/gaussian_model.py cannot be imported as a module here (it imports `plyfile` and
`simple_knn`), so the class body is compiled from its source text (ast) keeping only the methods this path replaces:

    setup_functions, __init__, get_scaling, get_opacity, get_xyz, replace_tensor_to_optimizer, _prune_optimizer,
    prune_points, cat_tensors_to_optimizer, densification_postfix, densify_and_split, densify_and_clone,
    densify_and_prune, add_densification_stats, reset_opacity            (scene/gaussian_model.py:38-62,122-149,300-303,357-514)

with `build_rotation` / `inverse_sigmoid` compiled the same way from utils/general_utils.py, and device="cuda"
mapped to the CPU for the duration of the run.  The optimizer is the reference's own
``torch.optim.Adam(l, lr=0.0, eps=1e-15)`` over its seven named groups (:216-230), stepped a few times so that
the moments are populated.

The fixture is COMPACT: column 0 of `_ins_feat` carries the row number (exact in fp32), which the reference copies
into every clone and split child, so the whole result is described by its row map + which rows are new + the
children's xyz / scaling + the `samples` the reference drew; the test rebuilds the expected tensors from the
seeded inputs.  A float64 checksum of every output tensor pins the rest.
"""
import ast
import os

import numpy as np
import torch
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
REF_MODEL = "/root/reference/scene/gaussian_model.py"
REF_UTILS = "/root/reference/utils/general_utils.py"
METHODS = ["setup_functions", "__init__", "get_scaling", "get_opacity", "get_xyz", "replace_tensor_to_optimizer",
           "_prune_optimizer", "prune_points", "cat_tensors_to_optimizer", "densification_postfix", "densify_and_split",
           "densify_and_clone", "densify_and_prune", "add_densification_stats", "reset_opacity"]
GROUPS = [("xyz", (3,), 1.6e-4), ("f_dc", (1, 3), 2.5e-3), ("f_rest", (15, 3), 1.25e-4), ("opacity", (1,), 0.05),
          ("scaling", (3,), 5e-3), ("rotation", (4,), 1e-3), ("ins_feat", (6,), 1e-3)]
ATTR = {"xyz": "_xyz", "f_dc": "_features_dc", "f_rest": "_features_rest", "opacity": "_opacity", "scaling": "_scaling",
        "rotation": "_rotation", "ins_feat": "_ins_feat"}
# (seed, P, max_grad, extent, size_threshold)
CASES = [(0, 500, 0.6, 4.0, 20), (1, 700, 0.5, 3.0, None), (2, 300, 5.0, 4.0, 20)]
ADAM_STEPS = 3
PERCENT_DENSE = 0.01


def case_inputs(seed, P):
    """Seeded model state (shared with the tests): parameters, per-step gradients, statistics."""
    g = torch.Generator().manual_seed(7000 + seed)
    params = {n: torch.randn(P, *shape, generator=g) for n, shape, _ in GROUPS}
    params["scaling"] = torch.randn(P, 3, generator=g) * 1.2 - 3.0          # log-scales: some above 0.01 * extent, some > 0.1 * extent
    params["opacity"] = torch.randn(P, 1, generator=g) * 3.0                # logits: some below sigmoid^-1(0.005)
    params["ins_feat"][:, 0] = torch.arange(P, dtype=torch.float32)         # the row tag
    grads = [{n: torch.randn(P, *shape, generator=g) * 0.1 for n, shape, _ in GROUPS} for _ in range(ADAM_STEPS)]
    for gr in grads:
        gr["ins_feat"][:, 0] = 0.0                                          # zero gradient: Adam leaves the tag untouched
    accum = torch.rand(P, 1, generator=g) * 3.0
    denom = torch.randint(0, 4, (P, 1), generator=g).float()                 # zeros -> NaN grads -> 0 (:490)
    radii = torch.rand(P, generator=g) * 40.0
    vs_grad = torch.randn(P, 3, generator=g)
    vis = torch.rand(P, generator=g) < 0.6
    prune_mask = torch.rand(P, generator=g) < 0.3
    return params, grads, accum, denom, radii, vs_grad, vis, prune_mask


def load_reference():
    ns = {"torch": torch, "nn": nn, "np": np}
    tree = ast.parse(open(REF_UTILS).read())
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in ("build_rotation", "inverse_sigmoid"):
            exec(compile(ast.Module(body=[node], type_ignores=[]), REF_UTILS, "exec"), ns)
    tree = ast.parse(open(REF_MODEL).read())
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "GaussianModel"][0]
    cls.body = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name in METHODS]
    assert sorted(n.name for n in cls.body) == sorted(METHODS), [n.name for n in cls.body]
    exec(compile(ast.Module(body=[cls], type_ignores=[]), REF_MODEL, "exec"), ns)
    return ns["GaussianModel"]


class _CudaToCpu:
    """device="cuda" -> cpu for the factory functions the reference's methods call; records torch.normal draws."""
    NAMES = ("zeros", "ones", "empty", "full", "tensor")

    def __init__(self, gen):
        self.gen, self.samples = gen, []

    def __enter__(self):
        self.saved = {n: getattr(torch, n) for n in self.NAMES + ("normal",)}
        for n in self.NAMES:
            f = self.saved[n]
            setattr(torch, n, (lambda f: lambda *a, **k: f(*a, **{**k, "device": "cpu"} if k.get("device") == "cuda" else k))(f))
        normal = self.saved["normal"]

        def rec_normal(mean, std, **k):
            out = normal(mean=mean, std=std, generator=self.gen)
            self.samples.append(out.detach().clone())
            return out
        torch.normal = rec_normal
        return self

    def __exit__(self, *exc):
        for n, f in self.saved.items():
            setattr(torch, n, f)


def build_model(GaussianModel, params, grads):
    m = GaussianModel(3)
    for n, _, _ in GROUPS:
        setattr(m, ATTR[n], nn.Parameter(params[n].clone().requires_grad_(True)))
    l = [{"params": [getattr(m, ATTR[n])], "lr": lr, "name": n} for n, _, lr in GROUPS]
    m.optimizer = torch.optim.Adam(l, lr=0.0, eps=1e-15)                    # scene/gaussian_model.py:230
    for gr in grads:
        for n, _, _ in GROUPS:
            getattr(m, ATTR[n]).grad = gr[n].clone()
        m.optimizer.step()
    m.percent_dense = PERCENT_DENSE
    return m


def snapshot(m):
    out = {}
    for n, _, _ in GROUPS:
        p = getattr(m, ATTR[n])
        st = m.optimizer.state[p]
        out[n] = (p.detach().clone(), st["exp_avg"].clone(), st["exp_avg_sq"].clone())
    return out


def describe(snap, prefix, store):
    """compact description of a post-operation state (see module docstring)"""
    tag = snap["ins_feat"][0][:, 0]
    src = tag.round().to(torch.int64)
    assert torch.equal(src.float(), tag)
    is_new = (snap["xyz"][2].abs().sum(dim=1) == 0)                                    # moments zeroed -> new row
    store[prefix + "_src"] = src.numpy().astype(np.int32)
    store[prefix + "_new"] = is_new.numpy()
    store[prefix + "_xyz"] = snap["xyz"][0].numpy()[is_new.numpy()]
    store[prefix + "_scaling"] = snap["scaling"][0].numpy()[is_new.numpy()]
    store[prefix + "_checksum"] = np.array([float(t.double().sum()) for n, _, _ in GROUPS for t in snap[n]])


def main():
    GaussianModel = load_reference()
    store = {"cases": np.array([(s, p, mg, ex, -1 if st is None else st) for s, p, mg, ex, st in CASES], dtype=np.float64)}
    for seed, P, max_grad, extent, size_threshold in CASES:
        params, grads, accum, denom, radii, vs_grad, vis, prune_mask = case_inputs(seed, P)
        k = f"s{seed}"
        gen = torch.Generator().manual_seed(9000 + seed)
        with _CudaToCpu(gen) as shim:
            # --- A: prune_points -------------------------------------------------------------------------------
            m = build_model(GaussianModel, params, grads)
            m.xyz_gradient_accum, m.denom, m.max_radii2D = accum.clone(), denom.clone(), radii.clone()
            m.prune_points(prune_mask.clone())
            describe(snapshot(m), k + "_prune", store)
            store[k + "_prune_stats"] = np.concatenate([m.xyz_gradient_accum.numpy().ravel(), m.denom.numpy().ravel(),
                                                       m.max_radii2D.numpy().ravel()])
            # --- B: add_densification_stats + densify_and_prune --------------------------------------------------
            m = build_model(GaussianModel, params, grads)
            m.xyz_gradient_accum, m.denom, m.max_radii2D = accum.clone(), denom.clone(), radii.clone()
            vsp = torch.zeros(P, 3); vsp.grad = vs_grad.clone()
            m.add_densification_stats(vsp, vis)
            store[k + "_stats_accum"] = m.xyz_gradient_accum.numpy().copy()
            store[k + "_stats_denom"] = m.denom.numpy().copy()
            shim.samples.clear()
            m.densify_and_prune(max_grad, 0.005, extent, size_threshold)
            assert len(shim.samples) == 1
            store[k + "_samples"] = shim.samples[0].numpy()
            describe(snapshot(m), k + "_densify", store)
            assert float(m.xyz_gradient_accum.abs().sum()) == 0 and float(m.max_radii2D.abs().sum()) == 0
            store[k + "_densify_n"] = np.array([m.get_xyz.shape[0], m.denom.shape[0], m.max_radii2D.shape[0]])
            # --- C: reset_opacity ----------------------------------------------------------------------------
            m = build_model(GaussianModel, params, grads)
            m.reset_opacity()
            st = m.optimizer.state[m._opacity]
            store[k + "_reset_opacity"] = m._opacity.detach().numpy().copy()
            assert float(st["exp_avg"].abs().sum()) == 0 and float(st["exp_avg_sq"].abs().sum()) == 0
        print("case", seed, P, "->", {n: int(v) for n, v in zip(("rows", "new"), (len(store[k + "_densify_src"]), store[k + "_densify_new"].sum()))},
              "samples", store[k + "_samples"].shape)
    path = os.path.join(HERE, "densify_golden.npz")
    np.savez_compressed(path, **store)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
