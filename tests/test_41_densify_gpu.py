"""GPU: opengaussian_amd.densify (HIP row-map kernels, include/ogs_optim.h) against vectors produced by RUNNING the
reference's own GaussianModel methods (tests/golden/make_densify_golden.py -> densify_golden.npz):
prune_points, add_densification_stats, densify_and_prune (clone + split + prune), reset_opacity
(scene/gaussian_model.py:300-303,357-514).

The fixture is compact (row map + new-row flags + the children's xyz / scaling + the reference's `samples` + float64
checksums); the expected tensors are rebuilt here from the seeded inputs: an old row is an exact copy of parameters
AND Adam moments, a new row copies its parent's parameters with zero moments, a split child additionally gets the
reference's xyz / scaling.  Copies are compared bit-exact; the children's xyz / scaling to 1e-6 (exp / log / the
3x3 product are evaluated on the device)."""
import os

import numpy as np
import pytest
import torch

from tests.golden.make_densify_golden import ADAM_STEPS, ATTR, GROUPS, PERCENT_DENSE, case_inputs

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "densify_golden.npz")


def _cases():
    g = np.load(GOLD)
    return [(int(r[0]), int(r[1]), float(r[2]), float(r[3]), None if r[4] < 0 else int(r[4])) for r in g["cases"]]


def _state(seed, P, dev, fused=True):
    """The model state the golden run started from: the seeded parameters after ADAM_STEPS steps of torch.optim.Adam on
    the CPU (torch's own optimizer, exactly what the generator ran), installed into a GPU optimizer."""
    from opengaussian_amd.densify import DensifyState
    from opengaussian_amd.optim import FusedAdam
    params, grads, accum, denom, radii, vs_grad, vis, prune_mask = case_inputs(seed, P)
    cpu = {n: torch.nn.Parameter(params[n].clone()) for n, _, _ in GROUPS}
    ref = torch.optim.Adam([{"params": [cpu[n]], "lr": lr, "name": n} for n, _, lr in GROUPS], lr=0.0, eps=1e-15)
    for gr in grads:
        for n, _, _ in GROUPS:
            cpu[n].grad = gr[n].clone()
        ref.step()
    gpu = {n: torch.nn.Parameter(cpu[n].detach().to(dev)) for n, _, _ in GROUPS}
    Opt = FusedAdam if fused else torch.optim.Adam
    opt = Opt([{"params": [gpu[n]], "lr": lr, "name": n} for n, _, lr in GROUPS], lr=0.0, eps=1e-15)
    for n, _, _ in GROUPS:
        st = ref.state[cpu[n]]
        opt.state[gpu[n]] = {"step": torch.tensor(float(ADAM_STEPS)), "exp_avg": st["exp_avg"].to(dev),
                             "exp_avg_sq": st["exp_avg_sq"].to(dev)}
    before = {n: (gpu[n].detach().clone(), opt.state[gpu[n]]["exp_avg"].clone(), opt.state[gpu[n]]["exp_avg_sq"].clone())
              for n, _, _ in GROUPS}
    state = DensifyState(opt, accum.to(dev), denom.to(dev), radii.to(dev), PERCENT_DENSE)
    return state, before, (vs_grad.to(dev), vis.to(dev), prune_mask.to(dev), accum, denom, radii)


def _check_against_map(state, before, src, is_new, child_xyz=None, child_scaling=None):
    """every tensor of the optimizer == the reference's row map applied to the state before the call"""
    P = state.params()
    dev = P["xyz"].device
    src_t = torch.from_numpy(src.astype(np.int64)).to(dev)
    new_t = torch.from_numpy(is_new).to(dev)
    for n, _, _ in GROUPS:
        p = P[n]
        st = state.optimizer.state[p]
        assert p.shape[0] == len(src) and p.requires_grad and p.is_leaf, n
        want_p = before[n][0][src_t]
        if child_xyz is not None and n in ("xyz", "scaling"):
            want_new = torch.from_numpy(child_xyz if n == "xyz" else child_scaling).to(dev)
            assert torch.equal(p.detach()[~new_t], want_p[~new_t]), n
            torch.testing.assert_close(p.detach()[new_t], want_new, rtol=1e-6, atol=1e-6)
        else:
            assert torch.equal(p.detach(), want_p), n                        # parameters: exact copies of the parent row
        for k, m in ((1, "exp_avg"), (2, "exp_avg_sq")):
            want_m = before[n][k][src_t] * (~new_t).reshape(-1, *([1] * (p.dim() - 1)))
            assert torch.equal(st[m], want_m), (n, m)                        # moments: copied for old rows, zero for new
        assert float(st["step"]) == float(ADAM_STEPS)


@pytest.mark.parametrize("seed,P,max_grad,extent,size_threshold", _cases())
@pytest.mark.parametrize("fused", [True, False])
def test_densification_matches_reference_golden(gpu_device, seed, P, max_grad, extent, size_threshold, fused):
    from opengaussian_amd import densify
    gold = np.load(GOLD)
    key = lambda name: gold[f"s{seed}_{name}"]
    dev = gpu_device

    # ---- prune_points (:391-410) ---------------------------------------------------------------------------------
    state, before, (vs_grad, vis, prune_mask, accum, denom, radii) = _state(seed, P, dev, fused)
    params = densify.prune_points(state, prune_mask)
    assert set(params) == {n for n, _, _ in GROUPS}
    src = key("prune_src")
    np.testing.assert_array_equal(src, np.nonzero(~prune_mask.cpu().numpy())[0])
    _check_against_map(state, before, src, np.zeros(len(src), bool))
    stats = np.concatenate([state.xyz_gradient_accum.cpu().numpy().ravel(), state.denom.cpu().numpy().ravel(),
                            state.max_radii2D.cpu().numpy().ravel()])
    np.testing.assert_array_equal(stats, key("prune_stats"))
    # the optimizer still steps after the surgery
    for p in state.params().values():
        p.grad = torch.ones_like(p)
    state.optimizer.step()

    # ---- add_densification_stats (:512-514) + densify_and_prune (:488-508) -------------------------------------------
    state, before, (vs_grad, vis, prune_mask, accum, denom, radii) = _state(seed, P, dev, fused)
    densify.add_densification_stats(state, vs_grad, vis)
    np.testing.assert_allclose(state.xyz_gradient_accum.cpu().numpy(), key("stats_accum"), rtol=1e-6, atol=1e-7)
    np.testing.assert_array_equal(state.denom.cpu().numpy(), key("stats_denom"))
    samples = torch.from_numpy(key("samples")).to(dev)
    params = densify.densify_and_prune(state, max_grad, 0.005, extent, size_threshold, samples=samples)
    src, is_new = key("densify_src"), key("densify_new")
    plan = state.last_plan
    np.testing.assert_array_equal(plan["src_row"].cpu().numpy(), src)          # same rows, same order as the reference
    np.testing.assert_array_equal(plan["kind"].cpu().numpy() != 0, is_new)
    assert plan["split_parents_selected"] * 2 == samples.shape[0]
    # new rows that are clones keep the parent's xyz; children get the reference's values: the golden holds xyz /
    # scaling of ALL new rows, clones included
    _check_against_map(state, before, src, is_new, key("densify_xyz"), key("densify_scaling"))
    n_out = len(src)
    assert tuple(key("densify_n")) == (n_out, n_out, n_out)
    assert state.xyz_gradient_accum.shape == (n_out, 1) and state.denom.shape == (n_out, 1) and state.max_radii2D.shape == (n_out,)
    assert float(state.xyz_gradient_accum.abs().sum()) == 0 and float(state.denom.abs().sum()) == 0 and float(state.max_radii2D.abs().sum()) == 0
    got_sum = np.array([float(t.double().sum()) for n, _, _ in GROUPS
                        for t in (params[n].detach(), state.optimizer.state[params[n]]["exp_avg"],
                                  state.optimizer.state[params[n]]["exp_avg_sq"])])
    np.testing.assert_allclose(got_sum, key("densify_checksum"), rtol=1e-6, atol=1e-6)
    kinds = plan["kind"].cpu().numpy()
    if seed != 2:
        assert (kinds == 1).any() and (kinds == 2).any() and (kinds == 0).any()     # the case really clones, splits and keeps
    for p in state.params().values():
        p.grad = torch.ones_like(p)
    state.optimizer.step()

    # ---- reset_opacity (:300-303) ---------------------------------------------------------------------------------------
    state, before, _ = _state(seed, P, dev, fused)
    out = densify.reset_opacity(state)
    torch.testing.assert_close(out["opacity"].detach().cpu(), torch.from_numpy(key("reset_opacity")), rtol=1e-6, atol=1e-6)
    st = state.optimizer.state[out["opacity"]]
    assert float(st["exp_avg"].abs().sum()) == 0 and float(st["exp_avg_sq"].abs().sum()) == 0


def test_cat_and_primitives(gpu_device):
    """cat_tensors_to_optimizer (:412-433) and the row-gather primitive on its own: -1 rows are zeros, the unfused
    sequence clone -> prune built from the primitives equals index arithmetic done by torch."""
    from opengaussian_amd import densify
    dev = gpu_device
    state, before, _ = _state(0, 500, dev)
    g = torch.Generator().manual_seed(1)
    ext = {n: torch.randn(37, *shape, generator=g).to(dev) for n, shape, _ in GROUPS}
    out = densify.cat_tensors_to_optimizer(state.optimizer, ext)
    for n, _, _ in GROUPS:
        st = state.optimizer.state[out[n]]
        assert torch.equal(out[n].detach(), torch.cat((before[n][0], ext[n])))
        assert torch.equal(st["exp_avg"], torch.cat((before[n][1], torch.zeros_like(ext[n]))))
        assert torch.equal(st["exp_avg_sq"], torch.cat((before[n][2], torch.zeros_like(ext[n]))))
    t = [torch.randn(1000, 45, generator=g).to(dev), torch.randn(1000, generator=g).to(dev).reshape(1000, 1),
         torch.randn(1000, 15, 3, generator=g).to(dev)]
    rows = torch.randint(-1, 1000, (2500,), generator=g).to(dev)
    kind = (torch.arange(2500, device=dev) % 3 == 0).to(torch.uint8)
    outs = densify.gather_rows(t, rows, kind, zero_new={1})
    for i, (src, o) in enumerate(zip(t, outs)):
        want = src[rows.clamp_min(0).long()] * (rows >= 0).reshape(-1, *([1] * (src.dim() - 1)))
        if i == 1:
            want = want * (kind == 0).reshape(-1, 1)
        assert torch.equal(o, want)
    assert densify.gather_rows(t, rows[:0])[0].shape == (0, 45)
    with pytest.raises(RuntimeError):
        densify.gather_rows([torch.zeros(4, 3)], torch.zeros(2, dtype=torch.int32, device=dev))      # CPU tensor: no CPU path


def test_densify_and_prune_draws_its_own_samples(gpu_device):
    """samples=None: the split offsets are drawn as the reference draws them (torch.normal, std = the selected
    parents' scaling repeated for the two copies, :443-445): same row map as with explicit samples, children within
    a few sigma of their parent."""
    from opengaussian_amd import densify
    dev = gpu_device
    seed, P, max_grad, extent, size_threshold = _cases()[0]
    gold = np.load(GOLD)
    state, before, (vs_grad, vis, *_rest) = _state(seed, P, dev)
    densify.add_densification_stats(state, vs_grad, vis)
    gen = torch.Generator(device=dev).manual_seed(3)
    densify.densify_and_prune(state, max_grad, 0.005, extent, size_threshold, generator=gen)
    plan = state.last_plan
    np.testing.assert_array_equal(plan["src_row"].cpu().numpy(), gold[f"s{seed}_densify_src"])
    child = plan["kind"] >= 2
    par = plan["src_row"][child].long()
    off = (state.params()["xyz"].detach()[child] - before["xyz"][0][par]).norm(dim=1)
    smax = torch.exp(before["scaling"][0][par]).max(dim=1).values
    assert float((off / smax).max()) < 8.0 and float(off.min()) > 0.0
