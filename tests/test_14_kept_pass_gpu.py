"""GPU: the frozen-geometry cache (rasterizer.KEPT_PASSES, ogs_raster_forward_reblend).

From stage 1 on the reference trains `_ins_feat` alone (train.py:431-436) and renders the stage-1 views without the random
footprint rescale (train.py:346-350): a camera's pass is kept, later passes blend the kept streams with the current feature channels.
The contract under test: a re-blend returns, BIT FOR BIT, what a full pass over the same inputs returns (images, depth, alpha,
radii), its feature gradient is the full pass' features-only gradient, and a key that changed never hits."""
import types

import pytest
import torch

from tests import helpers

pytestmark = pytest.mark.gpu


class ReferenceShapedGaussians:
    """The attributes of scene/gaussian_model.py:GaussianModel that render() and the cache key read (:122-169): raw parameters
    `_xyz, _scaling, _rotation, _opacity, _features_dc, _features_rest, _ins_feat` and the activated getters."""

    def __init__(self, sc, dev):
        self._xyz = sc.means3D.to(dev)
        self._scaling = torch.log(sc.scales.to(dev))
        self._rotation = sc.rotations.to(dev)
        self._opacity = torch.logit(sc.opacities.to(dev).clamp(1e-4, 1 - 1e-4))
        shs = sc.shs.to(dev)
        self._features_dc = shs[:, :1].contiguous()
        self._features_rest = shs[:, 1:].contiguous()
        self._ins_feat = (sc.ins_feat.to(dev) * 2 - 1).requires_grad_(True)
        self.active_sh_degree = 3
        self.max_sh_degree = 3

    get_xyz = property(lambda s: s._xyz)
    get_scaling = property(lambda s: torch.exp(s._scaling))
    get_rotation = property(lambda s: torch.nn.functional.normalize(s._rotation))
    get_opacity = property(lambda s: torch.sigmoid(s._opacity))
    get_features = property(lambda s: torch.cat((s._features_dc, s._features_rest), dim=1))

    def get_ins_feat(self, origin=False):
        return torch.nn.functional.normalize(self._ins_feat, dim=1)


@pytest.fixture()
def kept(gpu_device):
    from opengaussian_amd import rasterizer as R
    saved = R.KEPT_PASSES
    R.KEPT_PASSES = R.KeptPasses(budget_bytes=4 << 30)
    yield R
    R.KEPT_PASSES = saved


def _scene(dev, P=6000, W=200, H=136, f=150.0, seed=5):
    sc, cam = helpers.tiny_scene(P, W, H, f, seed=seed)
    return sc, cam.to(dev), W, H


def _fused(R, sc, cam, feats, dev, frozen_key, bg=(0.1, 0.2, 0.3)):
    rs = helpers.settings_for(cam, bg, 3, dev)
    m2 = torch.zeros(sc.means3D.shape[0], 3, device=dev)
    return R.rasterize_fused(sc.means3D.to(dev), m2, sc.opacities.to(dev), sc.shs.to(dev), feats, rs, scales=sc.scales.to(dev),
                             rotations=sc.rotations.to(dev), detach_extra_from_geometry=False, frozen_key=frozen_key)


def test_reblend_is_bit_identical_to_a_full_pass(kept, gpu_device):
    R, dev = kept, gpu_device
    sc, cam, W, H = _scene(dev)
    g = torch.Generator().manual_seed(1)
    f0 = sc.ins_feat.to(dev)
    f1 = torch.rand(f0.shape, generator=g).to(dev)
    key = ("cam0", ("v", 0), None)
    before = R.PASS_STATS["reblend"]
    miss = _fused(R, sc, cam, f0, dev, key)
    assert R.KEPT_PASSES.stats["admitted"] == 1 and R.PASS_STATS["reblend"] == before
    hit = _fused(R, sc, cam, f0, dev, key)
    assert R.PASS_STATS["reblend"] == before + 1 and R.KEPT_PASSES.stats["hits"] == 1
    for a, b, what in zip(miss, hit, ("color", "radii", "depth", "alpha")):
        assert torch.equal(a, b), what
    # new features: the re-blend against a full pass that never saw the cache; another background as well (not part of the key)
    full = _fused(R, sc, cam, f1, dev, None, bg=(0.7, 0.0, 0.4))
    again = _fused(R, sc, cam, f1, dev, key, bg=(0.7, 0.0, 0.4))
    assert R.PASS_STATS["reblend"] == before + 2
    for a, b, what in zip(full, again, ("color", "radii", "depth", "alpha")):
        assert torch.equal(a, b), what
    # and the kept entry is exact-size: the record array holds what the tiles packed, nothing else
    e = next(iter(R.KEPT_PASSES.slots.values()))
    assert e.sorted_rec.numel() == int(R._lib.lib().ogs_raster_sorted_bytes(e.D, 9)) and e.nbytes == R.KEPT_PASSES.nbytes


def test_reblend_feature_gradient_equals_the_full_pass(kept, gpu_device):
    R, dev = kept, gpu_device
    sc, cam, W, H = _scene(dev, seed=8)
    g = torch.Generator().manual_seed(2)
    gC = torch.randn(9, H, W, generator=g).to(dev)
    gA = torch.randn(1, H, W, generator=g).to(dev)
    key = ("cam0", ("v", 0), None)

    def grad_of(frozen_key, feats):
        leaf = feats.clone().requires_grad_(True)
        color, radii, depth, alpha = _fused(R, sc, cam, leaf, dev, frozen_key)
        ((color * gC).sum() + (alpha * gA).sum()).backward()          # the alpha term reaches no feature
        return leaf.grad

    f0 = sc.ins_feat.to(dev)
    f1 = torch.rand(f0.shape, generator=g).to(dev)
    grad_of(key, f0)                                                   # miss: admitted
    got = grad_of(key, f1)                                             # hit
    assert R.KEPT_PASSES.stats["hits"] == 1
    want = grad_of(None, f1)
    assert float(want.abs().max()) > 0
    # same kernels over the same blend state; the fp64 record makes the sums order-insensitive to ~1e-16
    torch.testing.assert_close(got, want, rtol=1e-6, atol=1e-9)


def test_changed_key_drops_the_entry_and_budget_stops_admission(kept, gpu_device):
    R, dev = kept, gpu_device
    sc, cam, W, H = _scene(dev, P=3000)
    f0 = sc.ins_feat.to(dev)
    _fused(R, sc, cam, f0, dev, ("cam0", ("v", 0), None))
    assert len(R.KEPT_PASSES.slots) == 1
    before = R.PASS_STATS["reblend"]
    _fused(R, sc, cam, f0, dev, ("cam0", ("v", 1), None))              # the parameters' version moved on: full pass, new entry
    st = R.KEPT_PASSES.stats
    assert st["stale"] == 1 and st["admitted"] == 2 and R.PASS_STATS["reblend"] == before and len(R.KEPT_PASSES.slots) == 1
    # an input that still requires grad never goes through the cache
    m3 = sc.means3D.to(dev).requires_grad_(True)
    rs = helpers.settings_for(cam, (0.0, 0.0, 0.0), 3, dev)
    R.rasterize_fused(m3, torch.zeros_like(m3), sc.opacities.to(dev), sc.shs.to(dev), f0, rs, scales=sc.scales.to(dev),
                      rotations=sc.rotations.to(dev), frozen_key=("cam0", ("v", 1), None))
    assert R.PASS_STATS["reblend"] == before and st["hits"] == 0
    # budget: nothing is admitted beyond it, nothing is evicted for it
    R.KEPT_PASSES = R.KeptPasses(budget_bytes=1 << 16)
    _fused(R, sc, cam, f0, dev, ("cam1", ("v", 0), None))
    assert R.KEPT_PASSES.stats["rejected_budget"] == 1 and not R.KEPT_PASSES.slots
    # ... except entries of another GENERATION of the model (they can never hit again): those go first
    sc2, cam2, _, _ = _scene(dev, P=3000, seed=9)
    one = R.KeptPasses(budget_bytes=4 << 30)
    R.KEPT_PASSES = one
    _fused(R, sc, cam, f0, dev, ("cam0", ("v", 0), None, "model-A"))
    one.budget_bytes = one.nbytes + 1024                               # room for nothing more
    _fused(R, sc2, cam2, sc2.ins_feat.to(dev), dev, ("cam1", ("v", 0), None, "model-A"))
    assert one.stats["rejected_budget"] == 1 and list(one.slots) == ["cam0"]
    one.budget_bytes = int(2.4 * one.nbytes)                           # room for one more of about the same size, not for two
    _fused(R, sc2, cam2, sc2.ins_feat.to(dev), dev, ("cam1", ("v", 0), None, "model-A"))
    assert sorted(one.slots) == ["cam0", "cam1"] and one.stats["admitted"] == 2
    _fused(R, sc, cam, f0, dev, ("cam2", ("w", 0), None, "model-B"))
    assert one.stats["dropped_old_generation"] == 2 and list(one.slots) == ["cam2"]
    R.KEPT_PASSES = R.KeptPasses(budget_bytes=0)                       # off
    _fused(R, sc, cam, f0, dev, ("cam1", ("v", 0), None))
    assert R.KEPT_PASSES.stats["misses"] == 0


def test_render_keeps_the_stage1_pass_and_invalidates_on_a_parameter_update(kept, gpu_device):
    """render() over a model shaped like the reference's GaussianModel with everything but `_ins_feat` detached
    (train.py:431-436), rescale=False (stage 1, train.py:346-350): second call = re-blend, same dict entries bit for bit as an
    uncached render; an in-place update of a parameter (what an optimizer step does) moves its version counter -> full pass."""
    from opengaussian_amd.renderer import render
    R, dev = kept, gpu_device
    sc, cam, W, H = _scene(dev, P=5000, seed=11)
    pipe = types.SimpleNamespace(debug=False, compute_cov3D_python=False, convert_SHs_python=False)
    bg = torch.tensor([0.2, 0.1, 0.3], device=dev)
    pc = ReferenceShapedGaussians(sc, dev)
    g = torch.Generator().manual_seed(4)
    gF = torch.randn(6, H, W, generator=g).to(dev)

    def step(model):
        out = render(cam, model, pipe, bg, iteration=40000, rescale=False)
        model._ins_feat.grad = None
        (out["ins_feat"] * gF).sum().backward()
        return out, model._ins_feat.grad.clone()

    before = R.PASS_STATS["reblend"]
    out0, g0 = step(pc)
    assert R.PASS_STATS["reblend"] == before and R.KEPT_PASSES.stats["admitted"] == 1
    # the training loop re-detaches every iteration (new tensor objects, same storage and version): still the same key
    for name in ("_xyz", "_scaling", "_rotation", "_opacity", "_features_dc", "_features_rest"):
        setattr(pc, name, getattr(pc, name).detach())
    with torch.no_grad():
        pc._ins_feat.add_(0.05 * torch.randn(pc._ins_feat.shape, generator=g).to(dev))      # the one thing that trains
    out1, g1 = step(pc)
    assert R.PASS_STATS["reblend"] == before + 1
    # reference: the same model through a cache that is switched off
    R.KEPT_PASSES, on = R.KeptPasses(budget_bytes=0), R.KEPT_PASSES
    ref1, gref1 = step(pc)
    R.KEPT_PASSES = on
    for k in ("render", "alpha", "depth", "silhouette", "ins_feat", "radii", "visibility_filter"):
        assert torch.equal(out1[k], ref1[k]), k
    torch.testing.assert_close(g1, gref1, rtol=1e-6, atol=1e-9)
    assert not torch.equal(out1["ins_feat"], out0["ins_feat"]) and torch.equal(out1["render"], out0["render"])
    # an optimizer-style in-place step of a geometry parameter: version counter moves, the kept pass must not be used
    with torch.no_grad():
        pc._xyz.add_(0.01)
    out2, _ = step(pc)
    assert R.PASS_STATS["reblend"] == before + 1 and R.KEPT_PASSES.stats["stale"] == 1
    assert not torch.equal(out2["render"], out1["render"])
    # rescale draws (stage 2, train.py:346-350) never use the cache: the footprints change with the draw
    torch.manual_seed(1)
    plain = 0
    for _ in range(6):                             # render()'s own CPU draws (gaussian_renderer/__init__.py:121-124)
        if float(torch.rand(1)) > 0.5:
            torch.rand(1)
        else:
            plain += 1
    assert 0 < plain < 6
    torch.manual_seed(1)
    n = R.PASS_STATS["reblend"]
    for _ in range(6):
        render(cam, pc, pipe, bg, iteration=60000, rescale=True)
    assert R.PASS_STATS["reblend"] - n == plain    # exactly the calls whose draw came out as "no rescale"


def test_render_keeps_the_outputs_of_a_pass_in_which_nothing_trains(kept, gpu_device):
    """Stage 2.2 (train.py:339-341: render_feat=False, render_cluster=True) and the rescaled calls of stage 2.1 render RGB through
    the unfused pass from frozen parameters: for a camera, a model state and a background tensor the same outputs every time.
    The second call returns them from rasterizer.KEPT_IMAGES -- bit for bit, as fresh tensors -- without launching a pass; another
    background tensor, a parameter update or a model that still trains -> a real pass."""
    from opengaussian_amd.renderer import render
    R, dev = kept, gpu_device
    saved, R.KEPT_IMAGES = R.KEPT_IMAGES, R.KeptImages(budget_bytes=1 << 30)
    try:
        sc, cam, W, H = _scene(dev, P=5000, seed=13)
        pipe = types.SimpleNamespace(debug=False, compute_cov3D_python=False, convert_SHs_python=False)
        bg = torch.tensor([0.2, 0.1, 0.3], device=dev)
        pc = ReferenceShapedGaussians(sc, dev)
        passes = lambda: sum(R.PASS_STATS[k] for k in ("blocking", "deferred", "tiny", "reblend"))
        call = lambda b=bg: render(cam, pc, pipe, b, iteration=60000, rescale=False, render_feat_map=False)
        n0 = passes()
        a = call()
        assert passes() == n0 + 1 and R.KEPT_IMAGES.stats["admitted"] == 1 and a["ins_feat"] is None
        b = call()
        assert passes() == n0 + 1 and R.KEPT_IMAGES.stats["hits"] == 1               # no pass was launched
        for k in ("render", "alpha", "depth", "radii", "visibility_filter"):
            assert torch.equal(a[k], b[k]), k
            assert a[k].data_ptr() != b[k].data_ptr()                               # clones: the caller may write into them
        b["render"].zero_()
        c = call()
        assert torch.equal(c["render"], a["render"]) and passes() == n0 + 1
        # another background TENSOR (random_background draws a new one per iteration): a real pass, and the right image
        bg2 = torch.tensor([0.9, 0.8, 0.0], device=dev)
        d = call(bg2)
        assert passes() == n0 + 2 and not torch.equal(d["render"], a["render"])
        R.KEPT_IMAGES, on = R.KeptImages(budget_bytes=0), R.KEPT_IMAGES
        ref = call(bg2)
        R.KEPT_IMAGES = on
        assert torch.equal(ref["render"], d["render"]) and torch.equal(call(bg2)["render"], ref["render"])
        # a parameter update invalidates
        with torch.no_grad():
            pc._opacity.add_(0.3)
        n1 = passes()
        e = call()
        assert passes() == n1 + 1 and not torch.equal(e["render"], a["render"])
        # a model that still trains never goes through the cache
        pc._xyz = pc._xyz.clone().requires_grad_(True)
        n2, hits = passes(), R.KEPT_IMAGES.stats["hits"]
        call(); call()
        assert passes() == n2 + 2 and R.KEPT_IMAGES.stats["hits"] == hits
    finally:
        R.KEPT_IMAGES = saved
