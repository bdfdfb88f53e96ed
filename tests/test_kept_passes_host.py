"""CPU: host logic of the frozen-geometry cache (rasterizer.KeptPasses, renderer._frozen_geometry_key) -- slot / key
bookkeeping and the conditions under which render() refuses to vouch for a pass.  The kept state itself is GPU-only
(tests/test_14_kept_pass_gpu.py)."""
import types

import torch

from opengaussian_amd import rasterizer as R
from opengaussian_amd.renderer import _frozen_geometry_key


def _entry(key, nbytes, generation=None):
    e = R._KeptPass()
    e.key, e.nbytes, e.hits, e.generation = key, nbytes, 0, generation
    return e


def test_lookup_hit_stale_and_drop_keep_the_byte_count():
    kp = R.KeptPasses(budget_bytes=1000)
    assert kp.lookup("cam0", "k0") is None and kp.stats["misses"] == 1
    kp.slots["cam0"] = _entry("k0", 300); kp.nbytes += 300
    kp.slots["cam1"] = _entry("k0", 200); kp.nbytes += 200
    assert kp.lookup("cam0", "k0") is kp.slots["cam0"] and kp.stats["hits"] == 1 and kp.slots["cam0"].hits == 1
    # a slot whose key moved on is dropped on lookup, the other slot stays
    assert kp.lookup("cam0", "k1") is None
    assert kp.stats["stale"] == 1 and "cam0" not in kp.slots and kp.nbytes == 200
    kp.drop("nobody")
    assert kp.nbytes == 200
    kp.clear()
    assert kp.nbytes == 0 and not kp.slots


def test_frozen_key_is_refused_unless_the_model_is_frozen_and_reference_shaped():
    cam = types.SimpleNamespace(world_view_transform=torch.eye(4), full_proj_transform=torch.eye(4), camera_center=torch.zeros(3),
                                image_height=8, image_width=8, FoVx=1.0, FoVy=1.0)
    pipe = types.SimpleNamespace(debug=False, compute_cov3D_python=False, convert_SHs_python=False)
    pc = types.SimpleNamespace(active_sh_degree=3)
    for name in ("_xyz", "_scaling", "_rotation", "_opacity", "_features_dc", "_features_rest"):
        setattr(pc, name, torch.zeros(4, 3))
    # CPU tensors: there is no kept state without the GPU
    assert _frozen_geometry_key(cam, pc, pipe, pc._xyz, 1.0) is None
    # missing attribute (a model that is not shaped like scene/gaussian_model.py:GaussianModel)
    del pc._features_rest
    assert _frozen_geometry_key(cam, pc, pipe, pc._xyz, 1.0) is None
    pc._features_rest = torch.zeros(4, 3)
    pc._opacity = torch.zeros(4, 1, requires_grad=True)        # still training: stage 0
    assert _frozen_geometry_key(cam, pc, pipe, pc._xyz, 1.0) is None
    assert _frozen_geometry_key(cam, pc, types.SimpleNamespace(debug=True), pc._xyz, 1.0) is None


def test_kept_images_hand_out_clones_and_follow_the_same_slot_rules():
    ki = R.KeptImages(budget_bytes=1 << 20)
    outs = (torch.arange(12.0).reshape(3, 2, 2), torch.tensor([3, 0, 5], dtype=torch.int32), torch.ones(1, 2, 2), torch.zeros(1, 2, 2))
    assert ki.lookup("cam0", "k0") is None and ki.stats["misses"] == 1
    assert ki.admit("cam0", "k0", holds=None, outputs=outs, generation="A")
    outs[0].zero_()                                            # the caller keeps working on what it got: the kept copy must not move
    hit = ki.lookup("cam0", "k0")
    assert hit is not None and torch.equal(hit[0], torch.arange(12.0).reshape(3, 2, 2)) and hit[1].dtype == torch.int32
    hit[0].fill_(7.0)                                          # ... and neither on what a hit handed out
    assert torch.equal(ki.lookup("cam0", "k0")[0], torch.arange(12.0).reshape(3, 2, 2)) and ki.stats["hits"] == 2
    nbytes = ki.nbytes
    assert nbytes == 12 * 4 + 3 * 4 + 4 * 4 + 4 * 4
    assert ki.lookup("cam0", "k1") is None and ki.stats["stale"] == 1 and ki.nbytes == 0       # another key: dropped
    # budget: entries of an older generation go first, then admission stops
    ki = R.KeptImages(budget_bytes=2 * nbytes + 8)
    fresh = lambda: tuple(t.clone() for t in (torch.arange(12.0).reshape(3, 2, 2), torch.tensor([3, 0, 5], dtype=torch.int32),
                                              torch.ones(1, 2, 2), torch.zeros(1, 2, 2)))
    assert ki.admit("cam0", "k", None, fresh(), "A") and ki.admit("cam1", "k", None, fresh(), "A")
    assert not ki.admit("cam2", "k", None, fresh(), "A") and ki.stats["rejected_budget"] == 1 and len(ki.slots) == 2
    assert ki.admit("cam2", "k", None, fresh(), "B") and list(ki.slots) == ["cam2"] and ki.nbytes == nbytes
    ki.clear()
    assert not ki.slots and ki.nbytes == 0
