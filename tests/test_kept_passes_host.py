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
