"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/*.h declares.
No compute calls (there is no GPU here); size queries are pure host arithmetic."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DECL = re.compile(r"^\s*(?:int|size_t|const\s+char\s*\*)\s+(ogs_[a-z0-9_]+)\s*\(", re.M)


@pytest.fixture(scope="module")
def lib():
    from opengaussian_amd import _lib, build
    if not os.path.exists(_lib.LIB_PATH):
        build.build(verbose=False)
    return _lib.lib()


def _declared():
    names = set()
    inc = os.path.join(ROOT, "include")
    for fn in sorted(os.listdir(inc)):
        if fn.endswith(".h"):
            names |= set(DECL.findall(open(os.path.join(inc, fn)).read()))
    return names


def test_every_declared_symbol_is_exported_and_bound(lib):
    from opengaussian_amd import _lib
    declared = _declared()
    assert len(declared) >= 15
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), f"{name} declared in include/ but not exported by libogs_hip.so"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature in opengaussian_amd/_lib.py"
    assert set(_lib.SIGNATURES) <= declared, f"bound but undeclared: {set(_lib.SIGNATURES) - declared}"


def test_scratch_size_queries(lib):
    assert lib.ogs_version() >= 100
    g3, g6 = lib.ogs_raster_geom_bytes(1000, 3), lib.ogs_raster_geom_bytes(1000, 6)
    assert g3 >= 1000 * 52 and g6 > g3
    assert lib.ogs_raster_geom_bytes(0, 3) > 0
    assert lib.ogs_raster_image_bytes(1920, 1080) >= 8160 * 8 + 1920 * 1080 * 4
    assert lib.ogs_raster_binning_tmp_bytes(10_000_000, 1920, 1080) >= 3 * 4 * 10_000_000
    assert lib.ogs_raster_backward_tmp_bytes(1000) >= 64 * 1000
    assert lib.ogs_kmeans_tmp_bytes(2_000_000, 9, 64) > 0


def test_struct_layout_matches_header():
    """ctypes mirrors of OgsRasterFwdArgs / OgsRasterBwdArgs must list the header's fields in order."""
    from opengaussian_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "ogs_raster.h")).read()
    for cls, cname in ((_lib.OgsRasterFwdArgs, "OgsRasterFwdArgs"), (_lib.OgsRasterBwdArgs, "OgsRasterBwdArgs")):
        body = hdr.split(f"typedef struct {cname} {{")[1].split(f"}} {cname};")[0]
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        fields = []
        for stmt in body.split(";"):
            stmt = stmt.strip()
            if not stmt:
                continue
            names = stmt.split(",")
            first = names[0].split()[-1].lstrip("*")
            fields.append(first)
            fields += [n.strip().lstrip("*") for n in names[1:]]
        assert fields == [f[0] for f in cls._fields_], (cname, fields)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from opengaussian_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.OgsError):
        _lib.lib()


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "opengaussian_amd")):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, fn)).read()
                assert "import oracle" not in src and "from oracle" not in src, f"{fn} reaches into oracle/"


def test_cpu_tensors_are_refused():
    import torch
    from opengaussian_amd.rasterizer import GaussianRasterizer
    from tests import helpers
    sc, cam = helpers.tiny_scene(16, 32, 32, 30.0)
    rast = GaussianRasterizer(helpers.settings_for(cam, (0, 0, 0), 3, "cpu"))
    with pytest.raises(RuntimeError, match="no CPU path"):
        rast(means3D=sc.means3D, means2D=torch.zeros(16, 3), opacities=sc.opacities, shs=sc.shs, scales=sc.scales,
             rotations=sc.rotations)
