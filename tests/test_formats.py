"""CPU: on-disk formats (opengaussian_amd/formats.py).  The reference writes them through plyfile / bitarray, which
are not installed, so these tests pin the byte layout against hand-computed known answers and round trips --
PARITY UNPINNED against files written by the reference itself."""
import types

import numpy as np
import torch

from opengaussian_amd import formats as F


def _model(P, seed=0):
    g = torch.Generator().manual_seed(seed)
    return dict(xyz=torch.randn(P, 3, generator=g), features_dc=torch.randn(P, 1, 3, generator=g),
                features_rest=torch.randn(P, 15, 3, generator=g), opacity=torch.randn(P, 1, generator=g) * 3,
                scaling=torch.randn(P, 3, generator=g), rotation=torch.randn(P, 4, generator=g),
                ins_feat=torch.rand(P, 6, generator=g) * 2 - 1)


def test_ply_header_layout_and_round_trip(tmp_path):
    m = _model(37)
    path = tmp_path / "point_cloud" / "iteration_1" / "point_cloud.ply"
    F.save_ply(str(path), **m)
    raw = path.read_bytes()
    head, body = raw.split(b"end_header\n", 1)
    lines = head.decode("ascii").split("\n")
    assert lines[:3] == ["ply", "format binary_little_endian 1.0", "element vertex 37"]
    names = [l.split()[2] for l in lines if l.startswith("property float")]
    assert names == F.attribute_names() and len(names) == 6 + 6 + 3 + 45 + 1 + 3 + 4        # scene/gaussian_model.py:249-262
    assert names[6:12] == ["ins_feat_r", "ins_feat_g", "ins_feat_b", "ins_feat_r2", "ins_feat_g2", "ins_feat_b2"]
    assert [l for l in lines if l.startswith("property uchar")] == ["property uchar red", "property uchar green", "property uchar blue"]
    assert len(body) == 37 * (68 * 4 + 3)
    # first vertex: x y z, zero normals, ins_feat, then f_dc in channel-major order
    first = np.frombuffer(body[:68 * 4], dtype="<f4")
    np.testing.assert_array_equal(first[:3], m["xyz"][0].numpy())
    np.testing.assert_array_equal(first[3:6], 0)
    np.testing.assert_array_equal(first[6:12], m["ins_feat"][0].numpy())
    np.testing.assert_array_equal(first[12:15], m["features_dc"][0, 0].numpy())
    np.testing.assert_array_equal(first[15:15 + 45], m["features_rest"][0].t().reshape(-1).numpy())   # [3,15] flattened
    back = F.load_ply(str(path))
    for k in m:
        assert back[k].shape == m[k].shape and torch.equal(back[k], m[k]), k
    v = F.read_ply_vertices(str(path))
    want = np.clip((m["ins_feat"][:, 0].numpy() + 1) / 2 * 255, 0, 255)
    want[torch.sigmoid(m["opacity"][:, 0]).numpy() < 0.1] = 128
    np.testing.assert_array_equal(v["red"], want.astype(np.uint8))


def test_index_bit_packing_known_answer():
    # 3-bit fields 5,1,7,2 -> 101 001 111 010 -> 1010 0111 | 1010 0000 (zero padded) = 0xA7 0xA0
    assert F.pack_indices(np.array([5, 1, 7, 2]), 3) == bytes([0xA7, 0xA0])
    np.testing.assert_array_equal(F.unpack_indices(bytes([0xA7, 0xA0]), 3, 12), [5, 1, 7, 2])
    ids = np.random.default_rng(0).integers(0, 641, 10007)
    np.testing.assert_array_equal(F.unpack_indices(F.pack_indices(ids, 14), 14, 14 * 10007), ids)


def test_codebook_round_trip_root_and_leaf(tmp_path):
    g = torch.Generator().manual_seed(1)
    N = 5003
    km = types.SimpleNamespace(cls_ids=torch.randint(0, 64, (N,), generator=g), centers=torch.randn(64, 9, generator=g),
                               leaf_cls_ids=torch.randint(0, 641, (N,), generator=g), leaf_centers=torch.randn(641, 6, generator=g))
    for mode, ids, cent in (("root", km.cls_ids, km.centers), ("leaf", km.leaf_cls_ids, km.leaf_centers)):
        F.save_kmeans([km], ["ins_feat"], str(tmp_path), mode=mode)
        d = tmp_path / f"{mode}_code_book"
        assert sorted(p.name for p in d.iterdir()) == ["kmeans_args.npy", "kmeans_centers.pth", "kmeans_inds.bin"]
        n_bits = int(np.ceil(np.log2(N)))                              # 13: sized by the point count (train.py:82)
        assert (d / "kmeans_inds.bin").stat().st_size == (N * n_bits + 7) // 8
        book, got = F.load_code_book(str(d))
        np.testing.assert_array_equal(got, ids.numpy())
        assert torch.equal(book["ins_feat"], cent)
