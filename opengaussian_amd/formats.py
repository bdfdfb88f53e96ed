"""On-disk formats of the reference, for checkpoint interop.  Host-side I/O only.

  * Gaussian PLY      scene/gaussian_model.py:249-298 (save_ply) / :304-356 (load_ply): binary little-endian PLY,
                      one `vertex` element, float32 properties
                      x y z nx ny nz ins_feat_r ins_feat_g ins_feat_b ins_feat_r2 ins_feat_g2 ins_feat_b2
                      f_dc_0..2 f_rest_0..44 opacity scale_0..2 rot_0..3  +  uchar red green blue (a preview colour:
                      (ins_feat[:, :3] + 1) / 2 * 255, grey 128 where sigmoid(opacity) < 0.1).
                      f_dc / f_rest are stored channel-major (the [P,K,3] tensors transposed to [P,3,K], flattened).
  * two-level codebook train.py:62-100 (save_kmeans) / utils/opengs_utlis.py:63-87 (load_code_book): the cluster ids
                      of all points as MSB-first n_bits-wide integers packed into bytes (`kmeans_inds.bin`, what
                      `bitarray.tofile` writes), `kmeans_args.npy` (a pickled dict: params, n_bits, total_len) and
                      `kmeans_centers.pth` (torch.save of {param: centres}).  n_bits = ceil(log2(number of POINTS))
                      -- the reference sizes the field by len(cls_ids), not by the number of clusters; kept.

The reference writes these through `plyfile` and `bitarray`, neither of which is installed here, so the byte
layout below is restated from their documented formats (PLY 1.0 binary_little_endian; bitarray's default
big-endian bit order) and is PARITY UNPINNED against files the reference itself wrote.
"""
from __future__ import annotations

import os

import numpy as np
import torch

INS_NAMES = ["ins_feat_r", "ins_feat_g", "ins_feat_b", "ins_feat_r2", "ins_feat_g2", "ins_feat_b2"]


def attribute_names(n_dc: int = 3, n_rest: int = 45, n_scale: int = 3, n_rot: int = 4):
    """construct_list_of_attributes (scene/gaussian_model.py:249-262)."""
    names = ["x", "y", "z", "nx", "ny", "nz"] + INS_NAMES
    names += [f"f_dc_{i}" for i in range(n_dc)] + [f"f_rest_{i}" for i in range(n_rest)]
    names += ["opacity"] + [f"scale_{i}" for i in range(n_scale)] + [f"rot_{i}" for i in range(n_rot)]
    return names


def save_ply(path, xyz, features_dc, features_rest, opacity, scaling, rotation, ins_feat):
    """Tensors as GaussianModel holds them: xyz [P,3], features_dc [P,1,3], features_rest [P,K-1,3], opacity [P,1]
    (pre-sigmoid), scaling [P,3] (log), rotation [P,4], ins_feat [P,6]."""
    to_np = lambda t: t.detach().cpu().numpy().astype(np.float32)
    xyz_n = to_np(xyz)
    P = xyz_n.shape[0]
    f_dc = to_np(features_dc.transpose(1, 2).flatten(start_dim=1).contiguous())
    f_rest = to_np(features_rest.transpose(1, 2).flatten(start_dim=1).contiguous())
    op, sc, rot, ins = to_np(opacity).reshape(P, 1), to_np(scaling), to_np(rotation), to_np(ins_feat)
    names = attribute_names(f_dc.shape[1], f_rest.shape[1], sc.shape[1], rot.shape[1])
    dtype = np.dtype([(n, "<f4") for n in names] + [("red", "u1"), ("green", "u1"), ("blue", "u1")])
    el = np.empty(P, dtype=dtype)
    cols = np.concatenate([xyz_n, np.zeros_like(xyz_n), ins, f_dc, f_rest, op, sc, rot], axis=1)
    for i, n in enumerate(names):
        el[n] = cols[:, i]
    vis = np.clip((ins[:, :3] + 1) / 2 * 255, 0, 255)
    vis[(1.0 / (1.0 + np.exp(-op[:, 0]))) < 0.1] = 128
    el["red"], el["green"], el["blue"] = vis[:, 0].astype(np.uint8), vis[:, 1].astype(np.uint8), vis[:, 2].astype(np.uint8)
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    header = ["ply", "format binary_little_endian 1.0", f"element vertex {P}"]
    header += [f"property float {n}" for n in names] + ["property uchar red", "property uchar green", "property uchar blue",
                                                        "end_header"]
    with open(path, "wb") as fh:
        fh.write(("\n".join(header) + "\n").encode("ascii"))
        fh.write(el.tobytes())


_PLY_TYPES = {"float": "<f4", "float32": "<f4", "double": "<f8", "float64": "<f8", "uchar": "u1", "uint8": "u1",
              "char": "i1", "int8": "i1", "short": "<i2", "int16": "<i2", "ushort": "<u2", "uint16": "<u2",
              "int": "<i4", "int32": "<i4", "uint": "<u4", "uint32": "<u4"}


def read_ply_vertices(path) -> np.ndarray:
    """Structured array of the `vertex` element of a binary little-endian PLY (scalar properties only)."""
    with open(path, "rb") as fh:
        if fh.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, count, props, in_vertex = None, None, [], False
        while True:
            line = fh.readline()
            if not line:
                raise ValueError(f"{path}: unterminated PLY header")
            tok = line.decode("ascii").split()
            if not tok or tok[0] == "comment":
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                if count is not None and not in_vertex:
                    pass
                in_vertex = tok[1] == "vertex"
                if in_vertex:
                    count = int(tok[2])
                elif count is None:
                    raise ValueError(f"{path}: element '{tok[1]}' precedes 'vertex' (unsupported)")
            elif tok[0] == "property" and in_vertex:
                if tok[1] == "list":
                    raise ValueError(f"{path}: list properties are not supported")
                props.append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        if fmt != "binary_little_endian":
            raise ValueError(f"{path}: format '{fmt}' not supported (binary_little_endian only)")
        return np.frombuffer(fh.read(count * np.dtype(props).itemsize), dtype=np.dtype(props), count=count)


def load_ply(path, max_sh_degree: int = 3, device="cpu"):
    """dict of float32 tensors shaped as GaussianModel.load_ply builds them (scene/gaussian_model.py:304-356):
    xyz [P,3], features_dc [P,1,3], features_rest [P,K-1,3], opacity [P,1], scaling [P,3], rotation [P,4], ins_feat [P,6]."""
    v = read_ply_vertices(path)
    col = lambda names: np.stack([np.asarray(v[n], dtype=np.float32) for n in names], axis=1)
    by_index = lambda prefix: sorted([n for n in v.dtype.names if n.startswith(prefix)], key=lambda s: int(s.split("_")[-1]))
    rest = by_index("f_rest_")
    if len(rest) != 3 * (max_sh_degree + 1) ** 2 - 3:
        raise ValueError(f"{path}: {len(rest)} f_rest properties, expected {3 * (max_sh_degree + 1) ** 2 - 3}")
    P = v.shape[0]
    t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device=device)
    return {
        "xyz": t(col(["x", "y", "z"])),
        "ins_feat": t(col(INS_NAMES)),
        "opacity": t(col(["opacity"])),
        "features_dc": t(col(["f_dc_0", "f_dc_1", "f_dc_2"]).reshape(P, 3, 1)).transpose(1, 2).contiguous(),
        "features_rest": t(col(rest).reshape(P, 3, (max_sh_degree + 1) ** 2 - 1)).transpose(1, 2).contiguous(),
        "scaling": t(col(by_index("scale_"))),
        "rotation": t(col(by_index("rot"))),
    }


# ---- two-level codebook ------------------------------------------------------------------------------------------
def pack_indices(ids: np.ndarray, n_bits: int) -> bytes:
    """MSB-first n_bits-wide fields, concatenated, zero-padded to a whole byte (dec2binary + bitarray.tofile)."""
    ids = np.asarray(ids, dtype=np.int64).reshape(-1)
    if n_bits < 1 or (ids.size and (ids.min() < 0 or ids.max() >= (1 << n_bits))):
        raise ValueError("ids do not fit n_bits")
    shifts = np.arange(n_bits - 1, -1, -1, dtype=np.int64)
    bits = ((ids[:, None] >> shifts[None, :]) & 1).astype(np.uint8).reshape(-1)
    return np.packbits(bits, bitorder="big").tobytes()


def unpack_indices(buf: bytes, n_bits: int, total_len: int) -> np.ndarray:
    bits = np.unpackbits(np.frombuffer(buf, dtype=np.uint8), bitorder="big")[:total_len].reshape(-1, n_bits).astype(np.int64)
    return (bits << np.arange(n_bits - 1, -1, -1, dtype=np.int64)[None, :]).sum(axis=1)


def save_kmeans(kmeans_list, quantized_params, out_dir, mode="root"):
    """train.py:62-100.  `kmeans_list[i]` exposes cls_ids / centers (root) or leaf_cls_ids / leaf_centers (leaf)."""
    out_dir = os.path.join(out_dir, "root_code_book" if mode == "root" else "leaf_code_book")
    os.makedirs(out_dir, exist_ok=True)
    blob, total, n_bits = b"", 0, 0
    fields = []
    for km in kmeans_list:
        ids = (km.cls_ids if mode == "root" else km.leaf_cls_ids).detach().cpu().numpy()
        n_bits = int(np.ceil(np.log2(len(ids))))              # sized by the number of points, as the reference does
        fields.append((ids, n_bits))
        total += len(ids) * n_bits
    # the reference extends ONE bitarray with every parameter's bits and pads only at the very end
    allbits = np.concatenate([((np.asarray(i, np.int64)[:, None] >> np.arange(b - 1, -1, -1)[None, :]) & 1).astype(np.uint8).reshape(-1)
                              for i, b in fields]) if fields else np.zeros(0, np.uint8)
    blob = np.packbits(allbits, bitorder="big").tobytes()
    with open(os.path.join(out_dir, "kmeans_inds.bin"), "wb") as fh:
        fh.write(blob)
    np.save(os.path.join(out_dir, "kmeans_args.npy"), {"params": quantized_params, "n_bits": n_bits, "total_len": total})
    centers = {p: (km.centers if mode == "root" else km.leaf_centers) for km, p in zip(kmeans_list, quantized_params)}
    torch.save(centers, os.path.join(out_dir, "kmeans_centers.pth"))


def load_code_book(base_path):
    """utils/opengs_utlis.py:63-87 -> (codebook dict, ins_feat ids [N]).  The args file is a pickled dict (what
    np.save writes for a dict); only load code books you wrote or trust."""
    args = np.load(os.path.join(base_path, "kmeans_args.npy"), allow_pickle=True).item()
    codebook = torch.load(os.path.join(base_path, "kmeans_centers.pth"))
    with open(os.path.join(base_path, "kmeans_inds.bin"), "rb") as fh:
        ids = unpack_indices(fh.read(), int(args["n_bits"]), int(args["total_len"]))
    ids = ids.reshape(len(args["params"]), -1)
    return codebook, {k: ids[i] for i, k in enumerate(args["params"])}["ins_feat"]
