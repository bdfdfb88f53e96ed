"""Synthetic scenes + cameras for the bench and the parity tests (SURVEY.md section 8(d) recipe).

There are no datasets in the container, so every measured workload is generated here with a
seeded CPU ``torch.Generator``.  Camera conventions follow the reference exactly, because they are
part of the rasterizer boundary contract:
  * ``projection_matrix`` restates utils/graphics_utils.py:54-74 (``getProjectionMatrix``);
  * ``Camera`` builds ``world_view_transform`` = W2C^T, ``full_proj_transform`` = W2C^T @ P^T and
    ``camera_center`` = inverse(W2C^T)[3,:3] as scene/cameras.py:71-78 does (row-vector convention).
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch


def projection_matrix(znear: float, zfar: float, fovx: float, fovy: float) -> torch.Tensor:
    """OpenGL-style perspective matrix with z_sign=+1 (utils/graphics_utils.py:54-74)."""
    tx, ty = math.tan(fovx / 2), math.tan(fovy / 2)
    top, right = ty * znear, tx * znear
    P = torch.zeros(4, 4)
    P[0, 0] = 2.0 * znear / (2 * right)
    P[1, 1] = 2.0 * znear / (2 * top)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


@dataclass
class Camera:
    """Minimal stand-in for scene/cameras.py:Camera -- only the attributes render() reads."""
    image_width: int
    image_height: int
    FoVx: float
    FoVy: float
    world_view_transform: torch.Tensor   # [4,4]  W2C^T
    full_proj_transform: torch.Tensor    # [4,4]  W2C^T @ P^T
    camera_center: torch.Tensor          # [3]
    bClusterOccur: object = None

    def to(self, device):
        return Camera(self.image_width, self.image_height, self.FoVx, self.FoVy,
                      self.world_view_transform.to(device), self.full_proj_transform.to(device),
                      self.camera_center.to(device), self.bClusterOccur)


def make_camera(W: int, H: int, fx: float, fy: float, R: torch.Tensor | None = None,
                t: torch.Tensor | None = None, znear: float = 0.01, zfar: float = 100.0) -> Camera:
    """W2C = [R | t] (world -> camera, +z forward).  Identity by default (camera at the origin)."""
    fovx = 2 * math.atan(W / (2 * fx))
    fovy = 2 * math.atan(H / (2 * fy))
    w2c = torch.eye(4)
    if R is not None:
        w2c[:3, :3] = R
    if t is not None:
        w2c[:3, 3] = t
    wvt = w2c.t().contiguous()
    proj_t = projection_matrix(znear, zfar, fovx, fovy).t().contiguous()
    full = wvt @ proj_t
    center = torch.linalg.inv(wvt)[3, :3].contiguous()
    return Camera(W, H, fovx, fovy, wvt, full.contiguous(), center)


def orbit_camera(W, H, fx, fy, view_index: int, num_views: int = 8, max_angle_deg: float = 6.0) -> Camera:
    """Camera ``view_index`` of a small fan of views about the scene centre (0,0,6): used for the
    one-view-per-GPU data-parallel bench so every rank renders a different, equally heavy view."""
    if num_views <= 1 or view_index == 0:
        return make_camera(W, H, fx, fy)
    ang = math.radians(max_angle_deg) * (2.0 * view_index / (num_views - 1) - 1.0)
    c, s = math.cos(ang), math.sin(ang)
    R = torch.tensor([[c, 0.0, s], [0.0, 1.0, 0.0], [-s, 0.0, c]])
    centre = torch.tensor([0.0, 0.0, 6.0])
    t = centre - R @ centre          # rotate about the scene centre
    return make_camera(W, H, fx, fy, R=R, t=t)


@dataclass
class Scene:
    means3D: torch.Tensor     # [P,3]
    scales: torch.Tensor      # [P,3]  (already exp-activated)
    rotations: torch.Tensor   # [P,4]  unit quaternions (r,x,y,z)
    opacities: torch.Tensor   # [P,1]  (already sigmoid-activated)
    shs: torch.Tensor         # [P,16,3]
    ins_feat: torch.Tensor    # [P,6]  normalise -> (.+1)/2, i.e. what render() feeds as colours

    def to(self, device):
        return Scene(*[getattr(self, f).to(device) for f in
                       ("means3D", "scales", "rotations", "opacities", "shs", "ins_feat")])


def make_scene(P: int, W: int, H: int, fx: float, fy: float, seed: int = 0,
               log_scale_mean: float = -4.5, log_scale_std: float = 0.7) -> Scene:
    """S1M-1080p is make_scene(1_000_000, 1920, 1080, 1000, 1000);
    C2 is make_scene(100_000, 800, 800, 700, 700)."""
    g = torch.Generator().manual_seed(seed)
    tanx, tany = W / (2 * fx), H / (2 * fy)
    z = torch.rand(P, generator=g) * 8.0 + 2.0
    u = torch.rand(P, generator=g) * 2.2 - 1.1
    v = torch.rand(P, generator=g) * 2.2 - 1.1
    n_cull = P // 100
    if n_cull:
        idx = torch.randperm(P, generator=g)[:n_cull]
        z[idx] = torch.rand(n_cull, generator=g) * 1.2 - 1.0     # U(-1, 0.2): behind the near plane
    means = torch.stack([u * z * tanx, v * z * tany, z], dim=1)
    scales = torch.exp(torch.randn(P, 3, generator=g) * log_scale_std + log_scale_mean)
    q = torch.randn(P, 4, generator=g)
    q = q / q.norm(dim=1, keepdim=True)
    opac = torch.sigmoid(torch.randn(P, 1, generator=g) * 2.0 - 1.0)
    f_dc = 0.5 * torch.randn(P, 1, 3, generator=g)
    f_rest = 0.1 * torch.randn(P, 15, 3, generator=g)
    shs = torch.cat([f_dc, f_rest], dim=1)
    feat = torch.rand(P, 6, generator=g)
    feat = (torch.nn.functional.normalize(feat, dim=1) + 1) / 2
    return Scene(means.contiguous(), scales.contiguous(), q.contiguous(), opac.contiguous(),
                 shs.contiguous(), feat.contiguous())
