"""dev helper: run the stage-1 step (fused 9-channel forward + features-only backward) a few times, for rocprofv3."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from opengaussian_amd.rasterizer import GaussianRasterizationSettings, rasterize_fused
from opengaussian_amd.synthetic import make_scene, orbit_camera
dev = torch.device("cuda:0")
P, W, H, f = 1_000_000, 1920, 1080, 1000.0
sc = make_scene(P, W, H, f, f, seed=0).to(dev)
cams = [orbit_camera(W, H, f, f, v, 8).to(dev) for v in range(8)]
sts = [GaussianRasterizationSettings(H, W, math.tan(c.FoVx / 2), math.tan(c.FoVy / 2), torch.zeros(3, device=dev), 1.0,
                                     c.world_view_transform, c.full_proj_transform, 3, c.camera_center, False, False) for c in cams]
feat = sc.ins_feat.clone().requires_grad_(True)
g = torch.randn(9, H, W, device=dev); g[:3] = 0
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    m2 = torch.zeros(P, 3, device=dev)
    color, radii, depth, alpha = rasterize_fused(sc.means3D, m2, sc.opacities, sc.shs, feat, sts[i % 8], scales=sc.scales, rotations=sc.rotations)
    feat.grad = None
    color.backward(g)
torch.cuda.synchronize()
print("ok")
