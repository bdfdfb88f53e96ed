"""dev/bench helper: stage-2.2/3 style fine-cluster loop of render() -- grouped passes vs one rasterizer call per
boolean-indexed subset (the reference's structure).  C4-like sizes: P = 2 M, 648x484, 64 x 10 leaves."""
import sys, time, types
import torch
sys.path.insert(0, ".")
from opengaussian_amd import renderer as R
from opengaussian_amd.synthetic import make_scene, make_camera
from tests.test_11_render_gpu import FakeGaussians

dev = torch.device("cuda:0")
P, W, H, f = 2_000_000, 648, 484, 500.0
sc = make_scene(P, W, H, f, f, seed=0)
cam = make_camera(W, H, f, f).to(dev)
pipe = types.SimpleNamespace(debug=False, compute_cov3D_python=False, convert_SHs_python=False)
g = torch.Generator().manual_seed(0)
leaf = torch.randint(0, 640, (P,), generator=g).to(dev)
pc = FakeGaussians(sc, dev)
bg = torch.zeros(3, device=dev)
for mode, kw in (("all 640 leaves (stage 3 / pseudo labels)", dict(leaf_cluster_idx=leaf)),
                 ("10 leaves of one root (stage 2.2 step)", dict(leaf_cluster_idx=leaf, selected_root_id=5))):
    for batch in (False, True):
        R.BATCH_SUBSETS = batch
        with torch.no_grad():
            for it in range(3):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                out = R.render(cam, pc, pipe, bg, iteration=1, rescale=False, render_feat_map=False, render_color=True, **kw)
                torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"{mode:45s} batched={batch!s:5s} {dt*1e3:9.2f} ms  ({len(out['leaf_clusters_imgs'])} images)", flush=True)
        del out
