"""dev helper: explain image mismatches between the HIP path and the oracle as threshold flips."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from tests import helpers
from tests.test_10_raster_gpu import _adversarial_scene
from oracle import raster_oracle as ro
kind, P, W, H = "depth_ties", 2500, 128, 96
f = 70.0
sc, cam = _adversarial_scene(kind, P, W, H, f, seed=41)
inp = helpers.oracle_inputs(sc, cam, use_sh=True)
bg = (0.3, 0.2, 0.1)
ref = ro.render_forward(W=W, H=H, tanfovx=W / (2 * f), tanfovy=H / (2 * f), bg=np.array(bg, np.float32), sh_degree=3, **inp)
(color, radii, depth, alpha), _ = helpers.hip_forward(inp, cam, bg, 3, "cuda:0")
diff = np.abs(color.cpu().numpy() - ref["color"]).max(0)
ys, xs = np.nonzero(diff > 1e-4)
g, b = ref["geom"], ref["binning"]
for y, x in zip(ys, xs):
    t = (y // 16) * ((W + 15) // 16) + x // 16
    lo, hi = b.ranges[t]
    ids = b.point_list[lo:hi]
    T = np.float32(1.0)
    near = []
    for j, i in enumerate(ids):
        dx = np.float32(g.xy[i, 0] - np.float32(x)); dy = np.float32(g.xy[i, 1] - np.float32(y))
        con = g.conic[i]
        power = np.float32(-0.5) * (con[0] * dx * dx + con[2] * dy * dy) - con[1] * dx * dy
        if power > 0: continue
        a = min(np.float32(0.99), np.float32(g.opacity[i] * np.exp(power)))
        if abs(a * 255 - 1) < 1e-3: near.append(("alpha", j, float(a * 255)))
        if a < 1 / 255: continue
        Tn = T * (1 - a)
        if abs(Tn / 1e-4 - 1) < 1e-2: near.append(("T", j, float(Tn)))
        if Tn < 1e-4: break
        T = Tn
    print(f"pixel ({x},{y}) diff {diff[y,x]:.2e} hipalpha {float(alpha[0,y,x]):.6f} refalpha {ref['alpha'][0,y,x]:.6f} list {hi-lo} near-threshold: {near}")
