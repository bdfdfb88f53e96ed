"""Dev helper: dump the sorted (tile, depth) keys + point list of one forward pass (config 9 of
tests/test_10_raster_gpu.py::test_random_configurations_backward_parity) -- run once per OGS_RADIX mode, compare offline."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests import helpers
rng = np.random.default_rng(77)
for it in range(12):
    W, H = int(rng.integers(40, 170)), int(rng.integers(30, 120))
    P = int(rng.choice([40, 300, 900, 1024, 1025, 2500]))
    f = float(rng.uniform(0.5, 1.4) * max(W, H))
    lsm = float(rng.uniform(-4.0, -1.2))
    mode = it % 3
    sc, cam = helpers.tiny_scene(P, W, H, f, seed=4000 + it, log_scale_mean=lsm, with_ties=bool(it % 4 == 0))
    if it % 4 == 1:
        sc.opacities[:] = torch.rand_like(sc.opacities) ** 3
    if mode != 1:
        _ = rng.uniform(0, 1, 3)
    if it != 9:
        continue
    inp = helpers.oracle_inputs(sc, cam, use_sh=True, use_cov=False)
    outs = []
    for rep in range(3):
        (color, radii, depth, alpha), leaves = helpers.hip_forward(inp, cam, (0.1, 0.2, 0.3), 3, torch.device("cuda:0"), requires_grad=True)
        keys, ranges, nc, pl = helpers.hip_export_binning(color)
        outs.append((keys.copy(), pl.copy(), color.detach().cpu().numpy()))
    same = all(np.array_equal(outs[0][0], o[0]) and np.array_equal(outs[0][1], o[1]) for o in outs)
    print("repeatable:", same, "D", len(outs[0][0]))
    np.savez(sys.argv[1], keys=outs[0][0], pl=outs[0][1], color=outs[0][2])
    from oracle import raster_oracle as ro
    ref = ro.render_forward(W=W, H=H, tanfovx=W / (2 * f), tanfovy=H / (2 * f), bg=np.array((0.1, 0.2, 0.3), np.float32), sh_degree=3, **inp)
    d = np.abs(outs[0][2] - ref["color"]).max(0)
    print("vs oracle: max", d.max(), "pixels > 1e-4:", int((d > 1e-4).sum()), np.argwhere(d > 1e-4)[:10].tolist())
    print("keys vs oracle", np.array_equal(outs[0][0], ref["binning"].keys_sorted), "pl", np.array_equal(outs[0][1], ref["binning"].point_list))
