"""Dev helper: repeat one small forward pass many times with allocator churn (garbage-filled scratch) and count runs whose
image / binning differ from the first run.  usage: python scripts/race_hunt.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests import helpers
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda:0")
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
    first = None
    bad = 0
    g = torch.Generator(device="cuda").manual_seed(0)
    for rep in range(reps):
        junk = [torch.randint(0, 2**31 - 1, (int(n),), device=dev, dtype=torch.int32, generator=g) for n in rng.integers(1000, 400000, 6)]
        del junk
        (color, radii, depth, alpha), leaves = helpers.hip_forward(inp, cam, (0.1, 0.2, 0.3), 3, dev, requires_grad=True)
        keys, ranges, nc, pl = helpers.hip_export_binning(color)
        cur = (color.detach().cpu().numpy(), keys.copy(), pl.copy(), nc.copy())
        if first is None:
            first = cur
            continue
        dc = np.abs(cur[0] - first[0]).max()
        if dc > 0 or not np.array_equal(cur[1], first[1]) or not np.array_equal(cur[2], first[2]) or not np.array_equal(cur[3], first[3]):
            bad += 1
            if bad <= 5:
                print(f"rep {rep}: image diff {dc:.3g} keys_equal {np.array_equal(cur[1], first[1])} pl_equal {np.array_equal(cur[2], first[2])} "
                      f"n_contrib_equal {np.array_equal(cur[3], first[3])} pixels {np.argwhere(np.abs(cur[0] - first[0]).max(0) > 0)[:6].tolist()}")
    print(f"mode {os.environ.get('OGS_RADIX', 'sweep')}: {bad} of {reps - 1} repeats differ from the first run")
