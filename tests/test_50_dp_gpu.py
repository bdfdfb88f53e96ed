"""GPU tests of the data-parallel exchange: the rank-1 SH-gradient exchange
(dp.ShGradExchange + ogs_sh_grad_from_views) must give what summing the dense per-view dL/dsh gives, and a
two-rank run (gloo, both ranks on the one GPU of the test box) must reproduce a single-process sum over views."""
import math
import os
import socket

import numpy as np
import pytest
import torch

from tests import helpers

pytestmark = pytest.mark.gpu


def _view_settings(W, H, f, view, nviews, device):
    from opengaussian_amd.rasterizer import GaussianRasterizationSettings
    from opengaussian_amd.synthetic import orbit_camera
    cam = orbit_camera(W, H, f, f, view_index=view, num_views=nviews).to(device)
    return GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5),
        bg=torch.zeros(3, device=device), scale_modifier=1.0, viewmatrix=cam.world_view_transform,
        projmatrix=cam.full_proj_transform, sh_degree=3, campos=cam.camera_center, prefiltered=False, debug=False), cam


def _one_view(scene, settings, gC, gF, sink):
    from opengaussian_amd.rasterizer import rasterize_fused
    leaves = dict(means3D=scene.means3D, scales=scene.scales, rotations=scene.rotations, opacities=scene.opacities,
                  shs=scene.shs, ins_feat=scene.ins_feat)
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in leaves.items()}
    m2 = torch.zeros(scene.means3D.shape[0], 3, device=scene.means3D.device, requires_grad=True)
    color, radii, depth, alpha = rasterize_fused(leaves["means3D"], m2, leaves["opacities"], leaves["shs"],
                                                 leaves["ins_feat"], settings, scales=leaves["scales"],
                                                 rotations=leaves["rotations"], sh_rgb_sink=sink)
    torch.autograd.backward([color], [torch.cat([gC, gF])])
    return leaves, radii


@pytest.mark.parametrize("nviews", [1, 3])
def test_sh_gradient_from_rank1_factors_equals_dense_sum(gpu_device, nviews):
    from opengaussian_amd import dp
    from opengaussian_amd.synthetic import make_scene
    P, W, H, f = 6000, 160, 96, 120.0
    scene = make_scene(P, W, H, f, f, seed=11).to(gpu_device)
    g = torch.Generator().manual_seed(5)
    gC = torch.randn(3, H, W, generator=g).to(gpu_device)
    gF = torch.randn(6, H, W, generator=g).to(gpu_device)
    dense_sum = torch.zeros(P, 16, 3, device=gpu_device, dtype=torch.float64)
    factors, campos, others = [], [], []
    for v in range(nviews):
        st, cam = _view_settings(W, H, f, v, nviews, gpu_device)
        dense, _ = _one_view(scene, st, gC, gF, None)
        sink = []
        comp, _ = _one_view(scene, st, gC, gF, sink)
        assert comp["shs"].grad is None and len(sink) == 1 and sink[0].shape == (P, 3)
        for k in ("means3D", "scales", "rotations", "opacities", "ins_feat"):
            # atomics order differs run to run: equal up to fp32 summation noise
            torch.testing.assert_close(comp[k].grad, dense[k].grad, rtol=1e-4, atol=1e-5 * float(dense[k].grad.abs().max()))
        dense_sum += dense["shs"].grad.double()
        factors.append(sink[0]); campos.append(cam.camera_center)
        others.append(dense["shs"].grad)
    ex = dp.ShGradExchange(P, 16, gpu_device)
    assert ex.world == 1
    if nviews == 1:
        ex.gather_async(factors[0])
        out = ex.rebuild(scene.means3D, torch.stack(campos), 3)
        # one view: same basis polynomials, same multiply -> the dense gradient up to the atomics noise of the
        # two separate backward runs feeding it
        torch.testing.assert_close(out, others[0], rtol=1e-4, atol=1e-5 * float(others[0].abs().max()))
        # and exactly rank 1: dsh[:, 0, :] == C0 * factor
        torch.testing.assert_close(out[:, 0, :], 0.28209479177387814 * factors[0], rtol=1e-6, atol=0)
    else:
        ex.world = nviews                       # single process standing in for V ranks: fill the gathered buffer
        ex.gathered = torch.stack(factors)
        out = ex.rebuild(scene.means3D, torch.stack(campos), 3)
        scale = float(dense_sum.abs().max())
        assert float((out.double() - dense_sum).abs().max()) / scale < 2e-5
        assert float(out.abs().sum()) > 0


def test_sh_gradient_unused_coefficients_are_zero(gpu_device):
    from opengaussian_amd import _lib
    P = 1000
    g = torch.Generator().manual_seed(0)
    m3 = torch.randn(P, 3, generator=g).to(gpu_device)
    fac = torch.randn(2, P, 3, generator=g).to(gpu_device)
    cp = torch.tensor([[0.0, 0.0, -5.0], [4.0, 0.0, 3.0]], device=gpu_device)
    out = torch.full((P, 16, 3), float("nan"), device=gpu_device)
    _lib.check(_lib.lib().ogs_sh_grad_from_views(P, 2, 1, 16, m3.data_ptr(), cp.data_ptr(), fac.data_ptr(),
                                                 out.data_ptr(), torch.cuda.current_stream().cuda_stream), "sh")
    torch.cuda.synchronize()
    assert torch.isfinite(out).all() and float(out[:, 4:].abs().max()) == 0.0 and float(out[:, :4].abs().min()) >= 0
    d = m3[None] - cp[:, None]
    d = d / d.norm(dim=-1, keepdim=True)
    want1 = (-0.4886025119029199 * d[..., 1:2] * fac).sum(0)       # Y_1 = -C1 * y
    torch.testing.assert_close(out[:, 1, :], want1, rtol=1e-4, atol=1e-5)
    # invalid arguments are refused, not launched
    assert _lib.lib().ogs_sh_grad_from_views(P, 0, 1, 16, m3.data_ptr(), cp.data_ptr(), fac.data_ptr(), out.data_ptr(), 0) != 0
    assert _lib.lib().ogs_sh_grad_from_views(P, 2, 3, 9, m3.data_ptr(), cp.data_ptr(), fac.data_ptr(), out.data_ptr(), 0) != 0


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_main(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), OGS_DIST_BACKEND="gloo")
    import torch.distributed as dist
    from opengaussian_amd import dp
    from opengaussian_amd.synthetic import make_scene
    dp.init_from_env("cuda")
    dev = torch.device("cuda", 0)
    P, W, H, f = 4000, 128, 80, 100.0
    scene = make_scene(P, W, H, f, f, seed=3).to(dev)
    g = torch.Generator().manual_seed(9)
    gC = torch.randn(3, H, W, generator=g).to(dev)
    gF = torch.randn(6, H, W, generator=g).to(dev)
    st, cam = _view_settings(W, H, f, rank, world, dev)
    sink = []
    leaves, radii = _one_view(scene, st, gC, gF, sink)
    names = ["means3D", "scales", "rotations", "opacities", "ins_feat"]
    bucket = dp.GradBucket([leaves[n].shape for n in names], dev, average=False)
    bucket.pack([leaves[n].grad for n in names])
    bucket.allreduce_async()
    ex = dp.ShGradExchange(P, 16, dev)
    ex.gather_async(sink[0])
    campos_all = torch.stack([_view_settings(W, H, f, r, world, dev)[1].camera_center for r in range(world)])
    dsh = ex.rebuild(scene.means3D, campos_all, 3)
    red = bucket.wait()
    # single-process truth: every view rendered here, dense gradients summed in float64
    want = {n: torch.zeros_like(leaves[n], dtype=torch.float64) for n in names + ["shs"]}
    for r in range(world):
        lv, _ = _one_view(scene, _view_settings(W, H, f, r, world, dev)[0], gC, gF, None)
        for n in want:
            want[n] += lv[n].grad.double()
    err = {n: float((t.double() - want[n]).abs().max() / (want[n].abs().max() + 1e-30)) for n, t in zip(names, red)}
    err["shs"] = float((dsh.double() - want["shs"]).abs().max() / want["shs"].abs().max())
    q.put((rank, err))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_exchange_matches_single_process_sum(gpu_device):
    import torch.multiprocessing as mp
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, err in res:
        for k, e in err.items():
            assert e < 1e-4, (rank, k, e)


def _sharded_adam_rank(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), OGS_DIST_BACKEND="gloo")
    import torch.distributed as dist
    from opengaussian_amd import dp
    from opengaussian_amd.optim import FusedAdam
    dp.init_from_env("cuda")
    dev = torch.device("cuda", 0)
    P = 1237                                             # slices cut through the middle of parameters
    shapes = [("xyz", (P, 3)), ("f_rest", (P, 15, 3)), ("opacity", (P, 1)), ("rotation", (P, 4)), ("ins_feat", (P, 6))]
    lrs = {"xyz": 1.6e-4, "f_rest": 1.25e-4, "opacity": 0.05, "rotation": 1e-3, "ins_feat": 1e-3}
    g = torch.Generator().manual_seed(0)
    init = {n: torch.randn(*s, generator=g).to(dev) for n, s in shapes}
    opt = dp.ShardedAdam(shapes, lrs, dev)
    opt.load(init)
    # single-process truth: FusedAdam on the summed gradients of all ranks
    ref_params = {n: torch.nn.Parameter(init[n].clone()) for n, _ in shapes}
    ref = FusedAdam([{"params": [ref_params[n]], "lr": lrs[n]} for n, _ in shapes], lr=0.0, eps=1e-15)
    for it in range(5):
        total = {}
        for r in range(world):
            gr = torch.Generator().manual_seed(100 * it + r)
            for n, s in shapes:
                v = torch.randn(*s, generator=gr).to(dev)
                # it == 3: rank 1's view saw nothing of `opacity` -> its .grad is None (contributes zeros), the step happens
                absent = it == 3 and r == 1 and n == "opacity"
                if not absent:
                    total[n] = v if n not in total else total[n] + v
                if r == rank:
                    opt.params[n].grad = None if absent else v
        # torch.optim.Adam semantics per parameter: no gradient on ANY rank -> the parameter is skipped
        # (value, moments and its own step counter untouched); ins_feat starts late, rotation pauses at step 2
        skipped = set()
        if it < 2:
            skipped.add("ins_feat")
        if it == 2:
            skipped.add("rotation")
        for n in skipped:
            opt.params[n].grad = None
        opt.step()
        for n, _ in shapes:
            ref_params[n].grad = None if n in skipped else total[n].clone()
        ref.step()
    err = {n: float((opt.params[n].detach() - ref_params[n].detach()).abs().max()) for n, _ in shapes}
    assert opt.step_counts == {"xyz": 5, "f_rest": 5, "opacity": 5, "rotation": 4, "ins_feat": 3}, opt.step_counts
    q.put((rank, err))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_adam_two_ranks_match_replicated_adam(gpu_device):
    """reduce-scatter -> Adam on the owned slice -> all-gather == Adam on the all-reduced gradient (gloo rehearsal of
    the exchange on one GPU; the RCCL reduce_scatter_tensor branch itself is not exercised here)."""
    import torch.multiprocessing as mp
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sharded_adam_rank, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, err in res:
        for k, e in err.items():
            assert e == 0.0, (rank, k, e)       # two-rank sums are order independent: bit-identical parameters


def _nccl_one_rank(port, q):
    """Child process: backend "nccl" (= RCCL) with ONE rank on cuda:0, every exchange primitive of dp.py forced to
    call the backend (dp.FORCE_COLLECTIVES), results compared with the no-collective single-process values."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                      OGS_DP_FORCE_COLLECTIVES="1")
    os.environ.pop("OGS_DIST_BACKEND", None)
    import torch.distributed as dist
    from opengaussian_amd import dp
    from opengaussian_amd.optim import FusedAdam
    from opengaussian_amd.synthetic import make_scene
    assert dp.FORCE_COLLECTIVES
    rank, world, local = dp.init_from_env("cuda")
    assert dist.is_initialized() and dist.get_backend() == "nccl" and world == 1
    dev = torch.device("cuda", 0)
    out = {}
    P, W, H, f = 3000, 128, 80, 100.0
    scene = make_scene(P, W, H, f, f, seed=3).to(dev)
    g = torch.Generator().manual_seed(9)
    gC = torch.randn(3, H, W, generator=g).to(dev)
    gF = torch.randn(6, H, W, generator=g).to(dev)
    st, cam = _view_settings(W, H, f, 0, 1, dev)
    sink = []
    leaves, radii = _one_view(scene, st, gC, gF, sink)
    dense, _ = _one_view(scene, st, gC, gF, None)
    names = ["means3D", "scales", "rotations", "opacities", "ins_feat"]
    # GradBucket: async all-reduce on the side stream, wait on the compute stream
    bucket = dp.GradBucket([leaves[n].shape for n in names], dev, average=False)
    bucket.pack([leaves[n].grad for n in names])
    bucket.allreduce_async()
    assert bucket._work is not None                      # the RCCL call was issued
    red = bucket.wait()
    out["bucket"] = max(float((a - leaves[n].grad).abs().max()) for a, n in zip(red, names))
    # ShGradExchange: async all-gather + local rebuild
    ex = dp.ShGradExchange(P, 16, dev)
    assert ex.on
    ex.gather_async(sink[0])
    assert ex._work is not None
    dsh = ex.rebuild(scene.means3D, cam.camera_center[None], 3)
    out["sh"] = float((dsh - dense["shs"].grad).abs().max() / dense["shs"].grad.abs().max())
    # radii MAX (async) and the blocking statistics reduction
    rmax, work = dp.reduce_max_radii(radii, async_op=True)
    assert work is not None
    work.wait()
    out["rmax"] = int((rmax != radii).sum())
    m2g = torch.randn(P, 3, device=dev)
    norm, cnt, rm2 = dp.reduce_densification_stats(m2g, radii)
    out["stats"] = float((norm - torch.norm(m2g[:, :2], dim=-1) * (radii > 0)).abs().max())
    # ShardedAdam: reduce_scatter_tensor -> fused Adam -> all_gather_into_tensor (the nccl branch)
    shapes = [("xyz", (P, 3)), ("f_rest", (P, 15, 3)), ("ins_feat", (P, 6))]
    lrs = {"xyz": 1.6e-4, "f_rest": 1.25e-4, "ins_feat": 1e-3}
    init = {n: torch.randn(*s, generator=g).to(dev) for n, s in shapes}
    opt = dp.ShardedAdam(shapes, lrs, dev)
    assert opt.on
    opt.load(init)
    refp = {n: torch.nn.Parameter(init[n].clone()) for n, _ in shapes}
    ref = FusedAdam([{"params": [refp[n]], "lr": lrs[n]} for n, _ in shapes], lr=0.0, eps=1e-15)
    for it in range(3):
        for n, s in shapes:
            v = torch.randn(*s, generator=g).to(dev)
            opt.params[n].grad = v
            refp[n].grad = v.clone()
        opt.step(); ref.step()
    out["sharded_adam"] = max(float((opt.params[n].detach() - refp[n].detach()).abs().max()) for n, _ in shapes)
    torch.cuda.synchronize()
    q.put(out)
    dist.barrier()
    dist.destroy_process_group()


def test_rccl_one_rank_drives_every_exchange_primitive(gpu_device):
    """RCCL readiness on the one GPU of the test box: a fresh child initialises backend "nccl"
    with world_size 1 and runs GradBucket.allreduce_async / wait, ShGradExchange.gather_async / rebuild,
    reduce_max_radii(async_op=True), reduce_densification_stats and ShardedAdam.step()'s reduce_scatter_tensor /
    all_gather_into_tensor branch through RCCL.  N > 1 over xGMI stays unmeasured until the driver's SCALE run."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_nccl_one_rank, args=(_free_port(), q))
    p.start()
    res = q.get(timeout=300)
    p.join(timeout=120)
    assert p.exitcode == 0
    assert res["bucket"] == 0.0 and res["rmax"] == 0 and res["stats"] == 0.0 and res["sharded_adam"] == 0.0, res
    assert res["sh"] < 2e-6, res


def test_bench_under_torchrun_one_rank(gpu_device):
    """bench.py through the driver's launcher (`python -m torch.distributed.run --nproc-per-node 1`) with the
    collectives forced on: the N > 1 exchange path (double-buffered buckets, side streams, async handles, RCCL)
    runs end to end on a small workload and prints its one JSON line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, OGS_DP_FORCE_COLLECTIVES="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("OGS_DIST_BACKEND", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
           "--workload", "C2-100k-800", "--no-cpu-baseline", "--no-kmeans", "--no-extras"]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 1 and out["value"] > 0 and "RCCL" in out["config"]["parallelism"], out["config"]


def test_bench_self_launches_two_ranks(gpu_device):
    """`python3 bench.py --gpus 2 ...` invoked DIRECTLY (the driver's N = 1 command form, no launcher, WORLD_SIZE unset):
    the parent starts the two ranks itself before any GPU call, relays rank 0's JSON line and returns the children's
    return code.  On the one-GPU test box both ranks share cuda:0, so the collectives go over
    gloo (OGS_DIST_BACKEND); on the 8-GPU node the same command runs over RCCL."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, OGS_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "OGS_DP_FORCE_COLLECTIVES"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--workload", "C2-100k-800", "--no-cpu-baseline", "--no-kmeans"]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines                                     # ONE JSON line, nothing else on stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["value"] > 0
    d = out["dist"]
    assert d["backend"] == "gloo" and d["world_size"] == 2
    assert sorted(x["rank"] for x in d["ranks"]) == [0, 1]
    assert len({x["pid"] for x in d["ranks"]}) == 2                   # two processes really ran
    assert d["exchange_bytes_per_step_per_rank"] > 0
    modes = d["ms_per_step_by_exchange_mode"]
    assert set(modes) == {"pipelined", "sync", "none"} and all(v > 0 for v in modes.values())
    assert d["exposed_exchange_ms_per_step"] is not None
    # round 4: every collective of the exchange timed ALONE, and the stage-1 step (features-only backward,
    # ins_feat all-reduce) as an untimed extra of the same command
    ca = d["collectives_alone"]
    assert {"grad_bucket_sum_allreduce", "sh_factor_allgather", "sh_rebuild_kernel", "radii_max_allreduce",
            "stage1_ins_feat_sum_allreduce"} <= set(ca), ca
    assert all(v["ms"] > 0 for v in ca.values())
    assert ca["stage1_ins_feat_sum_allreduce"]["bytes"] == 100_000 * 24
    s1 = d["stage1"]
    assert s1["ms_per_step"] > 0 and s1["exchange_bytes_per_step_per_rank"] == 100_000 * 24


def test_bench_stage1_switch_two_ranks(gpu_device):
    """`bench.py --gpus 2 --stage1`: the TIMED step is the stage-1 training step (everything but ins_feat detached,
    train.py:431-436: fused forward, features-only backward, one SUM all-reduce of dL/d ins_feat); the all-gradient step then
    comes as the untimed extra.  Two ranks over gloo on the one-GPU box."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, OGS_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "OGS_DP_FORCE_COLLECTIVES"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--stage1",
           "--workload", "C2-100k-800", "--no-cpu-baseline", "--no-kmeans"]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.strip()][-1])
    assert out["n_gpus"] == 2 and out["value"] > 0 and "STAGE-1" in out["config"]["workload"]
    d = out["dist"]
    assert d["exchange_bytes_per_step_per_rank"] == 100_000 * 24
    # (which kernel dominates is not asserted: two ranks share the one GPU of the test box, and an event pair around a launch
    # then brackets the other rank's kernels as well)
    assert out["roofline"] is not None and out["roofline"]["algorithmic_bytes_per_launch"] > 0, out["roofline"]
    assert d["stage0_all_gradient_step"]["ms_per_step"] > 0


# ---- densification on the sharded optimizer state (scene/gaussian_model.py:357-510) -----------
_DG = [("xyz", (3,), 1.6e-4), ("f_dc", (1, 3), 2.5e-3), ("f_rest", (15, 3), 1.25e-4), ("opacity", (1,), 0.05),
       ("scaling", (3,), 5e-3), ("rotation", (4,), 1e-3), ("ins_feat", (6,), 1e-3)]


def _densify_sharded_rank(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), OGS_DIST_BACKEND="gloo")
    import torch.distributed as dist
    from opengaussian_amd import densify, dp
    from opengaussian_amd.optim import FusedAdam
    dp.init_from_env("cuda")
    dev = torch.device("cuda", 0)
    P = 911                                              # slice boundaries cut through rows and through tensors
    g = torch.Generator().manual_seed(1)
    init = {n: torch.randn(P, *shape, generator=g) for n, shape, _ in _DG}
    init["scaling"] = torch.randn(P, 3, generator=g) * 1.2 - 3.0
    init["opacity"] = torch.randn(P, 1, generator=g) * 3.0
    init = {n: v.to(dev) for n, v in init.items()}
    lrs = {n: lr for n, _, lr in _DG}
    sharded = dp.ShardedAdam([(n, (P,) + shape) for n, shape, _ in _DG], lrs, dev)
    sharded.load(init)
    ref_params = {n: torch.nn.Parameter(init[n].clone()) for n, _, _ in _DG}
    ref = FusedAdam([{"params": [ref_params[n]], "lr": lrs[n], "name": n} for n, _, _ in _DG], lr=0.0, eps=1e-15)
    st_s = densify.DensifyState(sharded, torch.zeros(P, 1, device=dev), torch.zeros(P, 1, device=dev), torch.zeros(P, device=dev), 0.01)
    st_r = densify.DensifyState(ref, torch.zeros(P, 1, device=dev), torch.zeros(P, 1, device=dev), torch.zeros(P, device=dev), 0.01)
    out = {"rounds": []}

    def steps(n_steps, seed):
        """n optimizer steps: every rank contributes its own per-view gradient; the replicated optimizer gets their sum"""
        for it in range(n_steps):
            n_now = int(st_r.params()["xyz"].shape[0])
            total = {}
            for r in range(world):
                gr = torch.Generator().manual_seed(seed + 10 * it + r)
                for n, shape, _ in _DG:
                    v = (torch.randn(n_now, *shape, generator=gr) * 0.1).to(dev)
                    total[n] = v if n not in total else total[n] + v
                    if r == rank:
                        sharded.params[n].grad = v
            sharded.step()
            for n, _, _ in _DG:
                st_r.params()[n].grad = total[n].clone()
            ref.step()

    def compare(tag):
        worst = 0.0
        lo, hi = sharded.my
        for n, _, _ in _DG:
            a, b = sharded.params[n].detach(), st_r.params()[n].detach()
            assert a.shape == b.shape, (tag, n, a.shape, b.shape)
            worst = max(worst, float((a - b).abs().max()))
            # the owned moment slice == the same flat range of the replicated optimizer's moments
            off, cnt = sharded.offsets[n], a.numel()
            s0, s1 = max(off, lo), min(off + cnt, hi)
            if s1 > s0:
                for mine, key in ((sharded.exp_avg, "exp_avg"), (sharded.exp_avg_sq, "exp_avg_sq")):
                    want = ref.state[st_r.params()[n]][key].reshape(-1)[s0 - off:s1 - off]
                    worst = max(worst, float((mine[s0 - lo:s1 - lo] - want).abs().max()))
        out["rounds"].append((tag, worst, int(sharded.params["xyz"].shape[0])))

    steps(3, 1000)
    compare("after 3 steps")
    for rnd in range(2):
        n_now = int(st_r.params()["xyz"].shape[0])
        gs = torch.Generator().manual_seed(50 + rnd)
        accum, denom = (torch.rand(n_now, 1, generator=gs) * 3.0).to(dev), torch.randint(0, 4, (n_now, 1), generator=gs).float().to(dev)
        radii = (torch.rand(n_now, generator=gs) * 40.0).to(dev)
        for st in (st_s, st_r):
            st.xyz_gradient_accum, st.denom, st.max_radii2D = accum.clone(), denom.clone(), radii.clone()
        # the replicated run draws the split samples; the sharded run must come to the same children on EVERY rank:
        # its own draw differs per rank (different RNG streams), rank 0's is broadcast -> hand the replicated run that one
        torch.manual_seed(7 + rank)                      # per-rank RNG streams, as in separate training processes
        densify.densify_and_prune(st_s, 0.6, 0.005, 4.0, 20)
        # re-run the replicated optimizer with the samples rank 0 drew: recover them from the children it created
        plan = st_s.last_plan
        kind, src = plan["kind"], plan["src_row"]
        # replicated truth with the SAME children: draw with rank 0's seed
        torch.manual_seed(7 + 0)
        densify.densify_and_prune(st_r, 0.6, 0.005, 4.0, 20)
        assert torch.equal(st_r.last_plan["src_row"], src) and torch.equal(st_r.last_plan["kind"], kind)
        compare(f"after densify round {rnd}")
        steps(2, 2000 + 100 * rnd)
        compare(f"after densify round {rnd} + 2 steps")
    # prune_points and reset_opacity on the sharded state
    n_now = int(st_r.params()["xyz"].shape[0])
    mask = (torch.rand(n_now, generator=torch.Generator().manual_seed(9)) < 0.25).to(dev)
    densify.prune_points(st_s, mask); densify.prune_points(st_r, mask)
    densify.reset_opacity(st_s); densify.reset_opacity(st_r)
    compare("after prune_points + reset_opacity")
    steps(1, 3000)
    compare("after a final step")
    extra = {n: torch.randn(5, *shape, generator=torch.Generator().manual_seed(3)).to(dev) for n, shape, _ in _DG}
    densify.cat_tensors_to_optimizer(sharded, extra); densify.cat_tensors_to_optimizer(ref, extra)
    st_r_params = {g_["name"]: g_["params"][0] for g_ in ref.param_groups}
    out["cat"] = max(float((sharded.params[n].detach() - st_r_params[n].detach()).abs().max()) for n, _, _ in _DG)
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_densification_on_sharded_adam_two_ranks_matches_replicated(gpu_device):
    """densify_and_prune / prune_points / reset_opacity / cat_tensors_to_optimizer over dp.ShardedAdam (flat replicated
    parameters, moments only for the owned flat slice) on two ranks == the same calls over the replicated FusedAdam, bit for
    bit, through two densification rounds with optimizer steps in between: parameters on every rank AND the owned moment
    slices (re-partitioned at the new boundaries after every change of the point count)."""
    import torch.multiprocessing as mp
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_densify_sharded_rank, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sizes = set()
    for rank, out in res:
        assert len(out["rounds"]) == 7
        for tag, worst, n in out["rounds"]:
            assert worst == 0.0, (rank, tag, worst)
        assert out["cat"] == 0.0
        sizes.add(tuple(n for _, _, n in out["rounds"]))
    assert len(sizes) == 1                                # both ranks went through the same point counts
    assert len(set(next(iter(sizes)))) >= 3               # and the count really changed (grow, grow, prune)
