"""CPU, world_size 2 over gloo: the one-view-per-GPU exchange (gradient bucket all-reduce, densification
statistics) and the view sharding logic."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from opengaussian_amd import dp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    r, w, _ = dp.init_from_env("cpu")
    assert (r, w) == (rank, world)
    P = 37
    shapes = [(P, 3), (P, 16, 3), (P, 1)]
    bucket = dp.GradBucket(shapes, "cpu", average=True)
    g = torch.Generator().manual_seed(rank)
    grads = [torch.randn(*s, generator=g) for s in shapes]
    grads[2] = None                         # a family without gradient must contribute zeros
    bucket.pack(grads)
    bucket.allreduce_async()
    out = bucket.wait()
    # expected: mean over ranks
    exp = []
    for i, s in enumerate(shapes):
        acc = torch.zeros(*s)
        for rr in range(world):
            gg = torch.Generator().manual_seed(rr)
            gs = [torch.randn(*t, generator=gg) for t in shapes]
            if i != 2:
                acc += gs[i]
        exp.append(acc / world)
    ok = all(torch.allclose(a, b, atol=1e-6) for a, b in zip(out, exp))
    # densification statistics: SUM of norms / counts, MAX of radii
    m2 = torch.zeros(P, 3); m2[:, 0] = rank + 1.0
    radii = torch.arange(P, dtype=torch.int32) * (rank + 1) % 7
    norm, cnt, rmax = dp.reduce_densification_stats(m2, radii)
    exp_norm = sum((rr + 1.0) * ((torch.arange(P) * (rr + 1) % 7) > 0).float() for rr in range(world))
    exp_cnt = sum(((torch.arange(P) * (rr + 1) % 7) > 0).float() for rr in range(world))
    exp_max = torch.stack([(torch.arange(P) * (rr + 1) % 7) for rr in range(world)]).max(0)[0].to(torch.int32)
    ok = ok and torch.allclose(norm, exp_norm) and torch.allclose(cnt, exp_cnt) and torch.equal(rmax, exp_max)
    # the same statistics riding inside a (non-averaged) SUM bucket + the separate MAX collective (bench.py's form)
    b2 = dp.GradBucket([(2, P)], "cpu", average=False)
    b2.pack([dp.densification_stats(m2, radii)])
    b2.allreduce_async()
    s = b2.wait()[0]
    rm, work = dp.reduce_max_radii(radii, async_op=True)
    if work is not None:
        work.wait()
    ok = ok and torch.allclose(s[0], exp_norm) and torch.allclose(s[1], exp_cnt) and torch.equal(rm, exp_max)
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_grad_bucket_and_stats_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]


def test_single_process_bucket_is_identity():
    b = dp.GradBucket([(5, 3), (5,)], "cpu")
    g = [torch.arange(15.0).view(5, 3), torch.ones(5)]
    b.pack(g)
    b.allreduce_async()
    out = b.wait()
    assert torch.equal(out[0], g[0]) and torch.equal(out[1], g[1])


def test_shard_views():
    assert dp.shard_views(8, 3, 8) == [3]
    assert dp.shard_views(8, 1, 2) == [1, 3, 5, 7]
    assert sorted(sum((dp.shard_views(5, r, 4) for r in range(4)), [])) == list(range(5))


def _gather_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    dp.init_from_env("cpu")
    P = 23
    ex = dp.ShGradExchange(P, 16, "cpu")
    mine = torch.arange(P * 3, dtype=torch.float32).view(P, 3) * (rank + 1)
    ex.gather_async(mine)
    g = ex.wait()
    ok = g.shape == (world, P, 3) and all(torch.equal(g[r], torch.arange(P * 3, dtype=torch.float32).view(P, 3) * (r + 1))
                                          for r in range(world))
    try:                                    # the rebuild is a HIP kernel: it must refuse CPU tensors, not fall back
        ex.rebuild(torch.zeros(P, 3), torch.zeros(world, 3), 3)
        ok = False
    except RuntimeError:
        pass
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_sh_factor_all_gather_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gather_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]


def test_bench_self_launch_builds_a_torchrun_child_and_never_touches_the_gpu(monkeypatch, capsys):
    """bench.py --gpus N > 1 with WORLD_SIZE unset: the parent must start `python -m torch.distributed.run
    --nproc-per-node N bench.py <same args>` as a CHILD (subprocess, no exec), relay the JSON line and return its
    code -- without initialising the GPU itself (checked here by making any CUDA init raise)."""
    import importlib.util
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}

    def fake_run(cmd, **kw):
        seen["cmd"], seen["kw"] = cmd, kw
        return subprocess.CompletedProcess(cmd, 3, stdout='noise\n{"metric": "m", "n_gpus": 4}\n')

    def boom(*a, **k):
        raise AssertionError("the self-launching parent touched the GPU")
    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(torch.cuda, "is_available", boom)
    monkeypatch.setattr(torch.cuda, "_lazy_init", boom)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "2", "--warmup", "1"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 3                                          # the children's return code
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "127.0.0.1" in cmd
    i = cmd.index(os.path.join(root, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "2", "--warmup", "1"]
    assert capsys.readouterr().out.strip() == '{"metric": "m", "n_gpus": 4}'
