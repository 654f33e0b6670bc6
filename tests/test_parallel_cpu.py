"""CPU tier: the N>1 gradient exchange (gan_inpainting_amd/parallel.py) with world_size 2 on gloo.
Each rank fills a flat 'gradient' with rank-dependent values; after GradSync the buffers hold the
SUM and the optimizer scale 1/world turns it into the mean of the per-rank gradients."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import gan_inpainting_amd  # noqa: F401
    from gan_inpainting_amd import parallel
    r, w = parallel.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    sync = parallel.GradSync(bucket_floats=1000, use_side_stream=False)
    n = 4567
    flat = torch.arange(n, dtype=torch.float32) * (rank + 1)
    split = 3000
    sync.launch(flat, split, n)      # "decoder half" first, as the generator backward does
    sync.launch(flat, 0, split)
    sync.wait()
    expect = torch.arange(n, dtype=torch.float32) * sum(range(1, world + 1))
    ok = bool(torch.equal(flat, expect)) and abs(sync.grad_scale() - 1.0 / world) < 1e-12
    assert len(sync.buckets(0, n)) == 5
    # default: one collective per launched range; every network (key) has its own pending list
    s2 = parallel.GradSync(use_side_stream=False)
    assert s2.buckets(10, 4567) == [(10, 4557)] and s2.buckets(5, 5) == []
    fa = torch.full((100,), float(rank + 1))
    fb = torch.full((50,), 10.0 * (rank + 1))
    s2.launch(fa, 0, 60, key="G")
    s2.launch(fb, key="D")
    s2.launch(fa, 60, 100, key="G")
    assert len(s2._pending["G"]) == 2 and len(s2._pending["D"]) == 1
    s2.wait(key="D")
    assert s2._pending["D"] == [] and len(s2._pending["G"]) == 2
    s2.wait(key="G")
    tot = float(sum(range(1, world + 1)))
    ok = ok and bool(torch.all(fa == tot)) and bool(torch.all(fb == 10.0 * tot))
    # replicas start from rank 0's weights and running statistics, whatever each rank's RNG did before
    torch.manual_seed(100 + rank)
    net = torch.nn.Sequential(torch.nn.Conv2d(1, 4, 3), torch.nn.BatchNorm2d(4))
    net[1].running_mean.normal_()
    sync.broadcast_parameters([net])
    torch.manual_seed(100)
    want = torch.nn.Sequential(torch.nn.Conv2d(1, 4, 3), torch.nn.BatchNorm2d(4))
    want[1].running_mean.normal_()
    ok = ok and all(torch.equal(a, b) for a, b in zip(net.state_dict().values(), want.state_dict().values()))
    # fp16 wire format (GradSync(wire='fp16'), off by default): the range travels as fp16(scale * g) and comes back as fp32;
    # values chosen exactly representable so that the SUM is exact, plus small ones that only survive thanks to the scale
    s3 = parallel.GradSync(use_side_stream=False, wire="fp16", wire_scale=1024.0)
    fw = torch.arange(600, dtype=torch.float32) * 2.0 ** -14 * (rank + 1)     # down to 6e-5: below fp16's normal range unscaled;
    keep = fw.clone()                                                          # 3 * 599 < 2048: every sum is exact in fp16
    s3.launch(fw, 0, 300, key="W")
    s3.launch(fw, 300, 600, key="W")
    assert torch.equal(fw, keep)            # nothing is written before wait()
    s3.wait(key="W")
    ok = ok and bool(torch.equal(fw, torch.arange(600, dtype=torch.float32) * 2.0 ** -14 * tot))
    try:
        parallel.GradSync(use_side_stream=False, wire="fp8")
        ok = False
    except ValueError:
        pass
    q.put((rank, ok))
    dist.destroy_process_group()


def test_rendezvous_watchdog_names_the_stuck_rank():
    """A rank whose peers never arrive does not hang: parallel.init_from_env's watchdog says which rank is stuck and the process
    exits with code 5 (GI_RENDEZVOUS_TIMEOUT seconds; the driver's 8-GPU run then fails fast instead of timing out silently)."""
    import subprocess
    code = ("import os, sys; sys.path.insert(0, %r); import gan_inpainting_amd; from gan_inpainting_amd import parallel; "
            "parallel.init_from_env(backend='gloo')" % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="1", WORLD_SIZE="2", GI_RENDEZVOUS_TIMEOUT="3")
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode == 5, (p.returncode, p.stderr[-400:])
    assert "rank 1 is still in the process-group rendezvous" in p.stderr


def test_gradsync_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


def _bench_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      GI_DIST_BACKEND="gloo")
    import bench
    import gan_inpainting_amd  # noqa: F401
    from gan_inpainting_amd import parallel
    r, w = parallel.init_from_env()               # what bench.main() does first under torch.distributed.run
    dt = bench.max_over_ranks(0.5 + rank, w, torch.device("cpu"))      # rank 1 is the slow one
    rate = bench.whole_job_rate(w, bench.BS, 10, dt)
    # identical replicas: the broadcast bench.main() issues for both networks
    flat = torch.full((7,), float(rank))
    dist.broadcast(flat, 0)
    q.put((rank, r, w, dt, rate, float(flat.sum())))
    dist.destroy_process_group()


def test_bench_multi_rank_plumbing_world2_gloo():
    """bench.py's N > 1 path without a GPU: rank / world from the torch.distributed.run environment, MAX over ranks of the
    timed region, whole-job rate = world x per-rank batch x steps / that time, replicas broadcast from rank 0."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    for rank, (rk, r, w, dt, rate, bsum) in enumerate(res):
        assert (rk, r, w) == (rank, rank, 2)
        assert dt == 1.5 and abs(rate - 2 * 32 * 10 / 1.5) < 1e-9 and bsum == 0.0


def _run_bench(extra_env, *flags):
    import json
    import subprocess
    env = dict(os.environ, GI_DIST_BACKEND="gloo", **extra_env)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], env=env, capture_output=True, text=True, timeout=300)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p.returncode, [json.loads(ln) for ln in lines], p.stderr


def test_bench_gpus_flag_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment (how the driver calls it) must produce 2 ranks by
    itself: launch_ranks() -> two child processes with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*, rendezvous, the 1-element
    all-reduce that counts the ranks, ONE JSON line from rank 0 with n_gpus = 2 (the rehearsal workload runs no kernels)."""
    rc, lines, err = _run_bench({}, "--gpus", "2", "--workload", "rehearsal", "--steps", "3", "--warmup", "1")
    assert rc == 0, err
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2 and lines[0]["steps"] == 3
    # rank r sleeps (r + 1) ms per step: the reported time is the slowest rank's
    assert lines[0]["ms_per_step"] >= 2.0


def test_bench_exits_nonzero_when_a_rank_fails():
    rc, lines, err = _run_bench({"GI_BENCH_FAIL_RANK": "1"}, "--gpus", "2", "--workload", "rehearsal", "--steps", "2", "--warmup", "0")
    assert rc != 0 and not lines and "rank 1 exited" in err
