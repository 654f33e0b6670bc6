#!/usr/bin/env python3
"""Headline benchmark: training images/sec of the WGAN inpainting loop at 256x256, bs=32 per GPU
(BASELINE.json configs[3], `wgan_rmse`: U-Net generator + PatchGAN critic, RMSprop, weight clipping,
RMSE reconstruction, G updated every 5th batch as in wgan_l1.py:157-163 steady state), fp16 MFMA
compute with fp32 master weights, synthetic masked-image batches resident in HBM.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

With --gpus N > 1 and no WORLD_SIZE in the environment (a plain `python bench.py --gpus 8`) the process starts N rank
processes itself (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, one per GPU, RCCL through torch.distributed "nccl") BEFORE it
touches the GPU, relays rank 0's JSON line and exits non-zero if any rank fails (launch_ranks). `n_gpus` in the line is the
number of ranks that answered a 1-element SUM all-reduce, not the flag.

Prints ONE JSON line on rank 0. Extra objects:
  roofline      dominant kernel (the fp16 MFMA implicit-GEMM transposed conv on the u3 shape), timed live
                with HIP events on the kernel's stream: algorithmic FLOP / average launch time vs 2.5 PFLOP/s
  cpu_baseline  the CPU oracle (port of the reference's torch path) on this host's cores, bounded sample
  --kernel-only runs just the dominant-kernel launches (what profiles/*kernel* summarises with rocprofv3).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

H = W = 256
BS = 32
G_EVERY = 5            # wgan_l1.py:157-163 steady-state period (update_g_every)
PEAK_F16_TFLOPS = 2500.0   # MI355X dense fp16/bf16 MFMA (guides/MI355X_MICROARCH.md)
# algorithmic work, SURVEY.md 8d: G forward 11.845 GFLOP/img, D forward 3.258 GFLOP/img at 256x256
F_G, F_D = 11.845e9, 3.258e9


def synth(n, seed, device):
    g = torch.Generator(device="cpu").manual_seed(seed)
    ground = torch.rand((n, 1, H, W), generator=g)
    mask = torch.zeros((n, 1, H, W))
    for i in range(n):
        h = int(torch.randint(H // 8, H // 2 + 1, (1,), generator=g))
        w = int(torch.randint(W // 8, W // 2 + 1, (1,), generator=g))
        y0 = int(torch.randint(0, H - h + 1, (1,), generator=g))
        x0 = int(torch.randint(0, W - w + 1, (1,), generator=g))
        mask[i, 0, y0:y0 + h, x0:x0 + w] = 1.0
    return ground.to(device), mask.to(device)


def launch_ranks(n, argv):
    """Start n fresh rank processes of this script (children never re-exec; this parent makes no GPU call), wait for them,
    return the exit code: 0 only if every rank exited 0. Rank 0 writes to our stdout, the other ranks' stdout is dropped
    (they print nothing by contract), stderr is shared. A rank that fails takes the others down (exact PIDs)."""
    import socket
    import subprocess
    import tempfile
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    # watchdog: every rank drops rank<r>.ready into this directory once the process group exists (parallel.init_from_env); ranks
    # that have not after GI_RENDEZVOUS_TIMEOUT + 30 s are named and the run ends non-zero (the ranks' own watchdogs fire first)
    status = tempfile.mkdtemp(prefix="gi_ranks_")
    limit = float(os.environ.get("GI_RENDEZVOUS_TIMEOUT", "120")) + 30.0
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), GI_RANK_STATUS_DIR=status)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    live = list(procs)
    t_start, all_up = time.monotonic(), False
    while live:
        time.sleep(0.05)
        if not all_up and rc == 0:
            up = [os.path.exists(os.path.join(status, f"rank{r}.ready")) for r in range(n)]
            all_up = all(up)
            if not all_up and time.monotonic() - t_start > limit:
                stuck = [r for r in range(n) if not up[r]]
                print(f"bench.py: ranks {stuck} of {n} have not joined the process group after {limit:.0f} s; stopping the run", file=sys.stderr)
                rc = 4
                for q in live:
                    q.terminate()
        for p in list(live):
            c = p.poll()
            if c is None:
                continue
            live.remove(p)
            if c != 0 and rc == 0:
                rc = c if c > 0 else 1
                print(f"bench.py: rank {procs.index(p)} exited with {c}; stopping the other ranks", file=sys.stderr)
                for q in live:
                    q.terminate()
    for p in procs:
        try:
            p.wait(timeout=30)
        except subprocess.TimeoutExpired:
            p.kill()
    import shutil
    shutil.rmtree(status, ignore_errors=True)
    return rc


def ranks_alive(world, device):
    """Number of ranks that take part in a collective: a 1-element SUM all-reduce (so a run that silently lost ranks cannot
    report the --gpus flag as n_gpus)."""
    if world == 1:
        return 1
    one = torch.ones(1, device=device, dtype=torch.float32)
    torch.distributed.all_reduce(one, op=torch.distributed.ReduceOp.SUM)
    return int(round(float(one.item())))


def rehearsal_workload(args):
    """Plumbing only, no kernels and NOT a measurement (tests/test_parallel_cpu.py, GI_DIST_BACKEND=gloo on a box without
    a GPU): the launch, rendezvous, rank count, barrier / max-over-ranks timing and the one-line report of the real path."""
    import gan_inpainting_amd  # noqa: F401
    from gan_inpainting_amd import parallel
    rank, world = parallel.init_from_env()
    if os.environ.get("GI_BENCH_FAIL_RANK") == str(rank):
        sys.exit(3)
    dev = torch.device("cpu")
    alive = ranks_alive(world, dev)
    if world > 1:
        torch.distributed.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.001 * (rank + 1))
    if world > 1:
        torch.distributed.barrier()
    dt = max_over_ranks(time.perf_counter() - t0, world, dev)
    if rank == 0:
        print(json.dumps({"metric": "rehearsal (no kernels)", "value": whole_job_rate(alive, BS, args.steps, dt), "unit": "images/sec",
                          "n_gpus": alive, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "none", "data": "none",
                          "config": {"workload": "rehearsal of the multi-rank launch path; measures nothing"}}))
    if world > 1:
        torch.distributed.destroy_process_group()


def max_over_ranks(dt, world, device):
    """The contract's timing rule: the slowest rank's time for the K steps."""
    if world > 1:
        tt = torch.tensor([dt], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt.item())
    return dt


def whole_job_rate(world, per_rank_batch, steps, dt):
    """images/s over ALL ranks (weak scaling: every rank processes per_rank_batch images per step)."""
    return world * per_rank_batch * steps / dt


PEAK_F32_TFLOPS = 157.3    # MI355X f32-input MFMA (guides/MI355X_MICROARCH.md)


def dominant_kernel(iters, dtype="fp16", n=None, hs=32):
    """u3 of the generator: ConvTranspose2d(512 -> 128) with the fused input ReLU, i.e. four sub-pixel GEMMs of
    M = n * hs * hs, N = 128, K = 2048. Headline: 256x256 / bs=32 -> hs = 32. Timed with HIP events on the kernel's stream
    inside the library (gi_time_convT_s2)."""
    from gan_inpainting_amd import backend as B
    import ctypes as C
    code = B.dtype_code(dtype)
    n = BS if n is None else n
    ws, ca, cb = hs, 512, 128
    x = (torch.rand((n, hs, ws, ca), device="cuda") - 0.3).to(B.torch_dtype(code))
    w = ((torch.rand((ca, 4, 4, cb), device="cuda") * 2 - 1) * 0.02)
    phase = torch.empty(ca * 16 * cb, dtype=B.torch_dtype(code), device="cuda")
    B.check(B.lib().gi_pack_weights(B.get_ctx(), code, B.ptr(w), ca, cb, None, B.ptr(phase)))
    out = torch.empty((n, 2 * hs, 2 * ws, cb), dtype=B.torch_dtype(code), device="cuda")
    ms = C.c_float()
    B.check(B.lib().gi_time_convT_s2(B.get_ctx(), code, B.ptr(x), B.ptr(phase), B.ptr(out), n, hs, ws, ca, ca, cb, cb, iters,
                                     C.byref(ms)))
    flop = 2.0 * 4 * (n * hs * ws) * cb * (4 * ca)
    nbytes = (x.numel() + phase.numel() + out.numel()) * x.element_size()
    return dict(name=f"{B.last_kernel()} ({dtype} implicit GEMM, ConvTranspose2d 512->128 with the fused input ReLU, {hs}x{hs}->{2 * hs}x{2 * hs}, "
                     f"n={n}; generator u3)", ms=ms.value, flop=flop, bytes=nbytes)


def critic_conv2_kernel(iters):
    """The layer family with the most kernel time in the headline benchmark (profiles/r02_summary.md B: igemm6<0,128>, now
    igemm8<0>): the critic's conv2 on the stacked batch, Conv2d(64 -> 128) 128x128 -> 64x64 at n = 64, a 16-tap x 64-channel
    K loop (M = 262144, N = 128, K = 1024)."""
    from gan_inpainting_amd import backend as B
    import ctypes as C
    n, hw, cb, ca = 2 * BS, H // 2, 64, 128
    x = (torch.rand((n, hw, hw, cb), device="cuda") - 0.3).half()
    w = ((torch.rand((ca, 4, 4, cb), device="cuda") * 2 - 1) * 0.02)
    packed = torch.empty(ca * 16 * cb, dtype=torch.float16, device="cuda")
    B.check(B.lib().gi_pack_weights(B.get_ctx(), B.GI_F16, B.ptr(w), ca, cb, B.ptr(packed), None))
    out = torch.empty((n, hw // 2, hw // 2, ca), dtype=torch.float16, device="cuda")
    ms = C.c_float()
    B.check(B.lib().gi_time_conv_s2(B.get_ctx(), B.GI_F16, B.ptr(x), B.ptr(packed), B.ptr(out), n, hw, hw, cb, cb, ca, ca, iters, C.byref(ms)))
    flop = 2.0 * n * (hw // 2) ** 2 * ca * 16 * cb
    return dict(name=f"{B.last_kernel()} (fp16 implicit GEMM, Conv2d 64->128 {hw}x{hw}->{hw // 2}x{hw // 2}, n={n}; critic conv2 on the stacked batch)",
                ms=ms.value, flop=flop, bytes=(x.numel() + packed.numel() + out.numel()) * 2)


def roofline_of(k, dtype="fp16", traffic=None, traffic_src=None):
    peak = PEAK_F16_TFLOPS if dtype == "fp16" else PEAK_F32_TFLOPS
    achieved = k["flop"] / (k["ms"] * 1e-3) / 1e12
    return {"bound": "mfma", "kernel": k["name"], "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
            "avg_launch_ms": k["ms"], "flop_per_launch": k["flop"], "algorithmic_bytes_per_launch": k["bytes"], "traffic": traffic,
            "traffic_source": traffic_src}


def cpu_baseline(workload="wgan_rmse_256", full=True):
    """The oracle's schedule of the workload (oracle/torch_ref.py, fp32: the port of the reference's torch-CPU path) on the host
    cores at the GPU workload's own batch. Protocol: a SHORT thread sweep first (1 warm-up + 1 critic-only + 1
    generator-update batch at 8 / 32 / all host threads: torch-CPU throughput peaks well below the box's core count), then at
    the best thread count >= 3 warm-up + >= 10 timed critic-only batches and >= 3 timed generator-update batches, weighted
    4 : 1 like the GPU cadence. full=False (secondary workloads): 1 warm-up + 2 + 1 timed batches at the best thread count."""
    import numpy as np
    from oracle import params as op
    from oracle import torch_ref as orc
    dual = workload == "dual_d_256"
    gp = workload == "wgan_gp_128"
    c5 = workload == "config5_512"
    n, hw = BS, H

    def setup():
        PG = orc.to_torch(op.make_unet_params(1234))
        PD = orc.to_torch(op.make_patchgan_params(4321, hw, hw))
        ground, mask = op.synth_batch(0x5EED, n, hw, hw)
        masks = {k: torch.from_numpy(v) for k, v in op.synth_dropout_masks(1, 7, n, hw, hw).items()}
        st = dict(PG=PG, PD=PD, ground=torch.from_numpy(ground), mask=torch.from_numpy(mask), masks=masks)
        if dual:
            st["PD2"] = orc.to_torch(op.make_patchgan_params(9876, hw, hw))
            st["oG"] = orc.Adam(orc.trainable(PG))
            st["oD"] = orc.Adam(orc.trainable(PD) + orc.trainable(st["PD2"]))
        else:
            st["oG"], st["oD"] = orc.RMSprop(orc.trainable(PG)), orc.RMSprop(orc.trainable(PD))
        if c5:
            st["PS"] = orc.to_torch(op.make_unet_params(777, ngf=32, out_c=4), requires_grad=False)
            st["PV"] = orc.to_torch(op.make_vgg19_params(99), requires_grad=False)
            st["seg"] = torch.from_numpy(op.synth_segmentation(5, n, 4, hw, hw)[0])
        return st

    def batch(st, upd):
        if dual:      # every batch updates the generator and both discriminators
            return orc.dual_d_step(st["PG"], st["PD"], st["PD2"], st["oG"], st["oD"], st["ground"], st["mask"], 7, st["masks"])
        extra = orc.config5_extra(st["PV"], st["PS"], st["seg"]) if (c5 and upd) else None
        out = orc.wgan_step(st["PG"], st["PD"], st["oG"], st["oD"], st["ground"], st["mask"], 7, st["masks"], upd,
                            recon="l1" if gp else "rmse", extra=extra)
        if gp:        # + the penalty's forward and double backward (the extension has no step schedule in the oracle: the work of
            eps = torch.full((n, 1, 1, 1), 0.5)   # its term is timed beside the clipped WGAN batch, its gradients are dropped)
            orc.set_requires_grad(st["PD"], True)
            pen = orc.gradient_penalty(st["PD"], st["ground"], out["inpainted"], eps, lam=10.0)
            torch.autograd.grad(pen, orc.trainable(st["PD"]), allow_unused=True)
        return out

    def run(warm, n_critic, n_gen):
        st = setup()
        for _ in range(warm):
            batch(st, False)
        t = []
        for upd, reps in ((False, n_critic), (True, n_gen)):
            t0 = time.perf_counter()
            for _ in range(reps):
                batch(st, upd)
            t.append((time.perf_counter() - t0) / max(reps, 1))
        per_batch = t[1] if dual else ((G_EVERY - 1) * t[0] + t[1]) / G_EVERY
        return n / per_batch, t

    threads = torch.get_num_threads()
    sweep = {}
    try:
        for nt in sorted({min(8, threads), min(32, threads), threads}):
            torch.set_num_threads(nt)
            sweep[nt] = run(1, 0 if dual else 1, 1)[0]
        best = max(sweep, key=lambda k: sweep[k])
        torch.set_num_threads(best)
        warm, nc, ng = (3, 10, 3) if full else (1, 2, 1)
        v, t = run(warm, 0 if dual else nc, ng)
    finally:
        torch.set_num_threads(threads)
    what = {"wgan_rmse_256": "wgan_step (RMSE)", "wgan_gp_128": "wgan_step (L1) + gradient_penalty forward / double backward",
            "dual_d_256": "dual_d_step", "config5_512": "wgan_step + config5_extra"}[workload]
    return dict(value=v, unit="images/sec", cores=best, kind="port",
                sample=f"oracle/torch_ref {what} fp32, bs={n} at {hw}x{hw}, {best} torch threads (best of a 1+1+1-batch sweep over "
                       f"{sorted(sweep)} threads): {warm} warm-up + {0 if dual else nc} critic-only batches ({t[0]:.2f} s each) + {ng} "
                       f"batch(es) with G update ({t[1]:.2f} s), weighted {G_EVERY - 1}:1" + ("" if full else " (reduced sample: secondary workload)"),
                sweep={str(k): round(sweep[k], 3) for k in sorted(sweep)})


def ssim_workload(args, dev):
    """Secondary line (not the headline): the fused SSIM metric kernel, one step = ssim(ground, inpainted)
    on a resident 256x256 bs=32 pair. HBM-bound: 8 algorithmic bytes per pixel (two fp32 reads)."""
    from gan_inpainting_amd.lib import pytorch_ssim
    from oracle import params as op          # input generator + CPU oracle (checker / cpu_baseline leg only)
    from oracle import torch_ref as orc
    x, y = (torch.from_numpy(a) for a in op.synth_ssim_pair(0x551, BS, 1, H, W))
    dx, dy = x.to(dev), y.to(dev)
    for _ in range(max(args.warmup, 3)):
        out = pytorch_ssim.ssim(dx, dy)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(args.steps):
        out = pytorch_ssim.ssim(dx, dy)
    e1.record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / args.steps
    ms = e0.elapsed_time(e1) / args.steps
    nbytes = 2 * 4 * BS * H * W
    t0 = time.perf_counter()
    ref = float(orc.ssim(x, y))
    cpu_s = time.perf_counter() - t0
    return {"metric": "ssim_images_per_sec", "value": BS / wall, "unit": "images/s", "n_gpus": 1, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": wall * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "ssim(ground, inpainted) 256x256 bs=32, window 11 (experiment1_global_local_D.py:209)"},
            "roofline": {"bound": "hbm", "achieved": nbytes / (ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                         "frac": nbytes / (ms * 1e-3) / 1e9 / 8000.0, "traffic": None,
                         "note": "3 launches per call (tile kernel + 2 finish kernels), stream time per call"},
            "cpu_baseline": {"value": BS / cpu_s, "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
                             "sample": "one oracle.ssim call on the same pair"},
            "value_check": {"gpu": float(out), "oracle": ref}}


def vgg_workload(args, dev):
    """Secondary line (not the headline): perceptual + style terms of BASELINE configs[4] (512x512, 8 image
    pairs per GPU): VGG-19 features of output and target (16 images), 5 taps, Gram matrices. One step = one
    perceptual_and_style_loss call. Algorithmic work: 203.8 GFLOP per image x 16."""
    from gan_inpainting_amd.lib.models import networks
    n, hw = 8, 512
    torch.manual_seed(99)
    vgg = networks.VGG19Wrapper(max_pairs=n).to(dev)
    out, tgt = torch.rand(n, 1, hw, hw, device=dev), torch.rand(n, 1, hw, hw, device=dev)
    for _ in range(max(args.warmup, 2)):
        r = vgg.perceptual_and_style(out, tgt, 0.01, 0.01)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(args.steps):
        r = vgg.perceptual_and_style(out, tgt, 0.01, 0.01)
    e1.record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / args.steps
    ms = e0.elapsed_time(e1) / args.steps
    flop = 203.8e9 * 2 * n
    return {"metric": "vgg_pairs_per_sec", "value": n / wall, "unit": "image pairs/s", "n_gpus": 1, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": wall * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16", "data": "synthetic",
            "config": {"workload": "perceptual_and_style_loss 512x512, 8 pairs (wgan_perceptual_style_faceparsing.py:216), random-init VGG-19"},
            "roofline": {"bound": "mfma", "achieved": flop / (ms * 1e-3) / 1e12, "peak": PEAK_F16_TFLOPS, "unit": "TFLOP/s",
                         "frac": flop / (ms * 1e-3) / 1e12 / PEAK_F16_TFLOPS, "traffic": None,
                         "note": "whole call (13 convolutions with the pools in their epilogues + 5 Gram / perceptual passes + reductions), conv FLOP only"},
            "losses": {"perceptual": float(r[0]), "style": float(r[1])}}


def dual_d_workload(args, dev):
    """Secondary line: BASELINE configs[2], experiment1_global_local_D 256x256 bs=32 fp16 (generator + global and
    local LSGAN discriminators, Adam, lambda = 300 on global / masked RMSE, SSIM metric per batch as in the plugin)."""
    from gan_inpainting_amd import optim, trainer
    from gan_inpainting_amd.lib import pytorch_ssim
    from gan_inpainting_amd.lib.models import networks
    torch.manual_seed(1234)
    G = networks.get_network("generator", "unet", dtype=args.dtype).to(dev)
    Dg = networks.PatchGANDiscriminator(sigmoid=True, image_size=H, dtype=args.dtype).to(dev)
    Dl = networks.PatchGANDiscriminator(sigmoid=True, image_size=H, dtype=args.dtype).to(dev)
    oG = optim.Adam(G.parameters(), lr=0.0002, betas=(0.5, 0.999))
    oD = optim.Adam(optim.chain(Dl.parameters(), Dg.parameters()), lr=0.0002, betas=(0.5, 0.999))
    step = trainer.DualDStep(G, Dg, Dl, oG, oD)
    batches = [synth(BS, 0x5EED0000 + i, dev) for i in range(4)]

    def one(i):
        g, m = batches[i % 4]
        L = step(g, m)
        L["ssim"] = pytorch_ssim.ssim(g, step.gen)
        return L

    for i in range(args.warmup):
        one(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        L = one(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    k = dominant_kernel(args.kernel_iters, "fp16", BS, 32)
    extra = {"roofline": roofline_of(k, "fp16")}
    if not args.no_cpu_baseline:
        extra["cpu_baseline"] = cpu_baseline("dual_d_256", full=False)
    return {**extra, "metric": "training images/sec at 256x256 bs=32/GPU (experiment1_global_local_D)", "value": BS * args.steps / dt,
            "unit": "images/sec", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16" if args.dtype == "fp16" else "f32",
            "data": "synthetic",
            "config": {"workload": "experiment1_global_local_D 256x256 bs=32 fp16 (BASELINE.json configs[2]): G update + two LSGAN "
                                   "discriminators every batch, SSIM metric per batch"},
            "losses": {k: float(v.item()) for k, v in L.items()}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--preheat", type=int, default=400, help="untimed steps before the W warm-up steps (clock ramp of a cold GPU)")
    ap.add_argument("--dtype", default="fp16")
    ap.add_argument("--workload", default="wgan_rmse_256", choices=["wgan_rmse_256", "wgan_gp_128", "ssim_256", "vgg_512", "config5_512", "dual_d_256", "rehearsal"],
                    help="ssim_256 = the per-batch SSIM metric of experiment1_global_local_D.py:209 at 256x256 bs=32 (SURVEY 8f rank 2); wgan_gp_128 = BASELINE configs[1]: wgan_l1 128x128 bs=16 fp32 with the gradient-penalty extension (not the headline)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--overlap", type=int, default=int(os.environ.get("GI_BENCH_OVERLAP", "1")),
                    help="1: critic on a side HIP stream (trainer.WGANStep overlap=True)")
    ap.add_argument("--kernel-only", action="store_true")
    ap.add_argument("--short-k", action="store_true", help="with --kernel-only: the critic's conv2 (roofline_short_k) instead of the dominant kernel")
    ap.add_argument("--kernel-iters", type=int, default=50)
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    if args.workload == "rehearsal":
        return rehearsal_workload(args)

    import gan_inpainting_amd  # noqa: F401
    from gan_inpainting_amd import optim, parallel, trainer
    from gan_inpainting_amd.lib.models import networks
    global H, W, BS
    gp = args.workload == "wgan_gp_128"
    c5 = args.workload == "config5_512"
    if gp:
        H = W = 128
        BS = 16
        args.dtype = "fp32"
    if c5:      # BASELINE configs[4]: wgan_perceptual_style_faceparsing 512x512 bs=8/GPU fp16
        H = W = 512
        BS = 8

    rank, world = parallel.init_from_env()
    if world != args.gpus and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
    local = parallel.local_device()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    alive = ranks_alive(world, dev)
    if alive != world:
        print(f"bench.py: {alive} of {world} ranks answered the all-reduce", file=sys.stderr)
        sys.exit(4)

    if args.workload == "ssim_256":
        print(json.dumps(ssim_workload(args, dev)))
        return
    if args.workload == "vgg_512":
        print(json.dumps(vgg_workload(args, dev)))
        return
    if args.workload == "dual_d_256":
        print(json.dumps(dual_d_workload(args, dev)))
        return

    if args.kernel_only:
        k = critic_conv2_kernel(args.kernel_iters) if args.short_k else dominant_kernel(args.kernel_iters)
        print(json.dumps(dict(kernel=k["name"], avg_ms=k["ms"], tflops=k["flop"] / k["ms"] / 1e9)))
        return

    torch.manual_seed(1234)
    G = networks.get_network("generator", "unet", dtype=args.dtype).to(dev)
    torch.manual_seed(4321)
    D = networks.PatchGANDiscriminator(sigmoid=False, image_size=H, dtype=args.dtype).to(dev)
    oG = optim.RMSprop(G.parameters(), lr=0.00005)
    oD = optim.RMSprop(D.parameters(), lr=0.00005)
    sync = parallel.GradSync() if world > 1 else None
    if world > 1:   # identical replicas
        torch.distributed.broadcast(G.flat_params(), 0)
        torch.distributed.broadcast(D.flat_params(), 0)
        G.mark_dirty(), D.mark_dirty()
    if c5:
        import functools
        torch.manual_seed(777)
        seg = networks.UnetGenerator(1, 4, 7, ngf=32, norm_layer=functools.partial(torch.nn.BatchNorm2d, affine=True, track_running_stats=True),
                                     use_dropout='False', dtype=args.dtype).to(dev).eval()
        vgg = networks.VGG19Wrapper(max_pairs=BS).to(dev)
        step = trainer.WGANPerceptualStep(G, D, oG, oD, vgg=vgg, segment_model=seg, clip=0.01, sync=sync, overlap=bool(args.overlap))
        segments = [torch.randint(0, 4, (BS, H // 8, W // 8), device=dev).repeat_interleave(8, 1).repeat_interleave(8, 2).contiguous()
                    for _ in range(4)]
        step_fn = lambda g, m, u, i: step(g, m, u, segment=segments[i % 4])   # noqa: E731
    else:
        step = trainer.WGANStep(G, D, oG, oD, recon="l1" if gp else "rmse", clip=0.01, sync=sync, gp_lambda=10.0 if gp else 0.0,
                                overlap=bool(args.overlap))
        step_fn = lambda g, m, u, i: step(g, m, u)   # noqa: E731
    batches = [synth(BS, 0x5EED0000 + rank * 1000 + i, dev) for i in range(4)]
    torch.cuda.synchronize()
    step.inputs_resident = True    # batches are resident in HBM before the first step is issued

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    it = 0
    # A cold MI355X (first process on an idle box) runs the first second or so at reduced clocks: measured 3.39 ms/step for
    # the first process against 2.50 ms/step for the same binary started right after it. The same untimed steps as the W
    # warm-up ones, a fixed number of them (all ranks issue the same collectives), bring the card to its steady state first.
    for _ in range(args.preheat + args.warmup):
        g, m = batches[it % len(batches)]
        step_fn(g, m, it % G_EVERY == G_EVERY - 1, it)
        it += 1
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        g, m = batches[it % len(batches)]
        step_fn(g, m, it % G_EVERY == G_EVERY - 1, it)
        it += 1
    barrier()
    dt = time.perf_counter() - t0
    dt = max_over_ranks(dt, world, dev)
    losses = {k: float(v.item()) for k, v in step.L.items()}

    if rank != 0:
        return
    if c5:   # secondary workload: its dominant kernel is the same transposed convolution at 512x512 / bs=8 (u3: 64x64 maps)
        line = {"metric": "training images/sec at 512x512 bs=8/GPU (wgan_perceptual_style_faceparsing)",
                "value": alive * BS * args.steps / dt, "unit": "images/sec", "n_gpus": alive, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "f16", "data": "synthetic",
                "config": {"workload": "wgan_perceptual_style_faceparsing 512x512 bs=8/GPU fp16 (BASELINE.json configs[4]): WGAN + "
                                       "global/local RMSE + frozen face-parsing U-Net (ngf=32, random init) + VGG-19 perceptual/style "
                                       "(random init) + TV; G update every 5th batch"}, "losses": losses,
                "roofline": roofline_of(dominant_kernel(args.kernel_iters, "fp16", BS, 64), "fp16")}
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline("config5_512", full=False)
        print(json.dumps(line))
        return
    if gp:   # secondary workload, fp32: the f32-input MFMA path (peak 157.3 TFLOP/s); u3 at 128x128 / bs=16 is 16x16 -> 32x32
        line = {"metric": "training images/sec at 128x128 bs=16/GPU (wgan_l1 + gradient penalty, fp32)",
                "value": alive * BS * args.steps / dt, "unit": "images/sec", "n_gpus": alive, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": "wgan_l1 128x128 bs=16 fp32 + WGAN-GP (BASELINE.json configs[1])"}, "losses": losses,
                "roofline": roofline_of(dominant_kernel(args.kernel_iters, "fp32", BS, 16), "fp32")}
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline("wgan_gp_128", full=False)
        print(json.dumps(line))
        return
    # generator forward latency (train-mode forward as inside the loop), HIP events on the compute stream
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g, m = batches[0]
    masked = g * (1 - m)
    for _ in range(3):
        G._forward_raw(masked)
    e0.record()
    reps = 20
    for _ in range(reps):
        G._forward_raw(masked)
    e1.record()
    torch.cuda.synchronize()
    gen_fwd_ms = e0.elapsed_time(e1) / reps
    # the same in eval mode (running-statistics BatchNorm, no dropout): what eval.py / the evaluation pass run
    G.eval()
    for _ in range(3):
        G._forward_raw(masked)
    e0.record()
    for _ in range(reps):
        G._forward_raw(masked)
    e1.record()
    torch.cuda.synchronize()
    gen_fwd_eval_ms = e0.elapsed_time(e1) / reps
    # inference: eval mode, never differentiated -> BatchNorm folded into the convolutions (gi_net_set_inference)
    for _ in range(3):
        G._forward_raw(masked, inference=True)
    e0.record()
    for _ in range(reps):
        G._forward_raw(masked, inference=True)
    e1.record()
    torch.cuda.synchronize()
    gen_fwd_inf_ms = e0.elapsed_time(e1) / reps
    G.train()

    k = dominant_kernel(args.kernel_iters)
    traffic, traffic_src = None, None
    tp = os.path.join(ROOT, "profiles", "dominant_kernel_traffic.json")
    if os.path.exists(tp):
        try:
            tj = json.load(open(tp))
            traffic = tj.get("hbm_bytes_per_launch")
            # the PMC passes are separate rocprofv3 runs (tools/pmc_dominant.sh): say which round's file this number is
            traffic_src = f"profiles/dominant_kernel_traffic.json (round {tj.get('round')}, kernel {tj.get('kernel')}, commit {tj.get('commit')})"
        except Exception:
            traffic = None
    # algorithmic FLOP of one batch: D-only = F_G + 6 F_D ; with G update + 2 F_G + 2 F_D
    flop_batch = BS * ((F_G + 6 * F_D) + (2 * F_G + 2 * F_D) / G_EVERY)
    out = {
        "metric": "training images/sec at 256x256 bs=32/GPU",
        "value": whole_job_rate(alive, BS, args.steps, dt),
        "unit": "images/sec",
        "n_gpus": alive,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f16" if args.dtype == "fp16" else "f32",
        "data": "synthetic",
        "config": {"workload": "wgan_rmse 256x256 bs=32/GPU (BASELINE.json configs[3]): U-Net G (41.8M) + PatchGAN critic, "
                               "RMSprop 5e-5, clip 0.01, RMSE recon, G update every 5th batch, random-init weights",
                   "global_batch": world * BS, "parallelism": f"dp{world}",
                   "compute": "fp16 MFMA, fp32 accumulate, fp32 master weights" if args.dtype == "fp16" else "fp32 MFMA"},
        "generator_fwd_ms": gen_fwd_ms,
        "generator_fwd_eval_ms": gen_fwd_eval_ms,
        "generator_fwd_inference_ms": gen_fwd_inf_ms,
        "generator_fwd_mfma_frac": BS * F_G / (gen_fwd_ms * 1e-3) / (PEAK_F16_TFLOPS * 1e12),
        "step_algorithmic_tflops": flop_batch / (dt / args.steps) / 1e12,
        "losses": losses,
        "roofline": roofline_of(k, "fp16" if args.dtype == "fp16" else "fp32", traffic, traffic_src),
    }
    if args.dtype == "fp16":   # the short-K layer family with the most kernel time of the benchmark, beside the dominant kernel
        t2, t2_src = None, None
        tp2 = os.path.join(ROOT, "profiles", "short_k_kernel_traffic.json")
        if os.path.exists(tp2):
            try:
                tj = json.load(open(tp2))
                t2 = tj.get("hbm_bytes_per_launch")
                t2_src = f"profiles/short_k_kernel_traffic.json (round {tj.get('round')}, kernel {tj.get('kernel')}, commit {tj.get('commit')})"
            except Exception:
                t2 = None
        out["roofline_short_k"] = roofline_of(critic_conv2_kernel(args.kernel_iters), "fp16", t2, t2_src)
    if not args.no_cpu_baseline and world == 1:   # (the CPU restatement is timed at N = 1 only)
        out["cpu_baseline"] = cpu_baseline()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
