#!/usr/bin/env python3
"""Per-step cost and per-tile fixed cost of igemm8's sub-pixel-phase mode: the u3 shape (ConvTranspose2d cin -> 128, 32x32 -> 64x64, n = 32:
512 workgroups, one round) with cin = 256 .. 2048, i.e. 32 .. 256 K steps per workgroup; time = fixed + steps * per_step (least squares).
usage (GPU box): python tools/kloop_fit.py"""
import ctypes as C
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np
import torch
import gan_inpainting_amd  # noqa
from gan_inpainting_amd import backend as B

F16 = B.GI_F16
lib, ctx = B.lib(), B.get_ctx()


def run(kind, n, H, cin, cout, relu, iters=30):
    x = (torch.rand((n, H, H, cin), device="cuda") - 0.3).half()
    if kind == "convT":
        w = ((torch.rand((cin, 4, 4, cout), device="cuda") * 2 - 1) * 0.02)
        packed = torch.empty(cin * 16 * cout, dtype=torch.float16, device="cuda")
        B.check(lib.gi_pack_weights(ctx, F16, B.ptr(w), cin, cout, None, B.ptr(packed)))
        out = torch.empty((n, 2 * H, 2 * H, cout), dtype=torch.float16, device="cuda")
        fn = lib.gi_convT_s2_forward_ex
    else:
        w = ((torch.rand((cout, 4, 4, cin), device="cuda") * 2 - 1) * 0.02)
        packed = torch.empty(cout * 16 * cin, dtype=torch.float16, device="cuda")
        B.check(lib.gi_pack_weights(ctx, F16, B.ptr(w), cout, cin, B.ptr(packed), None))
        out = torch.empty((n, H // 2, H // 2, cout), dtype=torch.float16, device="cuda")
        fn = lib.gi_conv_s2_forward_ex
    ex = B.IgemmEx()
    ex.relu_cend = cin if relu else 0

    def launch():
        B.check(fn(ctx, F16, B.ptr(x), B.ptr(packed), B.ptr(out), n, H, H, cin, cin, cout, cout, relu, 0, None, 0, C.byref(ex)))
    for _ in range(3):
        launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        launch()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3, B.last_kernel()


for kind, H, relu, taps in (("convT", 32, 1, 4), ("convT", 32, 0, 4), ("conv", 128, 0, 16)):
    xs, ys = [], []
    for cin in ((256, 512, 1024, 2048) if kind == "convT" else (64, 128, 256, 512)):
        t, k = run(kind, 32, H, cin, 128, relu)
        steps = taps * cin // 32
        xs.append(steps); ys.append(t)
        print(f"{kind} relu={relu} cin={cin:5d} steps={steps:4d}  {t:7.1f} us   {k}")
    a, b = np.polyfit(xs, ys, 1)
    floor = 32 * 16 * 2 / 1.97e3      # 32 MFMAs x 16 cycles x 2 workgroups per SIMD pair at 1.97 GHz, us per step of one workgroup
    print(f"  -> fixed {b:.1f} us, per step {a * 1e3:.0f} ns (MFMA floor {floor * 1e3:.0f} ns per step with two workgroups per CU): loop efficiency {floor / a:.2f}")
