#!/usr/bin/env python3
"""A/B per layer: igemm6 (default) against igemm8 (GI_IGEMM8 = 2) on the halo-resident layers of the headline benchmark, same
process, interleaved rounds, HIP events around back-to-back launches. usage (GPU box): python tools/time_igemm8.py [rounds]"""
import ctypes as C
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
import gan_inpainting_amd  # noqa
from gan_inpainting_amd import backend as B

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
F16 = B.GI_F16
lib, ctx = B.lib(), B.get_ctx()
# (name, kind, n, H (input), cin, cout, relu_in, relu_cend)
LAYERS = [
    ("d2", "conv", 32, 128, 64, 128, 0, 0), ("d3", "conv", 32, 64, 128, 256, 0, 0), ("d4", "conv", 32, 32, 256, 512, 0, 0),
    ("u4", "convT", 32, 16, 1024, 256, 1, 512), ("u3", "convT", 32, 32, 512, 128, 1, 256), ("u2", "convT", 32, 64, 256, 64, 1, 128),
    ("critic conv2", "conv", 64, 128, 64, 128, 0, 0), ("critic conv3", "conv", 64, 64, 128, 256, 0, 0), ("critic conv4", "conv", 64, 32, 256, 512, 0, 0),
    ("critic conv4 dgrad", "convT", 64, 16, 512, 256, 0, 0), ("critic conv3 dgrad", "convT", 64, 32, 256, 128, 0, 0),
    ("critic conv2 dgrad", "convT", 64, 64, 128, 64, 0, 0), ("d2 dgrad", "convT", 32, 64, 128, 64, 0, 0), ("u2 dgrad", "conv", 32, 128, 64, 256, 0, 0),
    ("u3 dgrad", "conv", 32, 64, 128, 512, 0, 0),
]


def make(kind, n, H, cin, cout):
    x = (torch.rand((n, H, H, cin), device="cuda") - 0.3).half()
    if kind == "conv":
        w = ((torch.rand((cout, 4, 4, cin), device="cuda") * 2 - 1) * 0.02)
        packed = torch.empty(cout * 16 * cin, dtype=torch.float16, device="cuda")
        B.check(lib.gi_pack_weights(ctx, F16, B.ptr(w), cout, cin, B.ptr(packed), None))
        out = torch.empty((n, H // 2, H // 2, cout), dtype=torch.float16, device="cuda")
        flop = 2.0 * n * (H // 2) ** 2 * cout * 16 * cin
    else:
        w = ((torch.rand((cin, 4, 4, cout), device="cuda") * 2 - 1) * 0.02)
        packed = torch.empty(cin * 16 * cout, dtype=torch.float16, device="cuda")
        B.check(lib.gi_pack_weights(ctx, F16, B.ptr(w), cin, cout, None, B.ptr(packed)))
        out = torch.empty((n, 2 * H, 2 * H, cout), dtype=torch.float16, device="cuda")
        flop = 2.0 * 4 * n * H * H * cout * 4 * cin
    return x, packed, out, flop


def launch(kind, x, wp, out, n, H, cin, cout, relu, cend):
    ex = B.IgemmEx()
    ex.relu_cend = cend
    fn = lib.gi_conv_s2_forward_ex if kind == "conv" else lib.gi_convT_s2_forward_ex
    B.check(fn(ctx, F16, B.ptr(x), B.ptr(wp), B.ptr(out), n, H, H, cin, cin, cout, cout, relu, 0, None, 0, C.byref(ex)))


def timeit(args, iters=20):
    for _ in range(3):
        launch(*args)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        launch(*args)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


# argv[2] = the GI_IGEMM8 value of the second arm (default 2: every eligible layer on igemm8; 1: only layers with >= 512 workgroups);
# the first arm is GI_IGEMM8 = 0 (igemm6)
arm = int(sys.argv[2]) if len(sys.argv) > 2 else 2
print(f"{'layer':22s} {'igemm6 us':>10s} {'igemm8 us':>10s} {'ratio':>6s} {'TF/s 6':>8s} {'TF/s 8':>8s}   kernels")
for name, kind, n, H, cin, cout, relu, cend in LAYERS:
    x, wp, out, flop = make(kind, n, H, cin, cout)
    args = (kind, x, wp, out, n, H, cin, cout, relu, cend)
    t6, t8, k6, k8 = [], [], "", ""
    for _ in range(rounds):
        B.set_option("GI_IGEMM8", 0)
        t6.append(timeit(args)); k6 = B.last_kernel()
        B.set_option("GI_IGEMM8", arm)
        t8.append(timeit(args)); k8 = B.last_kernel()
    B.set_option("GI_IGEMM8", -1)
    a, b = sorted(t6)[len(t6) // 2], sorted(t8)[len(t8) // 2]
    # parity of the two arms on the same inputs (fp32 accumulation in a different K order)
    B.set_option("GI_IGEMM8", 0); launch(*args); o6 = out.float().clone()
    B.set_option("GI_IGEMM8", arm); launch(*args); o8 = out.float()
    B.set_option("GI_IGEMM8", -1)
    rel = float((o6 - o8).norm() / o6.norm())
    print(f"{name:22s} {a:10.1f} {b:10.1f} {b / a:6.2f} {flop / a / 1e6:8.0f} {flop / b / 1e6:8.0f}   {k6} | {k8}   rel diff {rel:.1e}")
