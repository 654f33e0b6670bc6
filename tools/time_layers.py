#!/usr/bin/env python3
"""Time single transposed-convolution layers (gi_time_convT_s2) for A/B comparisons of kernel variants.
usage: python tools/time_layers.py [iters]   (environment selects the variant, e.g. GI_IGEMM3_BM=512)"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import gan_inpainting_amd  # noqa: E402,F401
from gan_inpainting_amd import backend as B  # noqa: E402

SHAPES = [  # (name, n, hs, ws, ca, cb)  ConvTranspose2d(ca -> cb) forward, small side hs x ws
    ("u2 256->64 @64", 32, 64, 64, 256, 64),
    ("u3 512->128 @32", 32, 32, 32, 512, 128),
    ("u4 1024->256 @16", 32, 16, 16, 1024, 256),
    ("d2-dgrad 128->64 @64", 32, 64, 64, 128, 64),
    ("d3-dgrad 256->128 @32", 32, 32, 32, 256, 128),
]


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    for name, n, hs, ws, ca, cb in SHAPES:
        x = (torch.rand((n, hs, ws, ca), device="cuda") - 0.3).half()
        w = ((torch.rand((ca, 4, 4, cb), device="cuda") * 2 - 1) * 0.02)
        phase = torch.empty(ca * 16 * cb, dtype=torch.float16, device="cuda")
        B.check(B.lib().gi_pack_weights(B.get_ctx(), B.GI_F16, B.ptr(w), ca, cb, None, B.ptr(phase)))
        out = torch.empty((n, 2 * hs, 2 * ws, cb), dtype=torch.float16, device="cuda")
        ms = C.c_float()
        B.check(B.lib().gi_time_convT_s2(B.get_ctx(), B.GI_F16, B.ptr(x), B.ptr(phase), B.ptr(out), n, hs, ws, ca, ca, cb, cb, iters,
                                         C.byref(ms)))
        flop = 2.0 * 4 * (n * hs * ws) * cb * (4 * ca)
        print(f"{name:24s} {ms.value * 1e3:8.1f} us  {flop / ms.value / 1e9:7.1f} TFLOP/s  checksum {float(out.float().abs().mean()):.6f}")


if __name__ == "__main__" and not (len(sys.argv) > 2 and sys.argv[2] == "conv"):
    main()


def conv_forward_times(iters=30):
    """Conv2d 4x4/s2 forward (the stride-2 gather mode) on the encoder / critic shapes."""
    shapes = [  # (name, n, Hs, Ws (output), cin, cout)
        ("d2 64->128 @64", 32, 64, 64, 64, 128),
        ("d3 128->256 @32", 32, 32, 32, 128, 256),
        ("d4 256->512 @16", 32, 16, 16, 256, 512),
        ("u2-dgrad 64->256 @64", 32, 64, 64, 64, 256),
        ("u3-dgrad 128->512 @32", 32, 32, 32, 128, 512),
    ]
    lib, ctx = B.lib(), B.get_ctx()
    for name, n, hs, ws, cin, cout in shapes:
        x = (torch.rand((n, 2 * hs, 2 * ws, cin), device="cuda") - 0.3).half()
        w = ((torch.rand((cout, 4, 4, cin), device="cuda") * 2 - 1) * 0.02)
        packed = torch.empty(cout * 16 * cin, dtype=torch.float16, device="cuda")
        B.check(lib.gi_pack_weights(ctx, B.GI_F16, B.ptr(w), cout, cin, B.ptr(packed), None))
        out = torch.empty((n, hs, ws, cout), dtype=torch.float16, device="cuda")
        call = lambda: B.check(lib.gi_conv_s2_forward(ctx, B.GI_F16, B.ptr(x), B.ptr(packed), B.ptr(out), n, 2 * hs, 2 * ws, cin, cin, cout, cout,  # noqa: E731
                                                      0, 0, None, 0))
        for _ in range(3):
            call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            call()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        flop = 2.0 * n * hs * ws * cout * 16 * cin
        print(f"{name:24s} {ms * 1e3:8.1f} us  {flop / ms / 1e9:7.1f} TFLOP/s  checksum {float(out.float().abs().mean()):.6f}")


if __name__ == "__main__" and len(sys.argv) > 2 and sys.argv[2] == "conv":
    conv_forward_times(int(sys.argv[1]))
