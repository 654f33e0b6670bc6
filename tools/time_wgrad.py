#!/usr/bin/env python3
"""Time the 4x4/s2 weight-gradient GEMM (gi_wgrad_s2) on the critic's and generator's layer shapes.
usage: [GI_WGRAD_BLOCKS=N] python tools/time_wgrad.py [iters]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import gan_inpainting_amd  # noqa: E402,F401
from gan_inpainting_amd import backend as B  # noqa: E402

SHAPES = [  # (name, n, Hs, Ws, ca, cb): dW[ca][16][cb] = sum S[n,Hs,Ws,ca] x L[n,2Hs,2Ws,cb]
    ("D conv2 / G d2  128x64 @64", 32, 64, 64, 128, 64),
    ("D conv3 / G d3 256x128 @32", 32, 32, 32, 256, 128),
    ("D conv4 / G d4 512x256 @16", 32, 16, 16, 512, 256),
    ("G u3 512x128 @32", 32, 32, 32, 512, 128),
    ("G u2 256x64 @64", 32, 64, 64, 256, 64),
    ("G d5 512x512 @8", 32, 8, 8, 512, 512),
]


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    lib, ctx = B.lib(), B.get_ctx()
    for name, n, hs, ws, ca, cb in SHAPES:
        S = (torch.rand((n, hs, ws, ca), device="cuda") - 0.5).half()
        L = (torch.rand((n, 2 * hs, 2 * ws, cb), device="cuda") - 0.5).half()
        dW = torch.zeros(ca * 16 * cb, device="cuda")
        if os.environ.get("WGRAD_SCRATCH", "1") == "1":
            nb = lib.gi_wgrad_s2_scratch_bytes(B.GI_F16, n, hs, ws, ca, cb)
            scr = torch.empty(max(nb // 4, 4), dtype=torch.float32, device="cuda")
            call = lambda: B.check(lib.gi_wgrad_s2_ws(ctx, B.GI_F16, B.ptr(S), B.ptr(L), B.ptr(dW), n, hs, ws, ca, ca, cb, cb, 0, 1.0,  # noqa: E731
                                                      B.ptr(scr), nb))
        else:
            call = lambda: B.check(lib.gi_wgrad_s2(ctx, B.GI_F16, B.ptr(S), B.ptr(L), B.ptr(dW), n, hs, ws, ca, ca, cb, cb, 0, 1.0))  # noqa: E731
        for _ in range(3):
            call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            call()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        flop = 2.0 * n * hs * ws * ca * 16 * cb
        print(f"{name:30s} {ms * 1e3:8.1f} us  {flop / ms / 1e9:7.1f} TFLOP/s")


if __name__ == "__main__":
    main()
