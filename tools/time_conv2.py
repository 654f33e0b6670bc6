#!/usr/bin/env python3
"""The critic's conv2 (64 images, 128x128x64 -> 64x64x128, igemm8's gather mode) and the generator's d2 (32 images) through
gi_time_conv_s2: average us per launch. For tools/ablate_igemm8_gather.sh."""
import ctypes as C
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
import gan_inpainting_amd  # noqa
from gan_inpainting_amd import backend as B

F16 = B.GI_F16
lib, ctx = B.lib(), B.get_ctx()
res = []
for name, n in (("critic conv2", 64), ("generator d2", 32)):
    H, cin, cout = 128, 64, 128
    x = (torch.rand((n, H, H, cin), device="cuda") - 0.3).half()
    w = (torch.rand((cout, 4, 4, cin), device="cuda") * 2 - 1) * 0.02
    packed = torch.empty(cout * 16 * cin, dtype=torch.float16, device="cuda")
    B.check(lib.gi_pack_weights(ctx, F16, B.ptr(w), cout, cin, B.ptr(packed), None))
    out = torch.empty((n, H // 2, H // 2, cout), dtype=torch.float16, device="cuda")
    ms = C.c_float(0)
    B.check(lib.gi_time_conv_s2(ctx, F16, B.ptr(x), B.ptr(packed), B.ptr(out), n, H, H, cin, cin, cout, cout, 100, C.byref(ms)))
    res.append(f"{name} {ms.value * 1e3:6.1f} us [{B.last_kernel()}]")
print("   ".join(res))
