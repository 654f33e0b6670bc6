#!/usr/bin/env python3
"""The critic's conv2 input gradient (igemm8<3>, 64 images, 64x64 -> 128x128, 128 -> 64 channels) with and without the fused
LeakyReLU-backward mask; with the ablation build GI_EPI_DBG=1 keeps the stores inside a 64 KiB window and GI_EPI_DBG=2 the mask
loads (3: both): what the 134 MB store and the 134 MB mask read cost inside the launch. mask=2: the mask as 64-bit sign words. usage: python tools/time_mask.py"""
import ctypes as C
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
import gan_inpainting_amd  # noqa
from gan_inpainting_amd import backend as B

F16 = B.GI_F16
lib, ctx = B.lib(), B.get_ctx()
for name, n, H, cin, cout in (("critic conv2 dgrad", 64, 64, 128, 64), ("generator d2 dgrad", 32, 64, 128, 64)):
    x = (torch.rand((n, H, H, cin), device="cuda") - 0.3).half()
    w = ((torch.rand((cin, 4, 4, cout), device="cuda") * 2 - 1) * 0.02)
    packed = torch.empty(cin * 16 * cout, dtype=torch.float16, device="cuda")
    B.check(lib.gi_pack_weights(ctx, F16, B.ptr(w), cin, cout, None, B.ptr(packed)))
    out = torch.empty((n, 2 * H, 2 * H, cout), dtype=torch.float16, device="cuda")
    mask = (torch.rand((n, 2 * H, 2 * H, cout), device="cuda") - 0.5).half()
    pos = (mask > 0).to(torch.int32).reshape(-1, 8, 8)
    words = (pos << torch.arange(8, device="cuda", dtype=torch.int32)).sum(-1).to(torch.uint8).contiguous()
    res = []
    addt = (torch.rand((n, 2 * H, 2 * H, cout), device="cuda") - 0.5).half()
    for use_mask in (0, 1, 2, 3, 4):       # 3 / 4: as 1 / 2 with a second gradient (`add`)
        def launch():
            ex = B.IgemmEx()
            if use_mask:
                ex.mask = B.ptr(mask); ex.ldmask = cout; ex.mask_slope = 0.2
            if use_mask in (2, 4):
                ex.mask_bits = B.ptr(words)
            if use_mask >= 3:
                ex.add = B.ptr(addt); ex.ldadd = cout
            B.check(lib.gi_convT_s2_forward_ex(ctx, F16, B.ptr(x), B.ptr(packed), B.ptr(out), n, H, H, cin, cin, cout, cout, 0, 0, None, 0, C.byref(ex)))
            return ex.mask_applied
        for _ in range(3):
            applied = launch()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            launch()
        e1.record()
        torch.cuda.synchronize()
        res.append(f"mask={use_mask} (applied {applied}) {e0.elapsed_time(e1) / 30 * 1e3:6.1f} us [{B.last_kernel()}]")
    print(f"{name}: " + "   ".join(res))
