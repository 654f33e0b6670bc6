#!/usr/bin/env python3
"""Generator forward alone (256x256, bs=32, fp16, train mode), N times: run under
`rocprofv3 --kernel-trace --stats` to get the per-kernel chain of one forward."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
import gan_inpainting_amd  # noqa
from gan_inpainting_amd.lib.models import networks

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda")
torch.manual_seed(1)
G = networks.get_network("generator", "unet", dtype="fp16").to(dev).train()
G.always_sync = False
x = torch.rand(32, 1, 256, 256, device=dev)
with torch.no_grad():
    for _ in range(5):
        G(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        G(x)
    e1.record()
    torch.cuda.synchronize()
print(f"G forward {e0.elapsed_time(e1) / iters:.4f} ms")
