#!/usr/bin/env python3
"""One WGAN batch with a generator update on ONE stream (256x256, bs=32, fp16), N times: run under
`rocprofv3 --kernel-trace` and feed the trace to tools/chain_table.py to see every kernel of the batch in order."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
import gan_inpainting_amd  # noqa
from gan_inpainting_amd import optim, trainer
from gan_inpainting_amd.lib.models import networks

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 6
upd = (sys.argv[2] != "critic") if len(sys.argv) > 2 else True     # "critic": a critic-only batch
dev = torch.device("cuda")
n, hw = 32, 256
torch.manual_seed(1)
G = networks.get_network("generator", "unet", dtype="fp16").to(dev)
D = networks.PatchGANDiscriminator(sigmoid=False, image_size=hw, dtype="fp16").to(dev)
oG, oD = optim.RMSprop(G.parameters(), lr=5e-5), optim.RMSprop(D.parameters(), lr=5e-5)
step = trainer.WGANStep(G, D, oG, oD, recon="rmse", overlap=False)
ground = torch.rand(n, 1, hw, hw, device=dev)
mask = torch.zeros(n, 1, hw, hw, device=dev); mask[:, :, 64:160, 64:160] = 1
for _ in range(iters):
    step(ground, mask, True)
if not upd:
    step(ground, mask, False)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
step(ground, mask, upd)
e1.record()
torch.cuda.synchronize()
print(f"batch {'with generator update' if upd else 'critic only'}, one stream: {e0.elapsed_time(e1):.3f} ms")
