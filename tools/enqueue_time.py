#!/usr/bin/env python3
"""Host enqueue time vs GPU time of the headline step (is the step launch-bound?)."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
import gan_inpainting_amd  # noqa
from gan_inpainting_amd import optim, trainer
from gan_inpainting_amd.lib.models import networks
dev = torch.device("cuda")
torch.manual_seed(1)
G = networks.get_network("generator", "unet", dtype="fp16").to(dev)
D = networks.PatchGANDiscriminator(sigmoid=False, image_size=256, dtype="fp16").to(dev)
oG, oD = optim.RMSprop(G.parameters(), lr=5e-5), optim.RMSprop(D.parameters(), lr=5e-5)
step = trainer.WGANStep(G, D, oG, oD, recon="rmse", overlap=True)
step.inputs_resident = True
g = torch.rand(32, 1, 256, 256, device=dev); m = torch.zeros_like(g); m[:, :, 64:160, 64:160] = 1
for i in range(10):
    step(g, m, i % 5 == 4)
torch.cuda.synchronize()
N = 40
t0 = time.perf_counter()
for i in range(N):
    step(g, m, i % 5 == 4)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3 * (t1 - t0) / N:.3f} ms/step, total {1e3 * (t2 - t0) / N:.3f} ms/step, GPU still busy after enqueue for {1e3 * (t2 - t1):.1f} ms")
