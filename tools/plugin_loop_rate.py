#!/usr/bin/env python3
"""Rate of the plugin training loop (experiment_list._common.run_epochs) when the loader is free: batches are
pre-made tensors. Shows the host-side overhead the loop adds around the step (logging, statistics, transfers).
usage: plugin_loop_rate.py [cuda|pinned]   (where the pre-made batches live)"""
import os, sys, time, tempfile
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
import gan_inpainting_amd  # noqa
from gan_inpainting_amd.experiment_list import wgan_rmse

where = sys.argv[1] if len(sys.argv) > 1 else "cuda"
n, hw, nb = 32, 256, 100
torch.manual_seed(0)
g = torch.rand(n, 1, hw, hw)
m = torch.zeros(n, 1, hw, hw); m[:, :, 64:160, 64:160] = 1
seg = torch.zeros(n, 1, dtype=torch.long)
if where == "cuda":
    g, m = g.cuda(), m.cuda()
else:
    g, m = g.pin_memory(), m.pin_memory()


class Fixed:
    def __init__(self, k): self.k = k
    def __len__(self): return self.k
    def __iter__(self):
        for _ in range(self.k):
            yield g, m, seg


out = tempfile.mkdtemp()
state = dict(experiments=["wgan_rmse"], numepoch=int(os.environ.get("EPOCHS", "3")), batchsize=n, generator="unet", discriminator="patchgan", imagedim=hw, saveevery=100,
             updatediscevery=3, evalevery=100, debug="false", dtype="fp16", gp_lambda=0.0, g_every=5, outdir=out,
             train_fid=None, test_fid=None, inception_model=None, segmentation_model=None)
loaders = {"train": Fixed(nb), "test": Fixed(2), "extra": Fixed(2)}
t0 = time.perf_counter()
wgan_rmse.begin(state, loaders)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
ne = state["numepoch"]
print(f"{where}: {ne} epochs x {nb} batches in {dt:.2f} s  ->  {1e3 * dt / (ne * nb):.2f} ms/batch incl. set-up ({ne * nb * n / dt:.0f} images/s)")
