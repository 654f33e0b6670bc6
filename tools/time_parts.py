#!/usr/bin/env python3
"""Time the parts of a WGAN batch in isolation (256x256, bs=32, fp16): generator forward, stacked critic update,
critic forward + input gradient, generator backward + RMSprop. Tells what the two-stream schedule can hide."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
import gan_inpainting_amd  # noqa
from gan_inpainting_amd import optim, trainer
from gan_inpainting_amd.lib.models import networks

dev = torch.device("cuda")
n, hw = 32, 256
torch.manual_seed(1)
G = networks.get_network("generator", "unet", dtype="fp16").to(dev)
D = networks.PatchGANDiscriminator(sigmoid=False, image_size=hw, dtype="fp16").to(dev)
oG, oD = optim.RMSprop(G.parameters(), lr=5e-5), optim.RMSprop(D.parameters(), lr=5e-5)
step = trainer.WGANStep(G, D, oG, oD, recon="rmse", overlap=False)
ground = torch.rand(n, 1, hw, hw, device=dev)
mask = torch.zeros(n, 1, hw, hw, device=dev); mask[:, :, 64:160, 64:160] = 1
step(ground, mask, True)
o = step.ops


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def g_fwd():
    o.mask_apply(ground, mask, step.mask_c, step.masked, True)
    step._gen, step._gtok = step._fwd(G, step.masked)
    o.composite(step.masked, step._gen, step.mask_c, step.inpainted)


def d_update():
    oD.zero_grad()
    step._critic_stacked(ground, step.inpainted)
    oD.step()


def d_adv():
    p, t = step._fwd(D, step.inpainted)
    o.adv(p, trainer.MEAN, 0.0, step._loss("g_adv"), step.dpred, +1.0)
    step._dadv = step._bwd(D, t, step.dpred, True, False)


def g_bwd():
    g_fwd()
    oG.zero_grad()
    step._bwd_G(step._gtok, step.g_gen)
    oG.step()


with torch.no_grad():
    t_gf = timeit(g_fwd)
    t_du = timeit(d_update)
    t_da = timeit(d_adv)
    t_gb = timeit(g_bwd) - t_gf
    print(f"G forward + mask/composite   {t_gf:.3f} ms")
    print(f"critic update (stacked)      {t_du:.3f} ms")
    print(f"critic fwd + input gradient  {t_da:.3f} ms")
    print(f"G backward + RMSprop         {t_gb:.3f} ms")
    print(f"serial critic-only batch {t_gf + t_du:.3f} ms, batch with G update {t_gf + t_du + t_da + t_gb:.3f} ms, "
          f"5-batch average {(5 * (t_gf + t_du) + t_da + t_gb) / 5:.3f} ms")
