import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gan_inpainting_amd
from gan_inpainting_amd import optim, trainer
from gan_inpainting_amd.lib.models import networks
dev = torch.device("cuda")
for hw, nd in ((192, 6), (384, 7)):
    n = 4
    torch.manual_seed(1)
    G = networks.UnetGenerator(1, 1, nd, ngf=64, dtype="fp16").to(dev)
    D = networks.PatchGANDiscriminator(sigmoid=False, image_size=hw, dtype="fp16").to(dev)
    oG, oD = optim.RMSprop(G.parameters(), lr=5e-5), optim.RMSprop(D.parameters(), lr=5e-5)
    for overlap in (False, True):
        step = trainer.WGANStep(G, D, oG, oD, recon="rmse", overlap=overlap)
        ground = torch.rand(n, 1, hw, hw, device=dev)
        mask = torch.zeros(n, 1, hw, hw, device=dev); mask[:, :, hw // 4:hw // 2, hw // 4:hw // 2] = 1
        for it in range(6):
            L = step(ground, mask, it % 5 == 4)
        step.sync_for_logging(); torch.cuda.synchronize()
        print(hw, overlap, {k: float(v) for k, v in L.items()})
