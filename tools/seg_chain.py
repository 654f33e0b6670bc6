import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import gan_inpainting_amd  # noqa
from gan_inpainting_amd.lib.models import networks
x = torch.rand(8, 1, 512, 512, device="cuda")
a = networks.UnetGenerator(1, 4, 7, ngf=32, use_dropout="False", dtype="fp16").to("cuda").eval()
for _ in range(8):
    xx = x.clone().requires_grad_(True); a(xx).sum().backward()
torch.cuda.synchronize()
