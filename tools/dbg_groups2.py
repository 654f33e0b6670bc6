import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, torch
import gan_inpainting_amd
from gan_inpainting_amd.lib.models import networks
from oracle import params as op
hw, n, dtype = 256, 2, "fp32"
P = op.make_patchgan_params(77, H=hw, W=hw)
def mk():
    D = networks.PatchGANDiscriminator(sigmoid=False, image_size=hw, dtype=dtype)
    D.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in P.items()})
    return D.cuda().train()
a, _ = op.synth_batch(501, n, hw, hw); b, _ = op.synth_batch(502, n, hw, hw)
a, b = torch.from_numpy(a).cuda(), torch.from_numpy(b * 0.5 + 0.2).cuda()
da = torch.full((n, 1), 1.0 / n, device="cuda"); db = torch.zeros((n, 1), device="cuda")
D1 = mk(); D1.zero_grad()
ya, sa, ga = D1._forward_raw(a); D1._backward_raw(sa, ga, da, False, True)
D2 = mk(); D2.zero_grad()
y2, s2, g2 = D2._forward_raw(torch.cat([a, b]), 2)
D2._backward_raw(s2, g2, torch.cat([da, db]), False, True)
D6 = mk(); D6.zero_grad()           # same stacked input but ONE BatchNorm group over n (plain 2n batch) for contrast
torch.cuda.synchronize()
g1 = dict(D1.named_parameters())["model.9.bias"].grad.cpu().double()
g2_ = dict(D2.named_parameters())["model.9.bias"].grad.cpu().double()
d = (g1 - g2_).abs()
print("dbeta: |sep|", float(g1.norm()), "max abs diff", float(d.max()), "argmax", int(d.argmax()), "n(diff>1e-6)", int((d > 1e-6).sum()), "of", d.numel())
idx = torch.nonzero(d > 1e-6).flatten()[:20].tolist()
print("channels:", idx)
print("sep   :", [round(float(g1[i]), 6) for i in idx[:8]])
print("stack :", [round(float(g2_[i]), 6) for i in idx[:8]])
