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
for wa, wb in ((1.0, 0.0), (0.0, 1.0)):
    da = torch.full((n, 1), wa / n, device="cuda"); db = torch.full((n, 1), wb / n, device="cuda")
    D1 = mk(); D1.zero_grad()
    ya, sa, ga = D1._forward_raw(a); yb, sb, gb = D1._forward_raw(b)
    D1._backward_raw(sa, ga, da, False, True); D1._backward_raw(sb, gb, db, False, True)
    D2 = mk(); D2.zero_grad()
    y2, s2, g2 = D2._forward_raw(torch.cat([a, b]), 2)
    D2._backward_raw(s2, g2, torch.cat([da, db]), False, True)
    torch.cuda.synchronize()
    print("weights", wa, wb)
    for (n1, p1), (n2, p2) in zip(D1.named_parameters(), D2.named_parameters()):
        g1, g2_ = p1.grad.double(), p2.grad.double()
        print(f"  {n1:20s} |g| {float(g1.norm()):.3e} stacked-vs-sep {float((g1-g2_).norm()/(g1.norm()+1e-30)):.2e}")
print("---- separate path: (fwd a, fwd b, bwd a, bwd b) vs (fwd a, bwd a, fwd b, bwd b), unit weights")
da = torch.full((n, 1), 1.0 / n, device="cuda"); db = torch.full((n, 1), 1.0 / n, device="cuda")
D1 = mk(); D1.zero_grad()
ya, sa, ga = D1._forward_raw(a); yb, sb, gb = D1._forward_raw(b)
D1._backward_raw(sa, ga, da, False, True); D1._backward_raw(sb, gb, db, False, True)
D3 = mk(); D3.zero_grad()
ya, sa, ga = D3._forward_raw(a); D3._backward_raw(sa, ga, da, False, True)
yb, sb, gb = D3._forward_raw(b); D3._backward_raw(sb, gb, db, False, True)
D2 = mk(); D2.zero_grad()
y2, s2, g2 = D2._forward_raw(torch.cat([a, b]), 2)
D2._backward_raw(s2, g2, torch.cat([da, db]), False, True)
torch.cuda.synchronize()
for (n1, p1), (n3, p3), (n2, p2) in zip(D1.named_parameters(), D3.named_parameters(), D2.named_parameters()):
    g1, g3, g2_ = p1.grad.double(), p3.grad.double(), p2.grad.double()
    print(f"  {n1:20s} ffbb-vs-fbfb {float((g1-g3).norm()/(g1.norm()+1e-30)):.2e}   stacked-vs-fbfb {float((g2_-g3).norm()/(g3.norm()+1e-30)):.2e}")
print("---- running statistics after the forwards (D1 separate vs D2 stacked)")
for (k1, v1), (k2, v2) in zip(D1.state_dict().items(), D2.state_dict().items()):
    if "running" in k1:
        d = (v1.double() - v2.double()).abs().max().item() / (v1.double().abs().max().item() + 1e-30)
        print(f"  {k1:28s} rel max diff {d:.2e}")
# group-0-only forward of the stacked net against a plain forward of `a` alone, different companions
D4 = mk(); y4, _, _ = D4._forward_raw(torch.cat([a, a]), 2)
D5 = mk(); y5, _, _ = D5._forward_raw(a)
print("  D(a) alone vs first half of stacked [a|a]:", float((y4[:n] - y5).abs().max()), " second half:", float((y4[n:] - y5).abs().max()), " |y|", float(y5.abs().max()))
