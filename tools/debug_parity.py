"""Diagnostic (GPU box): per-parameter gradient error of the HIP path vs fp32/fp64 oracles."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import gan_inpainting_amd
from gan_inpainting_amd import optim, trainer
from gan_inpainting_amd.lib.models import networks
from oracle import params as op, torch_ref as orc
from util_golden import load, unpack_masks

def sd(P): return {k: torch.from_numpy(np.array(v)) for k, v in P.items()}
def l2(a, b): return float((a.double()-b.double()).norm()/(b.double().norm()+1e-30))
def mx(a, b): return float((a.double()-b.double()).abs().max()/(b.double().abs().max()+1e-30))

def unet_case(nd, N, HW, dtype, scale=1.0):
    seed = 100+nd+HW
    P = op.make_unet_params(seed, num_downs=nd)
    net = networks.UnetGenerator(1,1,nd,ngf=64,use_dropout="False",dtype=dtype); net.load_state_dict(sd(P)); net.set_loss_scale(scale); net=net.to("cuda").train()
    ground, mask = op.synth_batch(seed+7, N, HW, HW); x = torch.from_numpy(ground*(1-mask))
    R = torch.from_numpy(np.random.Generator(np.random.PCG64(seed)).standard_normal(size=(N,1,HW,HW),dtype=np.float32))
    y = net(x.cuda()); (y*R.cuda()).sum().backward(); torch.cuda.synchronize()
    masks = {k:v.cpu() for k,v in net.dropout_masks().items()}
    O32, O64 = orc.to_torch(P), orc.to_torch(P, dtype=torch.float64)
    y32 = orc.unet_forward(O32, x, nd, True, masks); (y32*R).sum().backward()
    y64 = orc.unet_forward(O64, x.double(), nd, True, masks); (y64*R.double()).sum().backward()
    print(f"== unet nd={nd} N={N} HW={HW} {dtype} scale={scale}: out max {mx(y.detach().cpu(), y64.detach()):.2e} l2 {l2(y.detach().cpu(), y64.detach()):.2e} | oracle32 out l2 {l2(y32.detach(), y64.detach()):.2e}")
    for name, p in net.named_parameters():
        g = p.grad.detach().cpu()
        print(f"   {name:62s} hip-vs-64 l2 {l2(g, O64[name].grad):.2e} max {mx(g, O64[name].grad):.2e} | o32-vs-64 l2 {l2(O32[name].grad, O64[name].grad):.2e} max {mx(O32[name].grad, O64[name].grad):.2e}")

def minimax_case():
    fx = load("minimax_steps"); seed, N = int(fx["seed"]), int(fx["N"])
    G = networks.get_network("generator","unet",dtype="fp32"); G.load_state_dict(sd(op.make_unet_params(seed))); G=G.to("cuda")
    D = networks.PatchGANDiscriminator(sigmoid=True,dtype="fp32"); D.load_state_dict(sd(op.make_patchgan_params(seed+1))); D=D.to("cuda")
    oG = optim.Adam(G.parameters(), lr=0.0002, betas=(0.5,0.999)); oD = optim.Adam(D.parameters(), lr=0.0002, betas=(0.5,0.999))
    step = trainer.MinimaxStep(G, D, oG, oD)
    res = {}
    for dt in (torch.float32, torch.float64):
        PG, PD = orc.to_torch(op.make_unet_params(seed), dtype=dt), orc.to_torch(op.make_patchgan_params(seed+1), dtype=dt)
        res[dt] = (PG, PD, orc.Adam(orc.trainable(PG)), orc.Adam(orc.trainable(PD)))
    for it in range(2):
        g, m = op.synth_batch(seed*100+it, N, 128, 128, fractional_edge=(it==0))
        masks = unpack_masks(fx, f"it{it}_")
        G.impose_dropout_masks(masks)
        L = step(torch.from_numpy(g).cuda(), torch.from_numpy(m).cuda())
        out = {}
        for dt in (torch.float32, torch.float64):
            PG, PD, aG, aD = res[dt]
            out[dt] = orc.minimax_step(PG, PD, aG, aD, torch.from_numpy(g).to(dt), torch.from_numpy(m).to(dt), 7, masks)
        print(f"== minimax it{it}: " + " ".join(f"{k}: hip {L[k].item():.6f} o32 {out[torch.float32][k]:.6f} o64 {out[torch.float64][k]:.6f} ref {float(fx[f'it{it}_{k}']):.6f} |" for k in ("d_loss_fake","g_adv","recon")))
        P32, P64 = res[torch.float32][0], res[torch.float64][0]
        worst = []
        for name, p in G.named_parameters():
            gg = p.grad.detach().cpu()
            worst.append((mx(gg, P64[name].grad), mx(P32[name].grad, P64[name].grad), l2(p.detach().cpu(), P64[name].detach()), l2(P32[name].detach(), P64[name].detach()), name))
        for w in worst: print(f"   {w[4]:62s} grad max hip {w[0]:.2e} o32 {w[1]:.2e} | param-after-step l2 hip {w[2]:.2e} o32 {w[3]:.2e}")

if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("all", "minimax"): minimax_case()
    if which in ("all", "unet"):
        unet_case(7, 2, 128, "fp32")
        unet_case(7, 2, 128, "fp16")
        unet_case(7, 8, 128, "fp16")
        unet_case(7, 8, 128, "fp16", scale=64.0)
