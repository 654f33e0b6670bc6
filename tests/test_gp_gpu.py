"""GPU tier: WGAN-GP gradient penalty (an EXTENSION: the reference clips weights and has no penalty;
parity is against the oracle's torch.autograd double backward, unpinned by the reference itself)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import gan_inpainting_amd  # noqa: F401,E402
from gan_inpainting_amd.lib.models import networks  # noqa: E402
from oracle import params as op  # noqa: E402
from oracle import torch_ref as orc  # noqa: E402
from gpu_util import close_to_either, rel_l2  # noqa: E402


@pytest.mark.parametrize("cfg", [(64, 4), (128, 3)])
def test_gradient_penalty_vs_oracle_double_backward(cfg):
    HW, N = cfg
    seed = 300 + HW
    P = op.make_patchgan_params(seed, HW, HW)
    D = networks.PatchGANDiscriminator(sigmoid=False, image_size=HW, dtype="fp32")
    D.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in P.items()})
    D = D.to("cuda").train()
    real, _ = op.synth_batch(seed + 1, N, HW, HW)
    fake, _ = op.synth_batch(seed + 2, N, HW, HW)
    eps = np.random.Generator(np.random.PCG64(seed + 3)).random(N).astype(np.float32)
    D.zero_grad()
    pen = D.gradient_penalty(torch.from_numpy(real).cuda(), torch.from_numpy(fake).cuda(), torch.from_numpy(eps), lam=10.0)
    torch.cuda.synchronize()
    res = {}
    for dt in (torch.float32, torch.float64):
        OP = orc.to_torch(P, dtype=dt)
        gp = orc.gradient_penalty(OP, torch.from_numpy(real).to(dt), torch.from_numpy(fake).to(dt),
                                  torch.from_numpy(eps).to(dt).view(-1, 1, 1, 1), lam=10.0)
        gp.backward()
        res[dt] = (OP, float(gp))
    print("penalty hip", float(pen), "oracle32", res[torch.float32][1], "oracle64", res[torch.float64][1])
    assert abs(float(pen) - res[torch.float64][1]) <= 2e-4 * abs(res[torch.float64][1]) + 1e-6
    bad = []
    for name, p in D.named_parameters():
        g32, g64 = res[torch.float32][0][name].grad, res[torch.float64][0][name].grad
        if g64 is None:   # Linear bias: the penalty does not depend on it
            assert float(p.grad.abs().max()) == 0.0, name
            continue
        ok, msg = close_to_either(f"gp{cfg} grad {name}", p.grad.detach().cpu(), g32, g64, 2e-3)
        if not ok and rel_l2(p.grad.detach().cpu(), g64) > 2e-3:
            bad.append(msg)
    assert not bad, "\\n".join(bad)
    # running statistics advanced like the oracle's (the penalty's forward is a train-mode forward)
    for k, v in D.state_dict().items():
        if k.endswith("running_mean"):
            assert rel_l2(v.cpu(), res[torch.float32][0][k]) < 1e-4, k


def test_wgan_gp_step_runs():
    from gan_inpainting_amd import optim, trainer
    G = networks.UnetGenerator(1, 1, 6, ngf=64, use_dropout="False", dtype="fp32").to("cuda")
    D = networks.PatchGANDiscriminator(sigmoid=False, image_size=64, dtype="fp32").to("cuda")
    step = trainer.WGANStep(G, D, optim.RMSprop(G.parameters(), lr=5e-5), optim.RMSprop(D.parameters(), lr=5e-5), gp_lambda=10.0)
    g, m = op.synth_batch(1, 4, 64, 64)
    w0 = D.flat_params().clone()
    L = step(torch.from_numpy(g).cuda(), torch.from_numpy(m).cuda(), True)
    assert torch.isfinite(L["gp"]).all() and float(L["gp"]) > 0
    assert float((D.flat_params() - w0).abs().max()) > 0 and float(D.flat_params().abs().max()) > 0.011   # not clipped


@pytest.mark.parametrize("n,overlap", [(4, False), (3, False), (3, True)])
def test_stacked_wgan_gp_critic_gradients_vs_oracle(n, overlap):
    """The default WGAN-GP configuration (BASELINE configs[1]): the critic's D(ground) | D(inpainted) pass runs
    stacked with two BatchNorm populations, the penalty on the n interpolates with ONE. The critic's parameter
    gradients of a whole batch, -(mean D(real) - mean D(fake)) + lam * penalty as wgan_l1.py:137-141 accumulates them,
    against the oracle (torch autograd, double backward for the penalty). Odd n: the interpolate batch does not
    split into two groups at all."""
    from gan_inpainting_amd import optim, trainer
    HW, nd, seed = 64, 6, 700 + n
    PG, PD = op.make_unet_params(seed, num_downs=nd), op.make_patchgan_params(seed + 1, HW, HW)
    sd = lambda P: {k: torch.from_numpy(np.array(v)) for k, v in P.items()}   # noqa: E731
    G = networks.UnetGenerator(1, 1, nd, ngf=64, use_dropout="False", dtype="fp32")
    G.load_state_dict(sd(PG))
    D = networks.PatchGANDiscriminator(sigmoid=False, image_size=HW, dtype="fp32")
    D.load_state_dict(sd(PD))
    G, D = G.cuda(), D.cuda()
    step = trainer.WGANStep(G, D, optim.RMSprop(G.parameters(), lr=5e-5), optim.RMSprop(D.parameters(), lr=5e-5), gp_lambda=10.0,
                            overlap=overlap, stacked=True)
    g, m = op.synth_batch(seed + 2, n, HW, HW)
    eps = np.random.Generator(np.random.PCG64(seed + 3)).random(n).astype(np.float32)
    step.gp_eps = torch.from_numpy(eps).cuda()
    L = step(torch.from_numpy(g).cuda(), torch.from_numpy(m).cuda(), False)
    torch.cuda.synchronize()
    masks = {k: v.cpu() for k, v in G.dropout_masks().items()}
    res = {}
    for dt in (torch.float32, torch.float64):
        OG, OD = orc.to_torch(PG, dtype=dt), orc.to_torch(PD, dtype=dt)
        ground, mask = torch.from_numpy(g).to(dt), torch.from_numpy(m).to(dt)
        mc, masked = orc.mask_pipeline(ground, mask)
        with torch.no_grad():
            inp = orc.composite(masked, orc.unet_forward(OG, masked, nd, True, masks), mc)
        orc.patchgan_forward(OD, ground, False, True).mean().backward()
        (-orc.patchgan_forward(OD, inp, False, True).mean()).backward()
        gp = orc.gradient_penalty(OD, ground, inp, torch.from_numpy(eps).to(dt).view(-1, 1, 1, 1), lam=10.0)
        gp.backward()
        res[dt] = (OD, float(gp))
    assert abs(float(L["gp"]) - res[torch.float64][1]) <= 1e-3 * abs(res[torch.float64][1]) + 1e-6
    bad = []
    for name, p in D.named_parameters():
        g32, g64 = res[torch.float32][0][name].grad, res[torch.float64][0][name].grad
        if float(g64.abs().max()) < 1e-9:     # Linear bias: d/db (mean D(real) - mean D(fake)) = 1 - 1, the penalty does not see it
            assert float(p.grad.abs().max()) <= 1e-6, name
            continue
        ok, msg = close_to_either(f"stacked gp n={n} grad {name}", p.grad.detach().cpu(), g32, g64, 2e-3)
        if not ok and rel_l2(p.grad.detach().cpu(), g64) > 2e-3:
            bad.append(msg)
    assert not bad, "\n".join(bad)
