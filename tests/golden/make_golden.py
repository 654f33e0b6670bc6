#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the UNMODIFIED reference (run in the build container only).

    python tests/golden/make_golden.py [--ref /root/reference]

What it does
  * registers an empty `torchvision` stub (the reference imports it at networks.py:3 but only
    uses it inside VGG19Wrapper.__init__, networks.py:371) and imports the reference's
    lib/models/networks.py and lib/models/util.py as they lie under --ref;
  * loads build-generated deterministic weights (oracle/params.py) with load_state_dict;
  * runs forward / backward / the per-batch step sequences (re-typed from
    experiment_list/minimaxgan_l1.py:110-173, wgan_l1.py:110-186,
    experiment1_global_local_D.py:139-200, with torch.optim / torch.nn loss built-ins exactly as
    those plugins call them; one=+1, mone=-1);
  * captures the Dropout keep-masks with forward hooks;
  * asserts that the repo's own oracle (oracle/torch_ref.py) reproduces every recorded number
    (this is the pin), and writes small fixtures (data only: seeds, outputs, losses, statistics).

Nothing from the reference is copied: the fixtures hold inputs' seeds and numeric outputs.
"""
import argparse
import itertools
import os
import sys
import types
from collections import OrderedDict

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import params as op  # noqa: E402
from oracle import torch_ref as orc  # noqa: E402


def import_reference(ref_root):
    tv = types.ModuleType("torchvision")
    tv.models = types.ModuleType("torchvision.models")
    sys.modules.setdefault("torchvision", tv)
    sys.modules.setdefault("torchvision.models", tv.models)
    sys.path.insert(0, ref_root)
    import lib.models.networks as networks  # noqa
    import lib.models.util as util  # noqa
    sys.path.pop(0)
    return networks, util


def load(module, P):
    sd = OrderedDict((k, torch.from_numpy(np.array(v))) for k, v in P.items())
    module.load_state_dict(sd, strict=True)
    return module


class MaskCapture:
    """Forward hooks on the reference's nn.Dropout modules: keep-mask = (output != 0)."""

    def __init__(self, net_G, num_downs):
        self.masks = {}
        self.levels = {}
        drops = [m for m in net_G.modules() if isinstance(m, torch.nn.Dropout)]
        # modules() is a pre-order walk and each block's Dropout follows its submodule, so the
        # first Dropout met belongs to the deepest dropout level (num_downs-1)
        for lvl, m in zip(reversed(op.dropout_levels(num_downs)), drops):
            m.register_forward_hook(self._hook(lvl))

    def _hook(self, lvl):
        def fn(mod, inp, out):
            if mod.training:
                self.masks[lvl] = (out != 0).to(torch.uint8).clone()
        return fn


def pack_masks(masks):
    out = {}
    for lvl, m in masks.items():
        a = m.numpy().astype(np.uint8)
        out[f"mask{lvl}_shape"] = np.array(a.shape, dtype=np.int64)
        out[f"mask{lvl}_bits"] = np.packbits(a.reshape(-1))
    return out


def grad_stats(module):
    names, absmean, head = [], [], []
    for n, p in module.named_parameters():
        names.append(n)
        absmean.append(float(p.grad.abs().mean()) if p.grad is not None else float("nan"))
        head.append(p.grad.reshape(-1)[:32].numpy().copy() if p.grad is not None else np.zeros(0, np.float32))
    return names, np.array(absmean, np.float64), head


def param_stats(module):
    return np.array([[float(p.double().sum()), float(p.double().abs().sum())] for _, p in module.named_parameters()])


def bn_stats(module):
    out = OrderedDict()
    for k, v in module.state_dict().items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            out[k] = v.numpy().copy()
    return out


def close(a, b, tol, what):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    err = np.max(np.abs(a - b) / (np.abs(b) + 1e-6)) if a.size else 0.0
    assert err <= tol, f"oracle != reference for {what}: rel err {err:.3e}"
    return err


def case_unet(networks, name, seed, N, HW, num_downs, train):
    torch.manual_seed(1000 + seed)
    P = op.make_unet_params(seed, num_downs=num_downs, ngf=64)
    if num_downs == 7:
        net = networks.get_network("generator", "unet")  # networks.py:17-19
    else:
        net = networks.UnetGenerator(1, 1, num_downs, ngf=64,
                                     norm_layer=networks.get_norm_layer(norm_type="batch"),
                                     use_dropout="False")
    load(net, P)
    net.train(train)
    cap = MaskCapture(net, num_downs)
    ground, mask = op.synth_batch(seed + 7, N, HW, HW)
    x = torch.from_numpy(ground * (1 - mask))
    rng = np.random.Generator(np.random.PCG64(seed + 99))
    R = torch.from_numpy(rng.standard_normal(size=(N, 1, HW, HW), dtype=np.float32))
    out = net(x)
    fx = dict(seed=seed, N=N, HW=HW, num_downs=num_downs, train=int(train), out=out.detach().numpy())
    # --- oracle pin
    OP = orc.to_torch(P)
    masks = {k: v for k, v in cap.masks.items()}
    oout = orc.unet_forward(OP, x, num_downs, train, masks)
    e = close(oout.detach().numpy(), fx["out"], 1e-6, name + ".out")
    if train:
        (out * R).sum().backward()
        names, absmean, head = grad_stats(net)
        (oout * R).sum().backward()
        for i, n in enumerate(names):
            close(float(OP[n].grad.abs().mean()), absmean[i], 1e-5, f"{name}.grad[{n}]")
        fx.update(grad_names=np.array(names), grad_absmean=absmean,
                  **{f"ghead_{i}": h for i, h in enumerate(head)})
        for k, v in bn_stats(net).items():
            close(OP[k].numpy(), v, 1e-6, f"{name}.{k}")
            fx["bn::" + k] = v
        fx.update(pack_masks(cap.masks))
    print(f"  {name}: out err {e:.2e}")
    return fx


def case_unet_norm(networks, name, seed, N, HW, num_downs, norm):
    """UnetGenerator built with get_norm_layer('instance') / ('none') (networks.py:30-45; never chosen by get_network, but
    part of the class's surface): InstanceNorm2d without affine parameters or running statistics, every convolution with
    a bias (:279-286), resp. Identity norm layers. Train-mode forward + backward and an eval-mode forward (InstanceNorm
    uses instance statistics in both) on the reference class; the oracle is asserted equal."""
    torch.manual_seed(2000 + seed)
    P = op.make_unet_params(seed, num_downs=num_downs, ngf=64, norm=norm)
    net = networks.UnetGenerator(1, 1, num_downs, ngf=64, norm_layer=networks.get_norm_layer(norm_type=norm), use_dropout="False")
    assert [k for k in net.state_dict().keys()] == [k for k in P.keys()], "state_dict layout"
    load(net, P)
    net.train(True)
    cap = MaskCapture(net, num_downs)
    ground, mask = op.synth_batch(seed + 7, N, HW, HW)
    x = torch.from_numpy(ground * (1 - mask))
    rng = np.random.Generator(np.random.PCG64(seed + 99))
    R = torch.from_numpy(rng.standard_normal(size=(N, 1, HW, HW), dtype=np.float32))
    xr = x.clone().requires_grad_(True)
    out = net(xr)
    (out * R).sum().backward()
    names, absmean, head = grad_stats(net)
    fx = dict(seed=seed, N=N, HW=HW, num_downs=num_downs, norm=np.array(norm), out=out.detach().numpy(), dx=xr.grad.numpy().copy(),
              grad_names=np.array(names), grad_absmean=absmean, **{f"ghead_{i}": h for i, h in enumerate(head)})
    fx.update(pack_masks(cap.masks))
    OP = orc.to_torch(P)
    xo = x.clone().requires_grad_(True)
    oout = orc.unet_forward(OP, xo, num_downs, True, dict(cap.masks), norm=norm)
    e = close(oout.detach().numpy(), fx["out"], 1e-6, name + ".out")
    (oout * R).sum().backward()
    for i, n in enumerate(names):
        close(float(OP[n].grad.abs().mean()), absmean[i], 1e-5, f"{name}.grad[{n}]")
    close(xo.grad.numpy(), fx["dx"], 1e-5, name + ".dx")
    net.eval()
    with torch.no_grad():
        ev = net(x)
        oev = orc.unet_forward(OP, x, num_downs, False, None, norm=norm)
    close(oev.numpy(), ev.numpy(), 1e-6, name + ".eval_out")
    fx["eval_out"] = ev.numpy()
    print(f"  {name}: out err {e:.2e}")
    return fx


def case_patchgan(networks, name, seed, N):
    P = op.make_patchgan_params(seed, 128, 128)
    fx = dict(seed=seed, N=N)
    ground, _ = op.synth_batch(seed + 3, N, 128, 128)
    rng = np.random.Generator(np.random.PCG64(seed + 5))
    r = torch.from_numpy(rng.standard_normal(size=(N, 1), dtype=np.float32))
    for sig in (True, False):
        net = load(networks.PatchGANDiscriminator(sigmoid=sig), P)  # networks.py:331, wgan_l1.py:58
        net.train(True)
        x = torch.from_numpy(ground).clone().requires_grad_(True)
        out = net(x)
        (out * r).sum().backward()
        names, absmean, head = grad_stats(net)
        tag = "sig" if sig else "lin"
        fx[f"out_{tag}"] = out.detach().numpy()
        fx[f"grad_absmean_{tag}"] = absmean
        fx[f"dx_absmean_{tag}"] = float(x.grad.abs().mean())
        fx[f"dx_head_{tag}"] = x.grad.reshape(-1)[:256].numpy().copy()
        fx["grad_names"] = np.array(names)
        OP = orc.to_torch(P)
        ox = torch.from_numpy(ground).clone().requires_grad_(True)
        oout = orc.patchgan_forward(OP, ox, sig, True)
        (oout * r).sum().backward()
        close(oout.detach().numpy(), fx[f"out_{tag}"], 1e-6, f"{name}.out_{tag}")
        close(ox.grad.numpy(), x.grad.numpy(), 1e-4, f"{name}.dx_{tag}")
        for i, n in enumerate(names):
            close(float(OP[n].grad.abs().mean()), absmean[i], 1e-5, f"{name}.grad_{tag}[{n}]")
        for k, v in bn_stats(net).items():
            close(OP[k].numpy(), v, 1e-6, f"{name}.{k}")
            fx[f"bn_{tag}::" + k] = v
        # eval-mode forward too
        net.train(False)
        fx[f"out_eval_{tag}"] = net(torch.from_numpy(ground)).detach().numpy()
        # (running stats were updated by the train-mode forward above, in both copies)
        close(orc.patchgan_forward(OP, torch.from_numpy(ground), sig, False).detach().numpy(),
              fx[f"out_eval_{tag}"], 1e-6, f"{name}.out_eval_{tag}")
    print(f"  {name}: ok")
    return fx


def _record_step(fx, i, rec):
    for k, v in rec.items():
        fx[f"it{i}_{k}"] = np.asarray(v)


def case_minimax(networks, util, name, seed, N, iters):
    """minimaxgan_l1.py:110-173 replayed on the reference modules."""
    PG, PD = op.make_unet_params(seed), op.make_patchgan_params(seed + 1)
    net_G = load(networks.get_network("generator", "unet"), PG)
    net_D = load(networks.get_network("discriminator", "patchgan"), PD)
    cap = MaskCapture(net_G, 7)
    bce, l1 = torch.nn.BCELoss(), torch.nn.L1Loss()
    G_opt = torch.optim.Adam(net_G.parameters(), lr=0.0002, betas=(0.5, 0.999))
    D_opt = torch.optim.Adam(net_D.parameters(), lr=0.0002, betas=(0.5, 0.999))
    OG, OD = orc.to_torch(PG), orc.to_torch(PD)
    oG, oD = orc.Adam(orc.trainable(OG)), orc.Adam(orc.trainable(OD))
    fx = dict(seed=seed, N=N, iters=iters)
    torch.manual_seed(seed)
    for it in range(iters):
        g_np, m_np = op.synth_batch(seed * 100 + it, N, 128, 128, fractional_edge=(it == 0))
        ground, mask = torch.from_numpy(g_np), torch.ceil(torch.from_numpy(m_np))
        masked = ground * (1 - mask)
        inpainted = net_G(masked)
        inpainted = masked + inpainted * mask
        util.set_requires_grad([net_D], True)
        D_opt.zero_grad()
        d_pred_real = net_D(ground).view(-1)
        d_loss_real = bce(d_pred_real, torch.ones(len(d_pred_real)))
        d_loss_real.backward()
        d_pred_fake = net_D(inpainted.detach()).view(-1)
        d_loss_fake = bce(d_pred_fake, torch.zeros(len(d_pred_fake)))
        d_loss_fake.backward()
        _, d_absmean, _ = grad_stats(net_D)
        D_opt.step()
        util.set_requires_grad([net_D], False)
        G_opt.zero_grad()
        d_pred_fake = net_D(inpainted).view(-1)
        g_adv = bce(d_pred_fake, torch.ones(len(d_pred_fake)))
        recon = l1(ground, inpainted)
        g_loss = g_adv + recon
        g_loss.backward()
        _, g_absmean, _ = grad_stats(net_G)
        G_opt.step()
        rec = dict(d_loss_real=float(d_loss_real), d_loss_fake=float(d_loss_fake), g_adv=float(g_adv),
                   recon=float(recon), g_loss=float(g_loss), g_grad_absmean=g_absmean,
                   d_grad_absmean=d_absmean, g_param_stats=param_stats(net_G),
                   d_param_stats=param_stats(net_D))
        _record_step(fx, it, rec)
        fx.update({f"it{it}_{k}": v for k, v in pack_masks(cap.masks).items()})
        if it == iters - 1:
            fx["final_inpainted"] = inpainted.detach().numpy()
        # --- oracle pin
        o = orc.minimax_step(OG, OD, oG, oD, torch.from_numpy(g_np), torch.from_numpy(m_np), 7, dict(cap.masks))
        for k in ("d_loss_real", "d_loss_fake", "g_adv", "recon", "g_loss"):
            close(o[k], rec[k], 2e-5, f"{name}.it{it}.{k}")
        gnames = [n for n, _ in net_G.named_parameters()]
        for i, n in enumerate(gnames):
            if "bias" not in n:
                close(o["g_grad_absmean"][n], g_absmean[i], 1e-4, f"{name}.it{it}.g_absmean[{n}]")
        ostats = np.array([[float(OG[n].double().sum()), float(OG[n].double().abs().sum())] for n in gnames])
        close(ostats[:, 1], rec["g_param_stats"][:, 1], 1e-6, f"{name}.it{it}.g_params")
        print(f"  {name} it{it}: g_loss {rec['g_loss']:.6f} d_real {rec['d_loss_real']:.6f} ok")
    fx["g_param_names"] = np.array([n for n, _ in net_G.named_parameters()])
    fx["d_param_names"] = np.array([n for n, _ in net_D.named_parameters()])
    return fx


def case_wgan(networks, util, name, seed, N, pattern):
    """wgan_l1.py:110-186 replayed on the reference modules; `pattern` forces the G-update
    decision per batch (the cadence rule itself is host logic, tested separately)."""
    PG, PD = op.make_unet_params(seed), op.make_patchgan_params(seed + 1)
    net_G = load(networks.get_network("generator", "unet"), PG)
    net_D = load(networks.PatchGANDiscriminator(sigmoid=False), PD)
    cap = MaskCapture(net_G, 7)
    l1 = torch.nn.L1Loss()
    G_opt = torch.optim.RMSprop(net_G.parameters(), lr=0.00005)
    D_opt = torch.optim.RMSprop(net_D.parameters(), lr=0.00005)
    one = torch.ones(1)
    mone = one * -1
    OG, OD = orc.to_torch(PG), orc.to_torch(PD)
    oG, oD = orc.RMSprop(orc.trainable(OG)), orc.RMSprop(orc.trainable(OD))
    fx = dict(seed=seed, N=N, pattern=np.array(pattern, np.int64))
    torch.manual_seed(seed)
    for it, upd in enumerate(pattern):
        g_np, m_np = op.synth_batch(seed * 100 + it, N, 128, 128)
        ground, mask = torch.from_numpy(g_np), torch.ceil(torch.from_numpy(m_np))
        masked = ground * (1 - mask)
        inpainted = net_G(masked)
        inpainted = masked + inpainted * mask
        util.set_requires_grad([net_D], True)
        D_opt.zero_grad()
        d_pred_real = net_D(ground)
        d_pred_fake = net_D(inpainted.detach())
        d_loss_real = torch.mean(d_pred_real).view(1)
        d_loss_real.backward(one)
        d_loss_fake = torch.mean(d_pred_fake).view(1)
        d_loss_fake.backward(mone)
        _, d_absmean, _ = grad_stats(net_D)
        D_opt.step()
        for p in net_D.parameters():
            p.data.clamp_(-0.01, 0.01)
        rec = dict(d_loss_real=float(d_loss_real), d_loss_fake=float(d_loss_fake), d_grad_absmean=d_absmean)
        if upd:
            util.set_requires_grad([net_D], False)
            G_opt.zero_grad()
            d_pred_fake = net_D(inpainted).view(-1)
            g_adv = torch.mean(d_pred_fake).view(1)
            recon = l1(ground, inpainted)
            g_loss = g_adv + recon
            g_loss.backward()
            _, g_absmean, _ = grad_stats(net_G)
            G_opt.step()
            rec.update(g_adv=float(g_adv), recon=float(recon), g_loss=float(g_loss), g_grad_absmean=g_absmean)
        rec.update(g_param_stats=param_stats(net_G), d_param_stats=param_stats(net_D))
        _record_step(fx, it, rec)
        fx.update({f"it{it}_{k}": v for k, v in pack_masks(cap.masks).items()})
        o = orc.wgan_step(OG, OD, oG, oD, torch.from_numpy(g_np), torch.from_numpy(m_np), 7, dict(cap.masks), bool(upd))
        for k in ("d_loss_real", "d_loss_fake") + (("g_adv", "recon", "g_loss") if upd else ()):
            close(o[k], rec[k], 5e-5, f"{name}.it{it}.{k}")
        dnames = [n for n, _ in net_D.named_parameters()]
        ostats = np.array([[float(OD[n].double().sum()), float(OD[n].double().abs().sum())] for n in dnames])
        close(ostats[:, 1], rec["d_param_stats"][:, 1], 1e-6, f"{name}.it{it}.d_params")
        print(f"  {name} it{it}: d_real {rec['d_loss_real']:.6f} d_fake {rec['d_loss_fake']:.6f} upd={upd} ok")
    fx["g_param_names"] = np.array([n for n, _ in net_G.named_parameters()])
    fx["d_param_names"] = np.array([n for n, _ in net_D.named_parameters()])
    return fx


def case_dual(networks, util, name, seed, N):
    """experiment1_global_local_D.py:139-200 (script-style file, loop body re-typed)."""
    PG, PDg, PDl = op.make_unet_params(seed), op.make_patchgan_params(seed + 1), op.make_patchgan_params(seed + 2)
    net_G = load(networks.get_network("generator", "unet"), PG)
    net_Dg = load(networks.get_network("discriminator", "patchgan"), PDg)
    net_Dl = load(networks.get_network("discriminator", "patchgan"), PDl)
    cap = MaskCapture(net_G, 7)
    mse = torch.nn.MSELoss()

    def rmse(a, b):  # loss.py:11-19 (the file itself cannot be imported: loss.py:4)
        return torch.sqrt(mse(a, b) + 1e-16)

    G_opt = torch.optim.Adam(net_G.parameters(), lr=0.0002, betas=(0.5, 0.999))
    D_opt = torch.optim.Adam(itertools.chain(net_Dl.parameters(), net_Dg.parameters()), lr=0.0002, betas=(0.5, 0.999))
    lambda1 = lambda2 = 300.0
    g_np, m_np = op.synth_batch(seed * 100, N, 128, 128)
    ground, mask = torch.from_numpy(g_np), torch.from_numpy(m_np)
    torch.manual_seed(seed)
    masked = ground * (1 - mask)
    util.set_requires_grad([net_Dg, net_Dl], False)
    G_opt.zero_grad()
    inpainted = net_G(masked)
    pg = net_Dg(inpainted).view(-1)
    pl = net_Dl(mask * inpainted).view(-1)
    g_adv_g = mse(pg, torch.ones(len(pg)))
    g_adv_l = mse(pl, torch.ones(len(pl)))
    rec_g = rmse(ground, inpainted)
    rec_l = rmse(mask * ground, mask * inpainted)
    g_loss = g_adv_g + g_adv_l + lambda1 * rec_g + lambda2 * rec_l
    g_loss.backward()
    _, g_absmean, _ = grad_stats(net_G)
    G_opt.step()
    util.set_requires_grad([net_Dg, net_Dl], True)
    D_opt.zero_grad()
    d_loss = (mse(net_Dg(ground).view(-1), torch.ones(N)) + mse(net_Dg(inpainted.detach()).view(-1), torch.zeros(N))
              + mse(net_Dl(ground * mask).view(-1), torch.ones(N))
              + mse(net_Dl(inpainted.detach() * mask).view(-1), torch.zeros(N)))
    d_loss.backward()
    D_opt.step()
    fx = dict(seed=seed, N=N, g_loss=float(g_loss), d_loss=float(d_loss), rmse_global=float(rec_g),
              rmse_local=float(rec_l), g_adv_global=float(g_adv_g), g_adv_local=float(g_adv_l),
              g_grad_absmean=g_absmean, g_param_stats=param_stats(net_G), dg_param_stats=param_stats(net_Dg),
              dl_param_stats=param_stats(net_Dl), **pack_masks(cap.masks))
    OG, ODg, ODl = orc.to_torch(PG), orc.to_torch(PDg), orc.to_torch(PDl)
    oG = orc.Adam(orc.trainable(OG))
    oD = orc.Adam(orc.trainable(ODl) + orc.trainable(ODg))
    o = orc.dual_d_step(OG, ODg, ODl, oG, oD, ground, mask, 7, dict(cap.masks))
    for k in ("g_loss", "d_loss", "rmse_global", "rmse_local"):
        close(o[k], fx[k], 5e-5, f"{name}.{k}")
    for tag, OP, net in (("dg", ODg, net_Dg), ("dl", ODl, net_Dl)):
        names = [n for n, _ in net.named_parameters()]
        ostats = np.array([[float(OP[n].double().sum()), float(OP[n].double().abs().sum())] for n in names])
        close(ostats[:, 1], fx[f"{tag}_param_stats"][:, 1], 1e-6, f"{name}.{tag}_params")
    print(f"  {name}: g_loss {fx['g_loss']:.5f} d_loss {fx['d_loss']:.5f} ok")
    return fx


def case_optim(name):
    """torch.optim.Adam / RMSprop vs the oracle's update rules on random tensors."""
    rng = np.random.Generator(np.random.PCG64(77))
    w0 = rng.standard_normal(4096).astype(np.float32)
    grads = [rng.standard_normal(4096).astype(np.float32) * 0.01 for _ in range(3)]
    fx = dict(w0=w0, **{f"g{i}": g for i, g in enumerate(grads)})
    for kind in ("adam", "rmsprop"):
        p = torch.nn.Parameter(torch.from_numpy(w0.copy()))
        q = torch.from_numpy(w0.copy()).requires_grad_(True)
        if kind == "adam":
            opt, oo = torch.optim.Adam([p], lr=0.0002, betas=(0.5, 0.999)), orc.Adam([q])
        else:
            opt, oo = torch.optim.RMSprop([p], lr=0.00005), orc.RMSprop([q])
        for g in grads:
            p.grad = torch.from_numpy(g.copy())
            q.grad = torch.from_numpy(g.copy())
            opt.step()
            oo.step()
        fx[f"w_{kind}"] = p.detach().numpy().copy()
        close(q.detach().numpy(), fx[f"w_{kind}"], 1e-6, f"{name}.{kind}")
    print(f"  {name}: ok")
    return fx


def case_init(networks, name, seed):
    """Default-initialised reference networks under torch.manual_seed(seed): per-tensor checksums.
    The backend's constructors must consume torch's global RNG in the same order."""
    torch.manual_seed(seed)
    g = networks.get_network("generator", "unet")
    d = networks.get_network("discriminator", "patchgan")
    fx = dict(seed=seed)
    for tag, net in (("g", g), ("d", d)):
        sdict = net.state_dict()
        keys = [k for k in sdict if not k.endswith("num_batches_tracked")]
        fx[f"{tag}_keys"] = np.array(keys)
        fx[f"{tag}_sum"] = np.array([float(sdict[k].double().sum()) for k in keys])
        fx[f"{tag}_abs"] = np.array([float(sdict[k].double().abs().sum()) for k in keys])
        fx[f"{tag}_head"] = np.stack([np.pad(sdict[k].reshape(-1)[:4].numpy().astype(np.float32), (0, max(0, 4 - sdict[k].numel())))
                                      for k in keys])
    print(f"  {name}: {len(fx['g_keys'])} + {len(fx['d_keys'])} tensors")
    return fx


SSIM_CASES = [  # (seed, n, c, h, w, window_size)
    (61, 3, 1, 40, 56, 11), (62, 2, 3, 33, 47, 11), (63, 2, 1, 128, 128, 11), (64, 2, 1, 24, 24, 7),
    (65, 1, 2, 9, 70, 11), (66, 2, 1, 64, 64, 3),
]


def case_ssim(ref_root, name):
    """lib/pytorch_ssim (pure torch) imported unmodified: ssim() and the SSIM module, both reductions."""
    sys.path.insert(0, ref_root)
    import lib.pytorch_ssim as rs  # noqa
    sys.path.pop(0)
    fx = {"cases": np.array(SSIM_CASES, dtype=np.int64)}
    for i, (seed, n, c, h, w, ws) in enumerate(SSIM_CASES):
        x, y = op.synth_ssim_pair(seed, n, c, h, w)
        tx, ty = torch.from_numpy(x), torch.from_numpy(y)
        m = rs.ssim(tx, ty, window_size=ws)
        per = rs.ssim(tx, ty, window_size=ws, size_average=False)
        m2 = rs.SSIM(window_size=ws)(tx, ty)
        assert torch.equal(m, m2)
        close(orc.ssim(tx, ty, ws), m, 0, "ssim mean %d" % i)
        close(orc.ssim(tx, ty, ws, size_average=False), per, 0, "ssim per-sample %d" % i)
        same = rs.ssim(tx, tx, window_size=ws)
        fx["mean_%d" % i] = m.numpy()
        fx["per_sample_%d" % i] = per.numpy()
        fx["self_%d" % i] = same.numpy()
        fx["mean64_%d" % i] = orc.ssim(tx.double(), ty.double(), ws).numpy()
    return fx


SEG_CASES = [  # (seed, n, num_classes, h, w, unique_labels, labels contain -1)
    (71, 3, 4, 32, 40, [0, 1, 2, 3], 0), (72, 2, 4, 64, 64, [0, 1, 2, 3], 0), (73, 2, 6, 24, 24, [1, 3, 5], 0),
    (74, 2, 4, 16, 24, [0, 1, 2, 3], 1),
]


def case_evalmetrics(ref_root, name):
    """lib/models/evaluate.py does not import (mixed tabs/spaces, undefined names), but
    calculate_segmentation_eval_metric (:179-224) is a self-contained torch function: its text is read
    from the reference AT GENERATION TIME, executed, and only its numeric outputs are stored. The
    reconstruction metrics of :127-158 are recorded from the oracle (composition of the pinned losses)."""
    src = open(os.path.join(ref_root, "lib/models/evaluate.py")).read()
    start = src.index("def calculate_segmentation_eval_metric")
    ns = {"torch": torch}
    exec(compile(src[start:], "reference:evaluate.py", "exec"), ns)
    ref_fn = ns["calculate_segmentation_eval_metric"]
    fx = {"seg_cases": np.array([c[:5] + (c[6],) for c in SEG_CASES], dtype=np.int64)}
    for i, (seed, n, K, h, w, uniq, neg) in enumerate(SEG_CASES):
        labels, logits = op.synth_segmentation(seed, n, K, h, w, with_minus_one=bool(neg))
        tl, to = torch.from_numpy(labels), torch.from_numpy(logits)
        m, a = ref_fn(tl, to, uniq)
        om, oa = orc.segmentation_eval_metric(tl, to, uniq)
        per = np.array([[float(m[u][k]) for k in ("precision", "recall", "iou")] for u in uniq], np.float32)
        acr = np.array([float(a[k]) for k in ("precision", "recall", "iou")], np.float32)
        close(np.array([[float(om[u][k]) for k in ("precision", "recall", "iou")] for u in uniq]), per, 0, "seg per-class %d" % i)
        close(np.array([float(oa[k]) for k in ("precision", "recall", "iou")]), acr, 0, "seg across %d" % i)
        fx["seg_unique_%d" % i] = np.array(uniq, np.int64)
        fx["seg_per_class_%d" % i] = per
        fx["seg_across_%d" % i] = acr
    # reconstruction metrics: two batches (one with fractional mask edges), both mask polarities
    for j, flip in enumerate((False, True)):
        tot = np.zeros(4)
        for b in range(2):
            g, mk = op.synth_batch(8100 + b, 3, 64, 64, fractional_edge=(b == 0))
            gen = np.random.Generator(np.random.PCG64(8200 + b)).random((3, 1, 64, 64), dtype=np.float32)
            r = orc.eval_recon_batch(torch.from_numpy(g), torch.from_numpy(gen), torch.from_numpy(mk), flip)
            tot += np.array([float(v) for v in r[1:]])
            fx["recon_out_sum_%d_%d" % (j, b)] = np.array(float(r[0].double().sum()))
        fx["recon_%d" % j] = (tot / 2).astype(np.float64)
    return fx


class _VggStandIn:
    """What loss.py:4 builds from torchvision (absent here): an object whose `.vgg19.features._modules`
    is the ordered module list of VGG-19 `features` (architecture restated in oracle/params.VGG19_CFG),
    loaded with the deterministic stand-in weights."""

    def __init__(self, P):
        layers, cin = [], 3
        for v in op.VGG19_CFG:
            if v == "M":
                layers.append(torch.nn.MaxPool2d(kernel_size=2, stride=2))
            else:
                layers += [torch.nn.Conv2d(cin, v, kernel_size=3, padding=1), torch.nn.ReLU(inplace=True)]
                cin = v
        self.vgg19 = types.SimpleNamespace(features=torch.nn.Sequential(*layers))
        self.vgg19.features.load_state_dict({k[len("features."):]: torch.from_numpy(v) for k, v in P.items()}, strict=True)


AUX_CASES = [(91, 2, 64), (92, 1, 128)]   # (seed, n, H=W)


def case_auxloss(ref_root, name):
    """The function texts of lib/models/loss.py from `def perceptual_loss` on (perceptual / style /
    gram / tv: pure torch, they only iterate a module list) are read from the reference AT GENERATION
    TIME and executed over the stand-in VGG; the file itself cannot be imported (its line 4 needs
    torchvision's pretrained download). Cross entropy is torch.nn.CrossEntropyLoss as the plugin calls it."""
    src = open(os.path.join(ref_root, "lib/models/loss.py")).read()
    ns = {"torch": torch}
    exec(compile(src[src.index("def perceptual_loss"):], "reference:loss.py", "exec"), ns)
    fx = {"aux_cases": np.array(AUX_CASES, dtype=np.int64)}
    for i, (seed, n, hw) in enumerate(AUX_CASES):
        P = op.make_vgg19_params(seed)
        ns["vgg"] = _VggStandIn(P)
        TP = {k: torch.from_numpy(v) for k, v in P.items()}
        g, mk = op.synth_batch(seed + 1, n, hw, hw)
        gen = np.random.Generator(np.random.PCG64(seed + 2)).random((n, 1, hw, hw), dtype=np.float32)
        ground = torch.from_numpy(g)
        out = torch.from_numpy(gen * np.ceil(mk) + g * (1 - np.ceil(mk)))            # an inpainted composite
        p, s = ns["perceptual_and_style_loss"](out, ground, weight_p=0.01, weight_s=0.01)   # plugin weights :216
        p1 = ns["perceptual_loss"](out, ground)
        s1 = ns["style_loss"](out, ground)
        op_, os_, pt, st = orc.perceptual_and_style_loss(TP, out, ground, 0.01, 0.01)
        close(op_, p, 1e-6, "perceptual %d" % i)
        close(os_, s, 1e-6, "style %d" % i)
        close(0.05 * sum(pt), p1, 1e-6, "perceptual_loss %d" % i)
        close(0.1 * sum(st), s1, 1e-6, "style_loss %d" % i)
        tv = ns["tv_loss"](out, 1)
        close(orc.tv_loss(out, 1), tv, 0, "tv %d" % i)
        feats = orc.vgg19_tap_features(TP, out)
        close(orc.gram_matrix(feats[1]), ns["gram_matrix"](feats[1]), 0, "gram %d" % i)
        fx["perceptual_%d" % i] = np.array(float(p)); fx["style_%d" % i] = np.array(float(s))
        fx["p_terms_%d" % i] = np.array([float(v) for v in pt]); fx["s_terms_%d" % i] = np.array([float(v) for v in st])
        fx["tv_%d" % i] = np.array(float(tv))
        fx["tap_absmean_%d" % i] = np.array([float(f.abs().mean()) for f in feats])
        fx["tap3_head_%d" % i] = feats[3][0, :8, :4, :4].numpy()
        # tv gradient and the weighted cross entropy (+ gradient) via autograd
        x = out.clone().requires_grad_(True)
        ns["tv_loss"](x, 1).backward()
        fx["tv_grad_sum_abs_%d" % i] = np.array(float(x.grad.abs().sum()))
        fx["tv_grad_head_%d" % i] = x.grad[0, 0, :4, :6].numpy()
        labels, logits = op.synth_segmentation(seed + 3, n, 4, hw, hw)
        z = torch.tanh(torch.from_numpy(logits)).requires_grad_(True)               # the seg U-Net ends in Tanh
        w = torch.tensor([0, 1.2, 0.7, 0.7])
        ce = torch.nn.CrossEntropyLoss(weight=w)(z, torch.from_numpy(labels))
        ce.backward()
        close(orc.weighted_cross_entropy(z.detach(), torch.from_numpy(labels), w), ce.detach(), 0, "ce %d" % i)
        fx["ce_%d" % i] = np.array(float(ce)); fx["ce_grad_sum_abs_%d" % i] = np.array(float(z.grad.abs().sum()))
        fx["ce_grad_head_%d" % i] = z.grad[0, :, :3, :5].numpy()
    return fx


def case_segnet(networks, name, seed, N, HW):
    """The frozen face-parsing network exactly as train.py:171-175 builds it (UnetGenerator(1,4,7,ngf=32,
    BatchNorm2d affine+running stats, use_dropout='False').eval()) with deterministic parameters, the
    weighted cross entropy of wgan_perceptual_style_faceparsing.py:67-68,212-213 on its output, and the
    gradient of that loss w.r.t. the input image (what reaches the generator)."""
    import functools
    P = op.make_unet_params(seed, num_downs=7, ngf=32, in_c=1, out_c=4)
    net = networks.UnetGenerator(1, 4, 7, ngf=32, norm_layer=functools.partial(torch.nn.BatchNorm2d, affine=True, track_running_stats=True),
                                 use_dropout='False')
    load(net, P)
    net.eval()
    g, _ = op.synth_batch(seed + 1, N, HW, HW)
    labels, _ = op.synth_segmentation(seed + 2, N, 4, HW, HW)
    x = torch.from_numpy(g).requires_grad_(True)
    y = net(x)
    ce = torch.nn.CrossEntropyLoss(weight=torch.tensor([0, 1.2, 0.7, 0.7]))(y, torch.from_numpy(labels))
    (0.01 * ce).backward()
    # oracle: same forward / same gradient
    TP = orc.to_torch(P, requires_grad=False)
    xo = torch.from_numpy(g).requires_grad_(True)
    yo = orc.unet_forward(TP, xo, 7, False, None)
    close(yo.detach(), y.detach(), 1e-5, "segnet forward")
    (0.01 * orc.weighted_cross_entropy(yo, torch.from_numpy(labels), torch.tensor([0, 1.2, 0.7, 0.7]))).backward()
    assert float((xo.grad - x.grad).abs().max()) <= 1e-5 * float(x.grad.abs().max()), "oracle != reference for segnet input gradient"
    return dict(seed=np.array(seed), N=np.array(N), HW=np.array(HW), ce=np.array(float(ce)),
                out_absmean=np.array(float(y.abs().mean())), out_head=y.detach()[0, :, :4, :6].numpy(),
                xgrad_abssum=np.array(float(x.grad.abs().sum())), xgrad_head=x.grad[0, 0, :6, :8].numpy(),
                xgrad_full=x.grad.numpy().astype(np.float32), out_full=y.detach().numpy().astype(np.float16))


def case_config5(networks, util, ref_root, name, seed, N, pattern):
    """wgan_perceptual_style_faceparsing.py:136-232 replayed on the reference modules (generator, critic,
    frozen face-parsing UnetGenerator(1,4,7,ngf=32).eval()), the reference's RMSELoss class and its
    perceptual_and_style_loss / tv_loss functions (texts executed at generation time over the stand-in
    VGG-19), torch's CrossEntropyLoss(weight=[0,1.2,0.7,0.7]) and RMSprop, one=+1 / mone=-1. Deviations forced
    by the file itself: `recon_loss` (:222) is undefined -> recon_global + recon_local; LocalLoss(RMSELoss())
    (:62) raises at construction -> the oracle's masked-RMSE extension."""
    import functools
    src = open(os.path.join(ref_root, "lib/models/loss.py")).read()
    ns = {"torch": torch, "nn": torch.nn}
    exec(compile(src[src.index("class RMSELoss"):src.index("def perceptual_loss")], "reference:loss.py", "exec"), ns)
    exec(compile(src[src.index("def perceptual_loss"):], "reference:loss.py", "exec"), ns)
    PG, PD = op.make_unet_params(seed), op.make_patchgan_params(seed + 1)
    PS = op.make_unet_params(seed + 2, num_downs=7, ngf=32, in_c=1, out_c=4)
    PV = op.make_vgg19_params(seed + 3)
    ns["vgg"] = _VggStandIn(PV)
    net_G = load(networks.get_network("generator", "unet"), PG)
    net_D = load(networks.PatchGANDiscriminator(sigmoid=False), PD)
    seg = load(networks.UnetGenerator(1, 4, 7, ngf=32, norm_layer=functools.partial(torch.nn.BatchNorm2d, affine=True, track_running_stats=True),
                                      use_dropout='False'), PS)
    seg.eval()
    cap = MaskCapture(net_G, 7)
    rmse_global = ns["RMSELoss"]()
    ce_crit = torch.nn.CrossEntropyLoss(weight=torch.tensor([0, 1.2, 0.7, 0.7]))
    G_opt = torch.optim.RMSprop(net_G.parameters(), lr=0.00005)
    D_opt = torch.optim.RMSprop(net_D.parameters(), lr=0.00005)
    one = torch.ones(1)
    mone = one * -1
    OG, OD = orc.to_torch(PG), orc.to_torch(PD)
    OS = orc.to_torch(PS, requires_grad=False)
    OV = {k: torch.from_numpy(v) for k, v in PV.items()}
    oG, oD = orc.RMSprop(orc.trainable(OG)), orc.RMSprop(orc.trainable(OD))
    fx = dict(seed=seed, N=N, pattern=np.array(pattern, np.int64))
    for it, upd in enumerate(pattern):
        g_np, m_np = op.synth_batch(seed * 100 + it, N, 128, 128)
        seg_np, _ = op.synth_segmentation(seed * 100 + 50 + it, N, 4, 128, 128)
        ground, mask, segment = torch.from_numpy(g_np), torch.ceil(torch.from_numpy(m_np)), torch.from_numpy(seg_np)
        masked = ground * (1 - mask)
        inpainted = masked + net_G(masked) * mask
        util.set_requires_grad([net_D], True)
        D_opt.zero_grad()
        d_loss_real = torch.mean(net_D(ground)).view(1)
        d_loss_real.backward(one)
        d_loss_fake = torch.mean(net_D(inpainted.detach())).view(1)
        d_loss_fake.backward(mone)
        D_opt.step()
        for p in net_D.parameters():
            p.data.clamp_(-0.01, 0.01)
        rec = dict(d_loss_real=float(d_loss_real), d_loss_fake=float(d_loss_fake))
        if upd:
            util.set_requires_grad([net_D], False)
            G_opt.zero_grad()
            g_adv = torch.mean(net_D(inpainted).view(-1)).view(1)
            recon_global = rmse_global(ground, inpainted)
            recon_local = orc.local_loss(ground, inpainted, mask, base="rmse")
            face = 0.01 * ce_crit(seg(inpainted), segment)
            g_p, g_s = ns["perceptual_and_style_loss"](inpainted, ground, weight_p=0.01, weight_s=0.01)
            g_tv = ns["tv_loss"](inpainted, tv_weight=1)
            g_loss = g_adv + recon_global + recon_local + g_p + g_s + face + g_tv
            g_loss.backward()
            _, g_absmean, _ = grad_stats(net_G)
            G_opt.step()
            rec.update(g_adv=float(g_adv), recon_global=float(recon_global), recon_local=float(recon_local), face_parsing=float(face),
                       perceptual=float(g_p), style=float(g_s), tv=float(g_tv), g_loss=float(g_loss), g_grad_absmean=g_absmean)
        rec.update(g_param_stats=param_stats(net_G), d_param_stats=param_stats(net_D))
        _record_step(fx, it, rec)
        fx.update({f"it{it}_{k}": v for k, v in pack_masks(cap.masks).items()})
        o = orc.wgan_step(OG, OD, oG, oD, torch.from_numpy(g_np), torch.from_numpy(m_np), 7, dict(cap.masks), bool(upd), recon="rmse",
                          extra=orc.config5_extra(OV, OS, segment))
        keys = ("d_loss_real", "d_loss_fake") + (("g_adv", "recon", "recon_local", "face_parsing", "perceptual", "style", "tv", "g_loss") if upd else ())
        for k in keys:
            close(o[k], rec["recon_global" if k == "recon" else k], 5e-5, f"{name}.it{it}.{k}")
        gnames = [n for n, _ in net_G.named_parameters()]
        ostats = np.array([float(OG[n].double().abs().sum()) for n in gnames])
        close(ostats, rec["g_param_stats"][:, 1], 1e-6, f"{name}.it{it}.g_params")
        print(f"  {name} it{it}: upd={upd} " + " ".join(f"{k}={rec[k]:.6g}" for k in rec if isinstance(rec[k], float)))
    fx["g_param_names"] = np.array([n for n, _ in net_G.named_parameters()])
    return fx


LOSS_CASES = [  # (seed, N, H, W, fractional_edge): masks are binary rectangles, or carry values in (0,1) on their border
    (801, 2, 48, 40, False), (802, 3, 32, 64, True), (803, 1, 128, 128, False),
]


def case_losses(ref_root, name):
    """The loss objects the reference's loops build, recorded from the reference classes themselves: the texts of
    lib/models/loss.py:11-47 (RMSELoss, LocalLoss) are executed at generation time (the module as a whole does not
    import: loss.py:4), and instantiated as the reference does - `loss.RMSELoss()` (minimaxgan_rmse.py:62),
    `loss.LocalLoss(nn.L1Loss)` (evaluate.py:115) and `LocalLoss(nn.MSELoss)` (what `LocalLoss(RMSELoss)` turns itself into,
    loss.py:30-31) - beside the torch built-ins nn.L1Loss / nn.MSELoss / nn.BCELoss / torch.mean in the forms of
    minimaxgan_l1.py:61-62,134-168, experiment1_global_local_D.py:119,162-196 and wgan_l1.py:137-177. Values and full
    gradients w.r.t. the prediction; the oracle's functions are asserted equal."""
    src = open(os.path.join(ref_root, "lib/models/loss.py")).read()
    ns = {"torch": torch, "nn": torch.nn}
    exec(compile(src[src.index("class RMSELoss"):src.index("def perceptual_loss")], "reference:loss.py", "exec"), ns)
    RMSELoss, LocalLoss = ns["RMSELoss"], ns["LocalLoss"]
    fx = {"cases": np.array([[c[0], c[1], c[2], c[3], int(c[4])] for c in LOSS_CASES], dtype=np.int64)}
    for i, (seed, N, H, W, frac) in enumerate(LOSS_CASES):
        y_np, m_np = op.synth_batch(seed, N, H, W, fractional_edge=frac)
        yh_np, _ = op.synth_batch(seed + 50, N, H, W)
        y, mask = torch.from_numpy(y_np), torch.from_numpy(m_np)
        recs = {}

        def run(tag, fn, ofn):
            yh = torch.from_numpy(yh_np.copy()).requires_grad_(True)
            v = fn(yh)
            v.backward()
            yo = torch.from_numpy(yh_np.copy()).requires_grad_(True)
            vo = ofn(yo)
            vo.backward()
            close(float(vo), float(v), 1e-6, f"{name}.{i}.{tag}")
            close(yo.grad.numpy(), yh.grad.numpy(), 1e-6, f"{name}.{i}.{tag}.grad")
            recs[tag] = float(v)
            fx[f"c{i}_{tag}"] = np.array(float(v), dtype=np.float64)
            fx[f"c{i}_{tag}_grad"] = yh.grad.numpy().copy()

        run("l1", lambda t: torch.nn.L1Loss()(y, t), lambda t: orc.l1_loss(y, t))
        run("mse", lambda t: torch.nn.MSELoss()(y, t), lambda t: torch.mean((t - y) ** 2))
        run("rmse", lambda t: RMSELoss()(t, y), lambda t: orc.rmse_loss(t, y))
        run("local_l1", lambda t: LocalLoss(torch.nn.L1Loss)(t, y, mask), lambda t: orc.local_loss(t, y, mask, "l1"))
        run("local_mse", lambda t: LocalLoss(torch.nn.MSELoss)(t, y, mask), lambda t: orc.local_loss(t, y, mask, "mse"))
        # LocalLoss(RMSELoss) (class passed as evaluate.py / the config-5 plugin intend): the reference constructor calls
        # RMSELoss(reduction='none'), which its __init__ does not accept
        try:
            LocalLoss(RMSELoss)
            raised = 0
        except TypeError:
            raised = 1
        fx[f"c{i}_local_rmse_ctor_raises"] = np.array(raised)
        print(f"  {name} case {i}: " + " ".join(f"{k}={v:.6g}" for k, v in recs.items()))
    # adversarial scalars on (n,) critic outputs
    rng = np.random.Generator(np.random.PCG64(880))
    for j, n in enumerate((2, 16, 32)):
        logit = rng.standard_normal(n).astype(np.float32) * 2.0
        prob = (1.0 / (1.0 + np.exp(-logit))).astype(np.float32)
        if j == 0:
            prob[0] = 1.0          # log(1 - p) clamps at -100 (nn.BCELoss)
        fx[f"adv{j}_prob"], fx[f"adv{j}_logit"] = prob, logit
        for tag, src_np, fn, ofn in (
                ("bce1", prob, lambda p: torch.nn.BCELoss()(p, torch.ones(n)), lambda p: orc.bce_loss(p, torch.ones(n))),
                ("bce0", prob, lambda p: torch.nn.BCELoss()(p, torch.zeros(n)), lambda p: orc.bce_loss(p, torch.zeros(n))),
                ("lsgan1", prob, lambda p: torch.nn.MSELoss()(p, torch.ones(n)), lambda p: orc.mse_loss(p, torch.ones(n))),
                ("lsgan0", prob, lambda p: torch.nn.MSELoss()(p, torch.zeros(n)), lambda p: orc.mse_loss(p, torch.zeros(n))),
                ("mean", logit, lambda p: torch.mean(p).view(1), lambda p: p.mean().view(1))):
            p = torch.from_numpy(src_np.copy()).requires_grad_(True)
            v = fn(p)
            v.backward(torch.ones_like(v))
            q = torch.from_numpy(src_np.copy()).requires_grad_(True)
            vo = ofn(q)
            vo.backward(torch.ones_like(vo))
            close(float(vo), float(v), 1e-6, f"{name}.adv{j}.{tag}")
            close(q.grad.numpy(), p.grad.numpy(), 1e-6, f"{name}.adv{j}.{tag}.grad")
            fx[f"adv{j}_{tag}"] = np.array(float(v), dtype=np.float64)
            fx[f"adv{j}_{tag}_grad"] = p.grad.numpy().copy()
    return fx


RESIZE_CASES = [  # (seed, in_h, in_w, size)
    (201, 218, 178, 128), (202, 1024, 1024, 256), (203, 200, 300, 64), (204, 50, 50, 128), (205, 100, 37, 16),
    (206, 9, 1000, 7), (207, 128, 128, 128),
]


def case_resize(name):
    """transforms.Resize(size) + ToTensor() (train.py:69-72) = Pillow's antialiased bilinear resize of the 'L'
    image + /255. torchvision is absent; the arithmetic is Pillow's (importable here, 12.2.0): outputs recorded
    from PIL.Image.resize itself, oracle/resize_ref.py asserted bit-equal."""
    from PIL import Image
    from oracle import resize_ref as rr
    fx = {"cases": np.array(RESIZE_CASES, dtype=np.int64)}
    for i, (seed, h, w, size) in enumerate(RESIZE_CASES):
        a = op.synth_u8_image(seed, h, w)
        nh, nw = rr.resized_output_size(h, w, size)
        ref = np.asarray(Image.fromarray(a, mode="L").resize((nw, nh), Image.BILINEAR))
        assert np.array_equal(rr.resize_bilinear_u8(a, nh, nw), ref), "oracle != Pillow for case %d" % i
        fx["out_%d" % i] = ref
    return fx


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    torch.set_num_threads(8)
    networks, util = import_reference(args.ref)
    cases = OrderedDict(
        unet128_train=lambda: case_unet(networks, "unet128_train", 11, 2, 128, 7, True),
        unet128_eval=lambda: case_unet(networks, "unet128_eval", 12, 2, 128, 7, False),
        unet256_train=lambda: case_unet(networks, "unet256_train", 13, 1, 256, 7, True),
        unet64_nd6_train=lambda: case_unet(networks, "unet64_nd6_train", 14, 2, 64, 6, True),
        patchgan128=lambda: case_patchgan(networks, "patchgan128", 21, 3),
        minimax_steps=lambda: case_minimax(networks, util, "minimax_steps", 31, 2, 2),
        wgan_steps=lambda: case_wgan(networks, util, "wgan_steps", 41, 2, [0, 1, 0]),
        dual_d_step=lambda: case_dual(networks, util, "dual_d_step", 51, 2),
        optim=lambda: case_optim("optim"),
        init_parity=lambda: case_init(networks, "init_parity", 7),
        ssim=lambda: case_ssim(args.ref, "ssim"),
        evalmetrics=lambda: case_evalmetrics(args.ref, "evalmetrics"),
        auxloss=lambda: case_auxloss(args.ref, "auxloss"),
        segnet=lambda: case_segnet(networks, "segnet", 95, 2, 128),
        resize=lambda: case_resize("resize"),
        config5_steps=lambda: case_config5(networks, util, args.ref, "config5_steps", 97, 2, [0, 1]),
        losses=lambda: case_losses(args.ref, "losses"),
        unet128_instance=lambda: case_unet_norm(networks, "unet128_instance", 15, 2, 128, 7, "instance"),
        unet128_none=lambda: case_unet_norm(networks, "unet128_none", 16, 2, 128, 7, "none"),
    )
    for name, fn in cases.items():
        if args.only and name not in args.only.split(","):
            continue
        print(name)
        fx = fn()
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **fx)
    print("done")


if __name__ == "__main__":
    main()
