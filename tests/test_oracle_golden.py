"""CPU tier: the oracle (oracle/torch_ref.py) against the fixtures recorded from the imported
reference (tests/golden/make_golden.py). This is the parity pin of the oracle."""
import numpy as np
import pytest
import torch

from oracle import params as op
from oracle import torch_ref as orc
from util_golden import load, relerr, unpack_masks


@pytest.mark.parametrize("name", ["unet128_train", "unet128_eval", "unet64_nd6_train"])
def test_unet_forward_backward(name):
    fx = load(name)
    seed, N, HW, nd, train = (int(fx[k]) for k in ("seed", "N", "HW", "num_downs", "train"))
    P = orc.to_torch(op.make_unet_params(seed, num_downs=nd))
    ground, mask = op.synth_batch(seed + 7, N, HW, HW)
    x = torch.from_numpy(ground * (1 - mask))
    out = orc.unet_forward(P, x, nd, bool(train), unpack_masks(fx))
    assert relerr(out.detach().numpy(), fx["out"]) < 1e-5
    if not train:
        return
    R = torch.from_numpy(np.random.Generator(np.random.PCG64(seed + 99)).standard_normal(
        size=(N, 1, HW, HW), dtype=np.float32))
    (out * R).sum().backward()
    names = [str(n) for n in fx["grad_names"]]
    for i, n in enumerate(names):
        assert relerr(float(P[n].grad.abs().mean()), fx["grad_absmean"][i]) < 1e-4, n
        assert relerr(P[n].grad.reshape(-1)[:32].numpy(), fx[f"ghead_{i}"], floor=1e-4) < 1e-3, n
    for k in fx.files:
        if k.startswith("bn::"):
            assert relerr(P[k[4:]].numpy(), fx[k]) < 1e-5, k


def test_patchgan():
    fx = load("patchgan128")
    seed, N = int(fx["seed"]), int(fx["N"])
    ground, _ = op.synth_batch(seed + 3, N, 128, 128)
    r = torch.from_numpy(np.random.Generator(np.random.PCG64(seed + 5)).standard_normal(size=(N, 1), dtype=np.float32))
    for sig, tag in ((True, "sig"), (False, "lin")):
        P = orc.to_torch(op.make_patchgan_params(seed, 128, 128))
        x = torch.from_numpy(ground).clone().requires_grad_(True)
        out = orc.patchgan_forward(P, x, sig, True)
        assert relerr(out.detach().numpy(), fx[f"out_{tag}"]) < 1e-5
        (out * r).sum().backward()
        assert relerr(x.grad.reshape(-1)[:256].numpy(), fx[f"dx_head_{tag}"], floor=1e-7) < 1e-3
        for i, n in enumerate(str(s) for s in fx["grad_names"]):
            assert relerr(float(P[n].grad.abs().mean()), fx[f"grad_absmean_{tag}"][i]) < 1e-4, n
        ev = orc.patchgan_forward(P, torch.from_numpy(ground), sig, False)
        assert relerr(ev.detach().numpy(), fx[f"out_eval_{tag}"]) < 1e-5


def _pstats(P, names):
    return np.array([[float(P[n].double().sum()), float(P[n].double().abs().sum())] for n in names])


def test_minimax_steps():
    fx = load("minimax_steps")
    seed, N, iters = int(fx["seed"]), int(fx["N"]), int(fx["iters"])
    PG, PD = orc.to_torch(op.make_unet_params(seed)), orc.to_torch(op.make_patchgan_params(seed + 1))
    oG, oD = orc.Adam(orc.trainable(PG)), orc.Adam(orc.trainable(PD))
    for it in range(iters):
        g, m = op.synth_batch(seed * 100 + it, N, 128, 128, fractional_edge=(it == 0))
        o = orc.minimax_step(PG, PD, oG, oD, torch.from_numpy(g), torch.from_numpy(m), 7,
                             unpack_masks(fx, f"it{it}_"))
        for k in ("d_loss_real", "d_loss_fake", "g_adv", "recon", "g_loss"):
            assert relerr(o[k], fx[f"it{it}_{k}"]) < 1e-4, (it, k)
        gn = [str(s) for s in fx["g_param_names"]]
        dn = [str(s) for s in fx["d_param_names"]]
        assert relerr(_pstats(PG, gn)[:, 1], fx[f"it{it}_g_param_stats"][:, 1]) < 1e-5
        assert relerr(_pstats(PD, dn)[:, 1], fx[f"it{it}_d_param_stats"][:, 1]) < 1e-5
    assert relerr(o["inpainted"].numpy(), fx["final_inpainted"], floor=1e-3) < 1e-3


def test_wgan_steps():
    fx = load("wgan_steps")
    seed, N = int(fx["seed"]), int(fx["N"])
    PG, PD = orc.to_torch(op.make_unet_params(seed)), orc.to_torch(op.make_patchgan_params(seed + 1))
    oG, oD = orc.RMSprop(orc.trainable(PG)), orc.RMSprop(orc.trainable(PD))
    for it, upd in enumerate(int(v) for v in fx["pattern"]):
        g, m = op.synth_batch(seed * 100 + it, N, 128, 128)
        o = orc.wgan_step(PG, PD, oG, oD, torch.from_numpy(g), torch.from_numpy(m), 7,
                          unpack_masks(fx, f"it{it}_"), bool(upd))
        keys = ("d_loss_real", "d_loss_fake") + (("g_adv", "recon", "g_loss") if upd else ())
        for k in keys:
            assert relerr(o[k], fx[f"it{it}_{k}"], floor=1e-4) < 1e-3, (it, k)
        dn = [str(s) for s in fx["d_param_names"]]
        st = _pstats(PD, dn)
        assert relerr(st[:, 1], fx[f"it{it}_d_param_stats"][:, 1]) < 1e-5
        # weight clipping, wgan_l1.py:151-153
        assert max(float(PD[n].abs().max()) for n in dn) <= 0.01 + 1e-9


def test_dual_d_step():
    fx = load("dual_d_step")
    seed, N = int(fx["seed"]), int(fx["N"])
    PG = orc.to_torch(op.make_unet_params(seed))
    PDg, PDl = orc.to_torch(op.make_patchgan_params(seed + 1)), orc.to_torch(op.make_patchgan_params(seed + 2))
    oG = orc.Adam(orc.trainable(PG))
    oD = orc.Adam(orc.trainable(PDl) + orc.trainable(PDg))
    g, m = op.synth_batch(seed * 100, N, 128, 128)
    o = orc.dual_d_step(PG, PDg, PDl, oG, oD, torch.from_numpy(g), torch.from_numpy(m), 7, unpack_masks(fx))
    for k in ("g_loss", "d_loss", "rmse_global", "rmse_local", "g_adv_global", "g_adv_local"):
        assert relerr(o[k], fx[k]) < 1e-4, k


def test_optimizers():
    fx = load("optim")
    for kind, cls in (("adam", orc.Adam), ("rmsprop", orc.RMSprop)):
        q = torch.from_numpy(fx["w0"].copy()).requires_grad_(True)
        oo = cls([q])
        for i in range(3):
            q.grad = torch.from_numpy(fx[f"g{i}"].copy())
            oo.step()
        assert relerr(q.detach().numpy(), fx[f"w_{kind}"]) < 1e-5


def test_wgan_cadence():
    # wgan_l1.py:157-163: period 140 while G_iter_count < 25 or % 500 == 0, else 5; never at batch 0
    assert not orc.wgan_update_g(0, 0)
    assert orc.wgan_update_g(140, 0) and not orc.wgan_update_g(5, 0)
    assert orc.wgan_update_g(5, 30) and not orc.wgan_update_g(6, 30)
    assert not orc.wgan_update_g(5, 500) and orc.wgan_update_g(280, 500)


def test_ssim_metric():
    """oracle.ssim == the imported reference's lib/pytorch_ssim (both reductions, 6 shapes / windows)."""
    fx = load("ssim")
    for i, (seed, n, c, h, w, ws) in enumerate(fx["cases"].tolist()):
        x, y = (torch.from_numpy(a) for a in op.synth_ssim_pair(seed, n, c, h, w))
        assert abs(float(orc.ssim(x, y, ws)) - float(fx[f"mean_{i}"])) <= 1e-6
        assert np.abs(orc.ssim(x, y, ws, size_average=False).numpy() - fx[f"per_sample_{i}"]).max() <= 1e-6
        assert abs(float(orc.ssim(x, x, ws)) - float(fx[f"self_{i}"])) <= 1e-6
        assert abs(float(fx[f"mean_{i}"]) - float(fx[f"mean64_{i}"])) <= 2e-6      # fp32 pipeline vs fp64


def test_evaluation_metrics():
    """oracle.segmentation_eval_metric == the reference's function (executed at fixture generation),
    incl. the -1 label quirk; oracle.eval_recon_batch reproduces its recorded values."""
    fx = load("evalmetrics")
    for i, (seed, n, K, h, w, neg) in enumerate(fx["seg_cases"].tolist()):
        labels, logits = op.synth_segmentation(seed, n, K, h, w, with_minus_one=bool(neg))
        uniq = fx[f"seg_unique_{i}"].tolist()
        m, a = orc.segmentation_eval_metric(torch.from_numpy(labels), torch.from_numpy(logits), uniq)
        per = np.array([[float(m[u][k]) for k in ("precision", "recall", "iou")] for u in uniq])
        assert np.abs(per - fx[f"seg_per_class_{i}"]).max() <= 1e-6
        assert np.abs(np.array([float(a[k]) for k in ("precision", "recall", "iou")]) - fx[f"seg_across_{i}"]).max() <= 1e-6
    for j, flip in enumerate((False, True)):
        tot = np.zeros(4)
        for b in range(2):
            g, mk = op.synth_batch(8100 + b, 3, 64, 64, fractional_edge=(b == 0))
            gen = np.random.Generator(np.random.PCG64(8200 + b)).random((3, 1, 64, 64), dtype=np.float32)
            r = orc.eval_recon_batch(torch.from_numpy(g), torch.from_numpy(gen), torch.from_numpy(mk), flip)
            tot += np.array([float(v) for v in r[1:]])
        assert np.abs(tot / 2 - fx[f"recon_{j}"]).max() <= 1e-6


def test_config5_losses():
    """oracle perceptual / style / tv / weighted CE == the values recorded by executing the reference's
    own loss.py functions over the stand-in VGG-19 (tests/golden/make_golden.py case_auxloss)."""
    fx = load("auxloss")
    seed, n, hw = fx["aux_cases"][0].tolist()
    P = {k: torch.from_numpy(v) for k, v in op.make_vgg19_params(seed).items()}
    g, mk = op.synth_batch(seed + 1, n, hw, hw)
    gen = np.random.Generator(np.random.PCG64(seed + 2)).random((n, 1, hw, hw), dtype=np.float32)
    out = torch.from_numpy((gen * np.ceil(mk) + g * (1 - np.ceil(mk))).astype(np.float32))
    p, s, pt, st = orc.perceptual_and_style_loss(P, out, torch.from_numpy(g), 0.01, 0.01)
    assert relerr(float(p), float(fx["perceptual_0"])) < 1e-5 and relerr(float(s), float(fx["style_0"])) < 1e-4
    assert relerr(np.array([float(v) for v in pt]), fx["p_terms_0"]) < 1e-5
    assert relerr(float(orc.tv_loss(out, 1)), float(fx["tv_0"])) < 1e-6
    labels, logits = op.synth_segmentation(seed + 3, n, 4, hw, hw)
    ce = orc.weighted_cross_entropy(torch.tanh(torch.from_numpy(logits)), torch.from_numpy(labels), torch.tensor([0, 1.2, 0.7, 0.7]))
    assert relerr(float(ce), float(fx["ce_0"])) < 1e-6


def test_resize_oracle_equals_pillow_fixture():
    """oracle/resize_ref.py == PIL.Image.resize(BILINEAR) bytes recorded in the fixture (and == Pillow itself when
    it is importable where the tests run)."""
    from oracle import resize_ref as rr
    fx = load("resize")
    for i, (seed, h, w, size) in enumerate(fx["cases"].tolist()):
        a = op.synth_u8_image(seed, h, w)
        nh, nw = rr.resized_output_size(h, w, size)
        assert np.array_equal(rr.resize_bilinear_u8(a, nh, nw), fx[f"out_{i}"]), i
        try:
            from PIL import Image
        except ImportError:
            continue
        assert np.array_equal(np.asarray(Image.fromarray(a, mode="L").resize((nw, nh), Image.BILINEAR)), fx[f"out_{i}"])


def test_loss_objects():
    """oracle's loss functions == the reference's RMSELoss / LocalLoss classes and the torch built-ins the plugins call
    (tests/golden/losses.npz, recorded from the executed reference class texts): values and full gradients."""
    fx = load("losses")
    for i, (seed, N, H, W, frac) in enumerate(fx["cases"].tolist()):
        y_np, m_np = op.synth_batch(seed, N, H, W, fractional_edge=bool(frac))
        yh_np, _ = op.synth_batch(seed + 50, N, H, W)
        y, mask = torch.from_numpy(y_np), torch.from_numpy(m_np)
        fns = dict(l1=lambda t: orc.l1_loss(y, t), mse=lambda t: torch.mean((t - y) ** 2), rmse=lambda t: orc.rmse_loss(t, y),
                   local_l1=lambda t: orc.local_loss(t, y, mask, "l1"), local_mse=lambda t: orc.local_loss(t, y, mask, "mse"))
        for tag, fn in fns.items():
            t = torch.from_numpy(yh_np.copy()).requires_grad_(True)
            v = fn(t)
            v.backward()
            assert relerr(float(v.detach()), fx[f"c{i}_{tag}"]) < 1e-6, (i, tag)
            ref = fx[f"c{i}_{tag}_grad"]
            assert np.abs(t.grad.numpy() - ref).max() <= 1e-6 * np.abs(ref).max(), (i, tag)
        assert int(fx[f"c{i}_local_rmse_ctor_raises"]) == 1     # LocalLoss(RMSELoss) cannot be built in the reference
    for j in range(3):
        prob, logit = fx[f"adv{j}_prob"], fx[f"adv{j}_logit"]
        n = len(prob)
        for tag, src, fn in (("bce1", prob, lambda p: orc.bce_loss(p, torch.ones(n))), ("bce0", prob, lambda p: orc.bce_loss(p, torch.zeros(n))),
                             ("lsgan1", prob, lambda p: orc.mse_loss(p, torch.ones(n))), ("lsgan0", prob, lambda p: orc.mse_loss(p, torch.zeros(n))),
                             ("mean", logit, lambda p: p.mean())):
            p = torch.from_numpy(src.copy()).requires_grad_(True)
            v = fn(p)
            v.backward()
            assert relerr(float(v.detach()), fx[f"adv{j}_{tag}"]) < 1e-6, (j, tag)
            ref = fx[f"adv{j}_{tag}_grad"]
            assert np.abs(p.grad.numpy() - ref).max() <= 1e-6 * np.abs(ref).max(), (j, tag)


@pytest.mark.parametrize("name", ["unet128_instance", "unet128_none"])
def test_unet_other_norm_layers(name):
    """UnetGenerator with get_norm_layer('instance') / ('none') (networks.py:30-45, :279-286): oracle == the reference
    class, train-mode forward + backward and eval-mode forward."""
    fx = load(name)
    seed, N, HW, nd, norm = int(fx["seed"]), int(fx["N"]), int(fx["HW"]), int(fx["num_downs"]), str(fx["norm"])
    P = op.make_unet_params(seed, num_downs=nd, norm=norm)
    OP = orc.to_torch(P)
    ground, mask = op.synth_batch(seed + 7, N, HW, HW)
    x = torch.from_numpy(ground * (1 - mask)).requires_grad_(True)
    R = torch.from_numpy(np.random.Generator(np.random.PCG64(seed + 99)).standard_normal(size=(N, 1, HW, HW), dtype=np.float32))
    y = orc.unet_forward(OP, x, nd, True, unpack_masks(fx), norm=norm)
    assert relerr(y.detach().numpy(), fx["out"], floor=1e-3) < 1e-4
    (y * R).sum().backward()
    assert np.abs(x.grad.numpy() - fx["dx"]).max() <= 1e-4 * np.abs(fx["dx"]).max()
    names = [str(n) for n in fx["grad_names"]]
    assert names == orc.named_parameter_keys(P)
    for n, ref in zip(names, fx["grad_absmean"]):
        assert abs(float(OP[n].grad.abs().mean()) - ref) <= 1e-4 * abs(ref) + 1e-7, n
    with torch.no_grad():
        ev = orc.unet_forward(OP, x.detach(), nd, False, None, norm=norm)
    assert relerr(ev.numpy(), fx["eval_out"], floor=1e-3) < 1e-4
