"""GPU tier: BASELINE.json's full sizes (256x256 bs=32 fp16; 512x512) through size-independent properties, plus
ragged / empty batches and size-limit refusals. The CPU oracle cannot run these sizes in seconds; parity at small
sizes is in the other files, here the same code paths are exercised at the sizes the benchmark uses."""
import numpy as np
import pytest
import torch

import gan_inpainting_amd  # noqa: F401
from gan_inpainting_amd import backend as B
from gan_inpainting_amd import optim, trainer
from gan_inpainting_amd.lib.models import networks

pytestmark = pytest.mark.gpu


def _batch(n, hw, seed):
    g = torch.Generator().manual_seed(seed)
    ground = torch.rand((n, 1, hw, hw), generator=g)
    mask = torch.zeros((n, 1, hw, hw))
    for i in range(n):
        y0, x0 = 16 + 3 * i, 24 + 2 * i
        mask[i, 0, y0:y0 + hw // 3, x0:x0 + hw // 2] = 0.5 if i % 2 else 1.0        # fractional values exercise ceil()
    return ground.cuda(), mask.cuda()


def test_headline_step_properties_256_bs32_fp16():
    """configs[3]: wgan_rmse 256x256 bs=32 fp16, two-stream + stacked critic, 6 batches with one generator update."""
    torch.manual_seed(7)
    G = networks.get_network("generator", "unet", dtype="fp16").cuda()
    D = networks.PatchGANDiscriminator(sigmoid=False, image_size=256, dtype="fp16").cuda()
    oG, oD = optim.RMSprop(G.parameters(), lr=5e-5), optim.RMSprop(D.parameters(), lr=5e-5)
    step = trainer.WGANStep(G, D, oG, oD, recon="rmse", clip=0.01, overlap=True, stacked=True)
    g0 = G.flat_params().clone()
    for it in range(6):
        ground, mask = _batch(32, 256, 100 + it)
        L = step(ground, mask, it == 4)
        torch.cuda.synchronize()
        inp = step.inpainted
        mc = torch.ceil(mask)
        assert torch.equal(inp[mc == 0], ground[mc == 0])                 # outside the ceil-ed mask: ground truth, bit for bit
        assert torch.isfinite(inp).all() and float(inp.abs().max()) <= 1.0 + 1e-6   # tanh output composited into [0,1] data
        assert all(torch.isfinite(v).all() for v in L.values())
        assert float(D.flat_params().abs().max()) <= 0.01 + 1e-9         # weight clipping after every critic update
        if it < 4:
            assert torch.equal(G.flat_params(), g0)                       # critic-only batches leave the generator alone
    assert not torch.equal(G.flat_params(), g0)
    assert step.poll_overflow() == 0                                      # no fp16 overflow at the chosen loss scales
    assert set(L) >= {"d_loss_real", "d_loss_fake", "g_adv", "recon"}


def test_generator_backward_is_linear_in_the_output_gradient_256_bs32():
    """d(params) for 2*dy equals 2 * d(params) for dy: the two forwards are bit-identical (split-K sums in a fixed order, exact
    BatchNorm statistics), every parameter gradient is summed in a fixed order, and powers of two commute with every fp16 /
    fp32 rounding of NORMAL numbers - what remains are fp16 gradient values in the subnormal range (below 6e-5 at this loss
    scale), whose rounding does not scale."""
    torch.manual_seed(3)
    G = networks.UnetGenerator(1, 1, 7, ngf=64, use_dropout=False, dtype="fp16").cuda().train()    # no dropout: same forward twice
    G.set_loss_scale(1024.0)
    x, _ = _batch(32, 256, 5)
    with torch.no_grad():   # an MSE-like output gradient (white-noise gradients cancel inside the network and only measure rounding)
        dy = ((G(x) - x) * (2.0 / x.numel())).contiguous()
    G.zero_grad()
    y, s, g = G._forward_raw(x)
    G._backward_raw(s, g, dy, False, True)
    g1 = G.flat_grads().clone()
    G.zero_grad()
    y2, s, g = G._forward_raw(x)
    G._backward_raw(s, g, 2 * dy, False, True)
    g2 = G.flat_grads()
    assert torch.equal(y, y2), "two train-mode forwards of the same batch differ"
    bad = (g2 != 2 * g1)
    rel = float((g2.double() - 2 * g1.double()).norm() / (2 * g1.double()).norm())
    print(f"linearity: {int(bad.sum())} of {g1.numel()} entries not exactly doubled, rel L2 {rel:.3e}")
    assert rel <= 1e-3


def test_eval_forward_is_independent_of_batch_composition_and_handle_size():
    """Eval mode (running statistics, no dropout): an image's output does not depend on its batch mates or on the
    handle's capacity (ragged last batch: 5 images through a handle sized for 32)."""
    torch.manual_seed(5)
    G = networks.get_network("generator", "unet", dtype="fp16").cuda().eval()
    x, _ = _batch(32, 256, 21)
    with torch.no_grad():
        full = G(x)
        part = G(x[:5].contiguous())                 # same handle (capacity 32), ragged batch
        G2 = networks.get_network("generator", "unet", dtype="fp16")
        G2.load_state_dict(G.state_dict())
        solo = G2.cuda().eval()(x[:5].contiguous())  # fresh handle sized for 5
    # the deep layers take different split-K shapes at different batch sizes: fp16-level agreement, not bitwise
    assert (full[:5] - part).abs().max().item() <= 2e-2
    assert torch.equal(part, solo)                       # same shapes, same kernels, fixed summation orders: bit for bit


def test_train_forward_is_permutation_equivariant():
    """Batch statistics do not depend on the order of the images."""
    torch.manual_seed(6)
    D = networks.PatchGANDiscriminator(sigmoid=False, image_size=256, dtype="fp16").cuda().train()
    x, _ = _batch(32, 256, 31)
    perm = torch.randperm(32, generator=torch.Generator().manual_seed(1)).cuda()
    with torch.no_grad():
        a = D(x)
        b = D(x[perm].contiguous())
    assert (a[perm] - b).abs().max().item() <= 1e-2 * max(1.0, a.abs().max().item())


def test_empty_and_oversized_requests_are_refused():
    G = networks.get_network("generator", "unet", dtype="fp16").cuda()
    with pytest.raises((B.BackendError, ValueError, RuntimeError)):
        G(torch.empty((0, 1, 256, 256), device="cuda"))                       # empty batch
    with pytest.raises((B.BackendError, ValueError)):
        G(torch.rand((1, 1, 200, 200), device="cuda"))                        # not a power-of-two size
    with pytest.raises((B.BackendError, ValueError)):
        G(torch.rand((1, 3, 256, 256), device="cuda"))                        # 3-channel input
    vgg = networks.VGG19Wrapper(max_pairs=40).cuda()
    with pytest.raises(B.BackendError):                                       # 80 images x 1024^2 x 64 ch > 2^31 elements
        vgg.perceptual_and_style(torch.rand((40, 1, 1024, 1024), device="cuda"), torch.rand((40, 1, 1024, 1024), device="cuda"), 1, 1)


def test_vgg_512_symmetry_and_zero_distance():
    """configs[4] size: perceptual/style of (x, x) vanish; the terms are symmetric in their arguments."""
    torch.manual_seed(8)
    vgg = networks.VGG19Wrapper(max_pairs=2).cuda()
    x, _ = _batch(2, 512, 41)
    y, _ = _batch(2, 512, 42)
    p0, s0 = vgg.perceptual_and_style(x, x, 1.0, 1.0)
    p1, s1 = vgg.perceptual_and_style(x, y, 1.0, 1.0)
    assert float(p0) == 0.0 and float(s0) == 0.0     # both images of a pair go through the same fixed-order sums (no atomics)
    p2, s2 = vgg.perceptual_and_style(y, x, 1.0, 1.0)
    assert float(p1) > 0 and float(p1) == float(p2) and float(s1) == float(s2)     # (a - b)^2 == (b - a)^2 term by term, same order


def test_dual_discriminator_step_properties_256_bs32_fp16():
    """configs[2]: experiment1_global_local_D 256x256 bs=32 fp16 (two LSGAN discriminators, G step first, mask not
    ceil-ed, output not composited), three batches."""
    torch.manual_seed(9)
    G = networks.get_network("generator", "unet", dtype="fp16").cuda()
    Dg = networks.PatchGANDiscriminator(sigmoid=True, image_size=256, dtype="fp16").cuda()
    Dl = networks.PatchGANDiscriminator(sigmoid=True, image_size=256, dtype="fp16").cuda()
    oG = optim.Adam(G.parameters(), lr=0.0002, betas=(0.5, 0.999))
    oD = optim.Adam(optim.chain(Dl.parameters(), Dg.parameters()), lr=0.0002, betas=(0.5, 0.999))
    step = trainer.DualDStep(G, Dg, Dl, oG, oD)
    for it in range(3):
        ground, mask = _batch(32, 256, 200 + it)
        w = [n.flat_params().clone() for n in (G, Dg, Dl)]
        L = step(ground, mask)
        torch.cuda.synchronize()
        assert all(torch.isfinite(v).all() for v in L.values()), {k: float(v) for k, v in L.items()}
        assert set(L) >= {"g_adv_global", "g_adv_local", "rmse_global", "rmse_local", "d_real_global", "d_fake_global", "d_real_local", "d_fake_local"}
        gen = step.gen
        assert torch.isfinite(gen).all() and float(gen.abs().max()) <= 1.0            # tanh output, not composited (:154)
        assert torch.equal(step.masked, ground * (1 - mask))                            # the mask is used as it comes (:144), fractional values too
        assert torch.equal(step.inpainted, mask * gen)                                  # the local discriminator's input (:157)
        # LSGAN losses of sigmoid outputs against 0 / 1 lie in [0, 1]; RMSE of [0,1] data against a tanh output in [0, 2]
        for k in ("g_adv_global", "g_adv_local", "d_real_global", "d_fake_global", "d_real_local", "d_fake_local"):
            assert 0.0 <= float(L[k]) <= 1.0, (k, float(L[k]))
        assert 0.0 < float(L["rmse_local"]) <= float(L["rmse_global"]) * 1.0001 <= 2.0  # masked difference <= full difference
        for n, w0 in zip((G, Dg, Dl), w):                                                # every network moved by at most lr per element (Adam, t = 1..3)
            d = (n.flat_params() - w0).abs().max().item()
            assert 0 < d <= 0.0002 * 1.8, d
    assert step.poll_overflow() == 0


def test_wgan_gp_step_properties_128_bs16_fp32():
    """configs[1]: wgan_l1 128x128 bs=16 fp32 with the gradient penalty (lambda 10) in place of the clipping, overlapped
    and stacked like the plugin runs it; 6 batches, generator update on the fifth."""
    torch.manual_seed(10)
    G = networks.get_network("generator", "unet", dtype="fp32").cuda()
    D = networks.PatchGANDiscriminator(sigmoid=False, image_size=128, dtype="fp32").cuda()
    oG, oD = optim.RMSprop(G.parameters(), lr=5e-5), optim.RMSprop(D.parameters(), lr=5e-5)
    step = trainer.WGANStep(G, D, oG, oD, recon="l1", gp_lambda=10.0, overlap=True, stacked=True)
    g0 = G.flat_params().clone()
    pens = []
    for it in range(6):
        ground, mask = _batch(16, 128, 300 + it)
        d0 = D.flat_params().clone()
        L = step(ground, mask, it == 4)
        torch.cuda.synchronize()
        assert all(torch.isfinite(v).all() for v in L.values())
        pens.append(float(L["gp"]))
        assert pens[-1] >= 0.0
        mc = torch.ceil(mask)
        assert torch.equal(step.inpainted[mc == 0], ground[mc == 0])
        assert not torch.equal(D.flat_params(), d0)
        if it < 4:
            assert torch.equal(G.flat_params(), g0)
    assert not torch.equal(G.flat_params(), g0)
    assert float(D.flat_params().abs().max()) > 0.011          # not clipped: the penalty replaces wgan_l1.py:151-153
    # the penalty of a critic with |grad_x D| far from 1 is positive; with lambda = 0 the same call returns exactly 0
    assert max(pens) > 0
    ground, mask = _batch(16, 128, 399)
    D.zero_grad()
    z = D.gradient_penalty(ground, (ground * 0.5).contiguous(), torch.full((16,), 0.25), lam=0.0)
    assert float(z) == 0.0 and float(D.flat_grads().abs().max()) == 0.0


def test_config5_step_properties_512_bs8_fp16():
    """configs[4]: wgan_perceptual_style_faceparsing 512x512 bs=8 fp16: critic batch, then a batch with the generator
    update carrying every extra (VGG-19 perceptual / style constants, TV, frozen ngf=32 face-parsing network + weighted CE)."""
    import functools
    torch.manual_seed(11)
    G = networks.get_network("generator", "unet", dtype="fp16").cuda()
    D = networks.PatchGANDiscriminator(sigmoid=False, image_size=512, dtype="fp16").cuda()
    seg = networks.UnetGenerator(1, 4, 7, ngf=32, norm_layer=functools.partial(torch.nn.BatchNorm2d, affine=True, track_running_stats=True),
                                 use_dropout='False', dtype="fp16").cuda()
    vgg = networks.VGG19Wrapper(max_pairs=8).cuda()
    oG, oD = optim.RMSprop(G.parameters(), lr=5e-5), optim.RMSprop(D.parameters(), lr=5e-5)
    step = trainer.WGANPerceptualStep(G, D, oG, oD, vgg=vgg, segment_model=seg, clip=0.01, overlap=True)
    seg_w = step.seg.flat_params().clone()
    g0 = G.flat_params().clone()
    for it, upd in enumerate((False, True)):
        ground, mask = _batch(8, 512, 400 + it)
        labels = torch.randint(0, 4, (8, 512, 512), generator=torch.Generator().manual_seed(it)).cuda()
        L = step(ground, mask, upd, segment=labels)
        torch.cuda.synchronize()
        assert all(torch.isfinite(v).all() for v in L.values()), {k: float(v) for k, v in L.items()}
        mc = torch.ceil(mask)
        assert torch.equal(step.inpainted[mc == 0], ground[mc == 0])
        assert float(D.flat_params().abs().max()) <= 0.01 + 1e-9
        if not upd:
            assert torch.equal(G.flat_params(), g0)
    assert set(L) >= {"d_loss_real", "d_loss_fake", "g_adv", "recon_global", "recon_local", "tv", "perceptual", "style", "face_parsing"}
    assert not torch.equal(G.flat_params(), g0)
    assert torch.equal(step.seg.flat_params(), seg_w)                     # the face-parsing network is frozen (:67-68)
    assert float(L["recon_local"]) > 0 and float(L["tv"]) > 0 and float(L["perceptual"]) > 0 and float(L["style"]) > 0
    # CE of a random-init 4-class net with weights [0,1.2,0.7,0.7], times 0.01: around 0.01 * ln 4
    assert 0.002 < float(L["face_parsing"]) < 0.1, float(L["face_parsing"])
    assert step.poll_overflow() == 0
