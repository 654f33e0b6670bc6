"""GPU tier: the single-layer C-ABI entry points at the shapes the HEADLINE benchmark dispatches (wgan_rmse 256x256: generator
at n = 32, stacked critic at n = 64), each against torch-CPU fp32 F.conv2d / F.conv_transpose2d / conv2d_weight, with the fused
epilogues the networks pass (BatchNorm statistics into the exact accumulators, activation backward with a second gradient,
BatchNorm-backward reduction) and an assertion on WHICH kernel served the shape (gi_debug_last_kernel). The layers are those of
UnetSkipConnectionBlock / PatchGANDiscriminator, lib/models/networks.py:285-318 and :335-345."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from gpu_util import B, from_nhwc, nhwc_dev, pack, quant, report  # noqa: E402

F16 = B.GI_F16
TOL = 2e-3   # relative to max|ref|: fp16 output rounding, fp32 accumulate


@pytest.fixture(params=["default", "igemm6", "igemm8"])
def family(request):
    """The halo-resident layers run three times: with the default choice (GI_IGEMM8 = 1: igemm8.hip - the same tile on four
    waves, two workgroups per CU - where the grid has >= 512 workgroups, igemm6 otherwise), with igemm6 everywhere and with
    igemm8 everywhere. expect(name, workgroups) maps igemm6's name to the kernel that must have served the layer."""
    mode = request.param
    if mode != "default":
        B.set_option("GI_IGEMM8", 0 if mode == "igemm6" else 2)

    def expect(name, workgroups):
        if mode == "igemm6" or not name.startswith("igemm6<") or ",64" in name or (mode == "default" and workgroups < 512):
            return name
        return f"igemm8<{name[7]}{',relu' if name.endswith('relu>') else ''}>"
    yield expect
    B.set_option("GI_IGEMM8", -1)


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(shape, generator=g) * 2 - 1) * scale


def _ws():
    return torch.full((16 << 20,), float("nan"), dtype=torch.float32, device="cuda")


def _acc(c):
    return torch.zeros(B.lib().gi_stat_acc_words(c), dtype=torch.int64, device="cuda")


def _read_acc(acc, c, reps, group=0):
    out = torch.empty((2, c), dtype=torch.float64, device="cuda")
    B.check(B.lib().gi_stat_acc_read(B.get_ctx(), B.ptr(acc), c, reps, group, B.ptr(out)))
    return out.cpu()


def _conv(n, H, W, cb, ca, ex=None, relu_in=0, seed=1):
    x = quant(_rand((n, cb, H, W), seed), F16)
    w = quant(_rand((ca, cb, 4, 4), seed + 1, 0.05), F16)
    packed, _ = pack(w, F16)
    xd = nhwc_dev(x, F16)
    out = torch.full((n, H // 2, W // 2, ca), float("nan"), dtype=torch.float16, device="cuda")
    ws = _ws()
    B.check(B.lib().gi_conv_s2_forward_ex(B.get_ctx(), F16, B.ptr(xd), B.ptr(packed), B.ptr(out), n, H, W, cb, cb, ca, ca, relu_in, 0,
                                          B.ptr(ws), ws.numel() * 4, C.byref(ex) if ex is not None else None))
    torch.cuda.synchronize()
    ref = F.conv2d(F.relu(x) if relu_in else x, w, None, stride=2, padding=1)
    return out, ref


def _convT(n, H, W, ca, cb, ex=None, relu_in=0, seed=5, x=None, w=None):
    x = quant(_rand((n, ca, H, W), seed), F16) if x is None else x
    w = quant(_rand((ca, cb, 4, 4), seed + 1, 0.05), F16) if w is None else w
    _, phase = pack(w, F16)
    xd = nhwc_dev(x, F16)
    out = torch.full((n, 2 * H, 2 * W, cb), float("nan"), dtype=torch.float16, device="cuda")
    ws = _ws()
    B.check(B.lib().gi_convT_s2_forward_ex(B.get_ctx(), F16, B.ptr(xd), B.ptr(phase), B.ptr(out), n, H, W, ca, ca, cb, cb, relu_in, 0,
                                           B.ptr(ws), ws.numel() * 4, C.byref(ex) if ex is not None else None))
    torch.cuda.synchronize()
    ref = F.conv_transpose2d(F.relu(x) if relu_in else x, w, None, stride=2, padding=1)
    return out, ref


# generator encoder d2..d7 at n = 32 and the critic's conv2..conv4 at n = 64 (two BatchNorm populations): Conv2d forward with
# the column statistics of the fp32 accumulators added to the exact accumulator block
CONV_FWD = [
    ("d2", 32, 128, 64, 128, 0, "igemm6<0,128>"),
    ("d3", 32, 64, 128, 256, 0, "igemm6<0,128>"),
    ("d4", 32, 32, 256, 512, 0, "igemm6<0,64>"),
    ("d5", 32, 16, 512, 512, 0, "igemm7<0,128>"),
    ("d6", 32, 8, 512, 512, 0, "igemm7<0,64>"),
    ("d7", 32, 4, 512, 512, 0, "igemm7<0,64>"),
    ("critic conv2", 64, 128, 64, 128, 2, "igemm6<0,128>"),
    ("critic conv3", 64, 64, 128, 256, 2, "igemm6<0,128>"),
    ("critic conv4", 64, 32, 256, 512, 2, "igemm6<0,128>"),
]


@pytest.mark.parametrize("case", CONV_FWD, ids=[c[0] for c in CONV_FWD])
def test_conv_forward_with_statistics_at_headline_shapes(case, family):
    name, n, HW, cb, ca, groups, kernel = case
    kernel = family(kernel, n * (HW // 2) ** 2 // 256 * (ca // 128))
    ex = B.IgemmEx()
    acc = _acc(ca)
    M = n * (HW // 2) ** 2
    ex.stat_acc = B.ptr(acc)
    ex.stat_reps = 4 if M // 128 > 1024 else (2 if M // 128 > 256 else 1)
    ex.stat_pg = M // 2 if groups == 2 else 0
    out, ref = _conv(n, HW, HW, cb, ca, ex)
    assert B.last_kernel() == kernel, (name, B.last_kernel())
    ok, msg = report(f"{name} conv forward [{B.last_kernel()}]", from_nhwc(out), ref, TOL)
    assert ok, msg
    assert ex.stat_used == 1, f"{name}: the kernel did not take the statistics"
    for g in range(max(groups, 1)):
        got = _read_acc(acc, ca, ex.stat_reps, g)
        r = ref if groups != 2 else ref[g * (n // 2):(g + 1) * (n // 2)]
        rs = torch.stack([r.double().sum((0, 2, 3)), (r.double() ** 2).sum((0, 2, 3))])
        scale = torch.stack([r.double().abs().sum((0, 2, 3)), (r.double() ** 2).sum((0, 2, 3))])
        err = float(((got - rs).abs() / scale).max())
        assert err <= 1e-5, f"{name} population {g}: column statistics off by {err:.2e} of sum|x|"


# generator decoder u7..u2 at n = 32: ConvTranspose2d forward with the in-place ReLU of the concat on the skip half only
CONVT_FWD = [
    ("u7", 32, 2, 512, 512, 0, "igemm7<1,64>"),
    ("u6", 32, 4, 1024, 512, 512, "igemm7<1,128>"),
    ("u5", 32, 8, 1024, 512, 512, "igemm3<1,64>"),
    ("u4", 32, 16, 1024, 256, 512, "igemm6<1,128,relu>"),
    ("u3", 32, 32, 512, 128, 256, "igemm6<1,128,relu>"),
    ("u2", 32, 64, 256, 64, 128, "igemm6<3,128,relu>"),
]


@pytest.mark.parametrize("case", CONVT_FWD, ids=[c[0] for c in CONVT_FWD])
def test_convT_forward_with_statistics_at_headline_shapes(case, family):
    name, n, HW, ca, cb, cend, kernel = case
    kernel = family(kernel, n * HW * HW // 256 * (4 * cb // 128))
    ex = B.IgemmEx()
    acc = _acc(cb)
    ex.stat_acc = B.ptr(acc)
    M = n * HW * HW
    ex.stat_reps = 4 if M // 128 * 4 > 1024 else (2 if M // 128 * 4 > 256 else 1)
    ex.relu_cend = cend
    x = quant(_rand((n, ca, HW, HW), 5), F16)
    if cend:   # the decoder half arrives already ReLU-ed (its BatchNorm pass stores it that way)
        x[:, cend:] = F.relu(x[:, cend:])
    out, ref = _convT(n, HW, HW, ca, cb, ex, relu_in=1 if cend else 0, x=x)
    assert B.last_kernel() == kernel, (name, B.last_kernel())
    ok, msg = report(f"{name} convT forward [{B.last_kernel()}]", from_nhwc(out), ref, TOL)
    assert ok, msg
    if ex.stat_used:
        got = _read_acc(acc, cb, ex.stat_reps, 0)
        rs = torch.stack([ref.double().sum((0, 2, 3)), (ref.double() ** 2).sum((0, 2, 3))])
        scale = torch.stack([ref.double().abs().sum((0, 2, 3)), (ref.double() ** 2).sum((0, 2, 3))])
        err = float(((got - rs).abs() / scale).max())
        assert err <= 1e-5, f"{name}: column statistics off by {err:.2e} of sum|x|"


def _sign_words(y_nhwc):
    """one 64-bit word per pixel, bit c = [y[p][c] > 0], as 8 bytes (little endian) - what the first layer's forward kernel writes"""
    pos = (y_nhwc[..., :64] > 0).to(torch.int32).reshape(-1, 8, 8)
    return (pos << torch.arange(8, device=pos.device, dtype=torch.int32)).sum(-1).to(torch.uint8).contiguous()


@pytest.mark.parametrize("bits", [False, True], ids=["fp16 mask", "sign words"])
def test_conv_dgrad_with_fused_leaky_relu_backward_of_d1(family, bits):
    """d2's input gradient at n = 32 (sub-pixel phases with 64 output channels: the dual-px kernel) with d1's activation
    backward in the epilogue: out = (g + [y > 0] * skip gradient) * (y > 0 ? 1 : 0.2), y = the saved LeakyReLU output of d1
    (networks.py:287: the skip IS lrelu(x)), both read from 128-channel concat buffers; `bits`: the mask as sign words."""
    n, Hs, ca, cb = 32, 64, 128, 64
    D = quant(_rand((n, ca, Hs, Hs), 31), F16)
    w = quant(_rand((ca, cb, 4, 4), 32, 0.05), F16)       # Conv2d weight [out = ca, in = cb] read as ConvTranspose2d [in, out]
    y = quant(_rand((n, cb, 2 * Hs, 2 * Hs), 33), F16)
    skip = quant(_rand((n, cb, 2 * Hs, 2 * Hs), 34), F16)
    ybuf = torch.zeros((n, 2 * Hs, 2 * Hs, 2 * cb), dtype=torch.float16, device="cuda")
    ybuf[..., :cb] = nhwc_dev(y, F16)
    gbuf = torch.zeros_like(ybuf)
    gbuf[..., :cb] = nhwc_dev(skip, F16)
    ex = B.IgemmEx()
    ex.mask = B.ptr(ybuf); ex.ldmask = 2 * cb; ex.mask_slope = 0.2   # noqa: E702
    ex.add = B.ptr(gbuf); ex.ldadd = 2 * cb                           # noqa: E702
    if bits:
        words = _sign_words(ybuf)
        ex.mask_bits = B.ptr(words)
    out, g = _convT(n, Hs, Hs, ca, cb, ex, x=D, w=w)
    assert B.last_kernel() == family("igemm6<3,128>", 1024) and ex.mask_applied == 1, (B.last_kernel(), ex.mask_applied)
    pos = (y > 0).float()
    ref = (g + pos * skip) * (pos + (1 - pos) * 0.2)
    ok, msg = report("d2 dgrad + fused LeakyReLU backward", from_nhwc(out), ref, TOL)
    assert ok, msg


@pytest.mark.parametrize("bits", [False, True], ids=["fp16 mask", "sign words"])
def test_critic_conv2_dgrad_with_fused_leaky_relu_backward_of_conv1(family, bits):
    """The stacked critic's conv2 input gradient (n = 64, 64x64 -> 128x128, 128 -> 64 channels) with conv1's LeakyReLU backward
    in the epilogue (networks.py:334-336), no second gradient; `bits`: the mask as the sign words conv1's forward wrote."""
    n, Hs, ca, cb = 64, 64, 128, 64
    D = quant(_rand((n, ca, Hs, Hs), 35), F16)
    w = quant(_rand((ca, cb, 4, 4), 36, 0.05), F16)
    y = quant(_rand((n, cb, 2 * Hs, 2 * Hs), 37), F16)
    y[:, :, ::7, ::5] = 0.0                                          # exact zeros take the slope ((float)y > 0 is false)
    ybuf = nhwc_dev(y, F16)
    ex = B.IgemmEx()
    ex.mask = B.ptr(ybuf); ex.ldmask = cb; ex.mask_slope = 0.2       # noqa: E702
    if bits:
        words = _sign_words(ybuf)
        ex.mask_bits = B.ptr(words)
    out, g = _convT(n, Hs, Hs, ca, cb, ex, x=D, w=w)
    assert B.last_kernel() == family("igemm6<3,128>", 2048) and ex.mask_applied == 1, (B.last_kernel(), ex.mask_applied)
    pos = (y > 0).float()
    ref = g * (pos + (1 - pos) * 0.2)
    ok, msg = report("critic conv2 dgrad + fused LeakyReLU backward", from_nhwc(out), ref, TOL)
    assert ok, msg


@pytest.mark.parametrize("layer", ["conv3", "conv2"])
def test_critic_dgrad_with_fused_batchnorm_backward_reduction(layer, family):
    """The stacked critic (n = 64, two BatchNorm populations): the input-gradient GEMM of conv4 / conv3 produces the gradient
    g w.r.t. LeakyReLU(BatchNorm(x)) of conv3 / conv2 and adds sum dz, sum dz * xhat per channel and population to the exact
    accumulators, dz = g * (fma(x, scale, shift) > 0 ? 1 : 0.2), xhat = (x - mean) * inv (networks.py:338-344)."""
    n, Hs, ca, cb = (64, 16, 512, 256) if layer == "conv3" else (64, 32, 256, 128)
    D = quant(_rand((n, ca, Hs, Hs), 41), F16)
    w = quant(_rand((ca, cb, 4, 4), 42, 0.05), F16)
    x = quant(_rand((n, cb, 2 * Hs, 2 * Hs), 43, 2.0), F16)          # the layer's raw convolution output
    st = torch.rand((2, 4, cb), generator=torch.Generator().manual_seed(44)) + 0.5   # [population][scale|shift|mean|inv][c]
    st[:, 1] -= 1.0
    st[:, 2] -= 1.0
    std = st.cuda().contiguous()
    xd = nhwc_dev(x, F16)
    acc = _acc(cb)
    ex = B.IgemmEx()
    ex.bwd_x = B.ptr(xd); ex.bwd_ldx = cb                      # noqa: E702
    base = std.data_ptr()
    ex.bwd_scale, ex.bwd_shift, ex.bwd_mean, ex.bwd_inv = base, base + 4 * cb, base + 8 * cb, base + 12 * cb
    ex.bwd_stride = 4 * cb
    ex.bwd_slope = 0.2
    ex.bwd_acc = B.ptr(acc)
    tiles = n * Hs * Hs // 256
    ex.bwd_reps = 4 if tiles * 4 > 1024 else (2 if tiles * 4 > 256 else 1)
    ex.bwd_pg = n // 2 * 4 * Hs * Hs
    out, g = _convT(n, Hs, Hs, ca, cb, ex, x=D, w=w)
    assert B.last_kernel() == family("igemm6<1,128>", n * Hs * Hs // 256 * (4 * cb // 128)) and ex.bwd_applied == 1, (B.last_kernel(), ex.bwd_applied)
    ok, msg = report(f"critic {layer} output gradient", from_nhwc(out), g, TOL)
    assert ok, msg
    gk = from_nhwc(out).double()                                   # the kernel reduces the fp16 gradient it stores
    for p in range(2):
        sl = slice(p * (n // 2), (p + 1) * (n // 2))
        sc, sh, mu, inv = (st[p, i].view(1, cb, 1, 1) for i in range(4))
        z = torch.addcmul(sh, x[sl], sc)                           # fma(x, scale, shift) in fp32 like the kernel
        dz = gk[sl] * torch.where(z > 0, 1.0, 0.2).double()
        xhat = ((x[sl] - mu) * inv).double()
        ref = torch.stack([dz.sum((0, 2, 3)), (dz * xhat).sum((0, 2, 3))])
        scale = torch.stack([dz.abs().sum((0, 2, 3)), (dz * xhat).abs().sum((0, 2, 3))])
        got = _read_acc(acc, cb, ex.bwd_reps, p)
        err = float(((got - ref).abs() / scale).max())
        assert err <= 2e-4, f"critic {layer} population {p}: fused BatchNorm-backward sums off by {err:.2e} of sum|dz|"


@pytest.mark.parametrize("case", [("u2 dgrad", 32, 128, 64, 256), ("u3 dgrad", 32, 64, 128, 512)], ids=["u2 dgrad", "u3 dgrad"])
def test_decoder_dgrad_with_batchnorm_backward_sums_in_the_upper_column_half(case, family):
    """A decoder level's input gradient (the 4x4 / s2 gather on its ConvTranspose2d weights, csrc/net.hip unet_backward): the upper
    half of the output columns is the gradient at the NEXT level's BatchNorm output, whose ReLU the consumer applies (networks.py:289
    uprelu): the tiles of that column range add sum dz, sum dz * xhat, dz = g * [fma(x, scale, shift) > 0], to the exact accumulators
    (gi_igemm_ex::bwd_c0 / bwd_c). Only the kernels that serve >= 512 workgroups take a column range; the others leave bwd_applied 0."""
    name, n, HW, cin, cout = case
    cb = cout // 2
    Hs = HW // 2
    x = quant(_rand((n, cb, Hs, Hs), 71, 2.0), F16)                  # the next level's raw decoder output
    st = torch.rand((4, cb), generator=torch.Generator().manual_seed(72)) + 0.5     # [scale|shift|mean|inv][c]
    st[1] -= 1.0
    st[2] -= 1.0
    std = st.cuda().contiguous()
    xd = nhwc_dev(x, F16)
    acc = _acc(cb)
    ex = B.IgemmEx()
    ex.bwd_x = B.ptr(xd); ex.bwd_ldx = cb                            # noqa: E702
    base = std.data_ptr()
    ex.bwd_scale, ex.bwd_shift, ex.bwd_mean, ex.bwd_inv = base, base + 4 * cb, base + 8 * cb, base + 12 * cb
    ex.bwd_stride = 4 * cb
    ex.bwd_slope = 0.0
    ex.bwd_acc = B.ptr(acc)
    tiles = n * Hs * Hs // 256
    ex.bwd_reps = 4 if tiles > 1024 else (2 if tiles > 256 else 1)
    ex.bwd_c0, ex.bwd_c = cb, cb
    out, g = _conv(n, HW, HW, cin, cout, seed=73, ex=ex)
    wgs = n * Hs * Hs // 256 * (cout // 128)
    kern = family("igemm6<0,128>", wgs)
    assert B.last_kernel() == kern, (name, B.last_kernel(), kern)
    ok, msg = report(f"{name} output gradient", from_nhwc(out), g, TOL)
    assert ok, msg
    if not kern.startswith("igemm8"):
        assert ex.bwd_applied == 0, "only igemm8 takes a column range"
        return
    assert ex.bwd_applied == 1
    gk = from_nhwc(out).double()[:, cb:]                             # the kernel reduces the fp16 gradient it stores
    sc, sh, mu, inv = (st[i].view(1, cb, 1, 1) for i in range(4))
    z = torch.addcmul(sh, x, sc)
    dz = gk * (z > 0).double()
    xhat = ((x - mu) * inv).double()
    ref = torch.stack([dz.sum((0, 2, 3)), (dz * xhat).sum((0, 2, 3))])
    scale = torch.stack([dz.abs().sum((0, 2, 3)), (dz * xhat).abs().sum((0, 2, 3))])
    got = _read_acc(acc, cb, ex.bwd_reps, 0)
    err = float(((got - ref).abs() / scale).max())
    assert err <= 2e-4, f"{name}: BatchNorm-backward sums of the upper column half off by {err:.2e} of sum|dz|"


# The generator's input gradients at n = 32 as unet_backward issues them (csrc/net.hip): a decoder level's is the 4x4 / s2 gather on
# its ConvTranspose2d weights (u2: 64 -> 256 channels onto the 64x64 grid, u3, u4), an encoder level's the sub-pixel phases on its
# Conv2d weights (d3: 256 -> 128 channels onto the 64x64 grid, d4); networks.py:285-309 builds the layers.
DGRAD = [
    ("u2 dgrad", "conv", 32, 128, 64, 256, "igemm6<0,128>"),
    ("u3 dgrad", "conv", 32, 64, 128, 512, "igemm6<0,128>"),
    ("u4 dgrad", "conv", 32, 32, 256, 1024, "igemm6<0,128>"),
    ("d3 dgrad", "convT", 32, 32, 256, 128, "igemm6<1,128>"),
    ("d4 dgrad", "convT", 32, 16, 512, 256, "igemm6<1,128>"),
]


@pytest.mark.parametrize("case", DGRAD, ids=[c[0] for c in DGRAD])
def test_generator_input_gradients_at_headline_shapes(case, family):
    name, kind, n, HW, cin, cout, kernel = case
    if kind == "conv":
        wgs = n * (HW // 2) ** 2 // 256 * (cout // 128)
        out, ref = _conv(n, HW, HW, cin, cout, seed=61)
    else:
        wgs = n * HW * HW // 256 * (4 * cout // 128)
        out, ref = _convT(n, HW, HW, cin, cout, seed=63)
    assert B.last_kernel() == family(kernel, wgs), (name, B.last_kernel(), wgs)
    ok, msg = report(f"{name} [{B.last_kernel()}]", from_nhwc(out), ref, TOL)
    assert ok, msg


# weight gradients: the critic's conv2..conv4 at n = 64 (wgrad3<4>: 4 x 16 pixel tiles) and the generator's layers at n = 32 with
# the operands unet_backward passes (csrc/net.hip): a decoder level's S is the whole concat buffer read through the parent's in-place
# ReLU (relu_S = 1, networks.py:289), an encoder level's L is the lower half of a concat buffer, i.e. rows 2 * cb channels apart
# (the upper half is NaN here: it must not be read). u5: 8 x 8 maps, wgrad3<3>, two pixel-range splits.
WGRAD = [   # (name, n, Hs, ca, cb, relu_S, ldL / cb, kernel)
    ("critic conv2", 64, 64, 128, 64, 0, 1, "wgrad3<4>"),
    ("critic conv3", 64, 32, 256, 128, 0, 1, "wgrad3<4>"),
    ("critic conv4", 64, 16, 512, 256, 0, 1, "wgrad3<4>"),
    ("u2", 32, 64, 256, 64, 1, 1, "wgrad3<4>"),
    ("u3", 32, 32, 512, 128, 1, 1, "wgrad3<4>"),
    ("u4", 32, 16, 1024, 256, 1, 1, "wgrad3<4>"),
    ("u5", 32, 8, 1024, 512, 1, 1, "wgrad3<3>"),
    ("u6", 32, 4, 1024, 512, 1, 1, "wgrad<f16>"),
    ("d2", 32, 64, 128, 64, 0, 2, "wgrad3<4>"),
    ("d3", 32, 32, 256, 128, 0, 2, "wgrad3<4>"),
    ("d4", 32, 16, 512, 256, 0, 2, "wgrad3<4>"),
    ("d5", 32, 8, 512, 512, 0, 2, "wgrad<f16>"),
]


@pytest.mark.parametrize("case", WGRAD, ids=[c[0] for c in WGRAD])
def test_weight_gradients_at_headline_shapes(case):
    name, n, Hs, ca, cb, relu_S, ldm, kernel = case
    S = quant(_rand((n, ca, Hs, Hs), 7, 0.5), F16)
    L = quant(_rand((n, cb, 2 * Hs, 2 * Hs), 8, 0.5), F16)
    ref = torch.nn.grad.conv2d_weight(L, (ca, cb, 4, 4), F.relu(S) if relu_S else S, stride=2, padding=1)
    Sd = nhwc_dev(S, F16)
    Ld = torch.full((n, 2 * Hs, 2 * Hs, ldm * cb), float("nan"), dtype=torch.float16, device="cuda")
    Ld[..., :cb] = nhwc_dev(L, F16)
    lib, ctx = B.lib(), B.get_ctx()
    nbytes = lib.gi_wgrad_s2_scratch_bytes(F16, n, Hs, Hs, ca, cb)
    ws = torch.full((max(nbytes // 4, 4),), float("nan"), dtype=torch.float32, device="cuda")
    outs = []
    for _ in range(2):
        dW = torch.full((ca, 4, 4, cb), 1.0, dtype=torch.float32, device="cuda")
        B.check(lib.gi_wgrad_s2_ws(ctx, F16, B.ptr(Sd), B.ptr(Ld), B.ptr(dW), n, Hs, Hs, ca, ca, cb, ldm * cb, relu_S, 1.0, B.ptr(ws), nbytes))
        torch.cuda.synchronize()
        outs.append(dW.clone())
    assert B.last_kernel() == kernel, (name, B.last_kernel())
    if name == "u5":
        assert nbytes == 2 * ca * 16 * cb * 4, "u5 runs with two pixel-range splits"
    assert torch.equal(outs[0], outs[1]), f"{name}: weight gradient differs between two runs"
    ok, msg = report(f"{name} weight gradient [{kernel}]", (outs[0] - 1.0).cpu().permute(0, 3, 1, 2), ref, TOL)
    assert ok, msg
