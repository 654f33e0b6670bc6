"""GPU tier: single-layer C-ABI entry points against torch-CPU fp32 references of the same op
(F.conv2d / F.conv_transpose2d / conv2d_weight), both compute types, split-K and direct paths."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from gpu_util import B, from_nhwc, nhwc_dev, pack, quant, report, tdt  # noqa: E402

TOL = {B.GI_F32: 2e-5, B.GI_F16: 2e-3}   # relative to max|ref|; fp16: fp16 output rounding, fp32 accumulate


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(shape, generator=g) * 2 - 1) * scale


CONV_CASES = [
    # n, H, W, cb(in), ca(out), use_ws
    (2, 16, 16, 64, 128, True),
    (2, 16, 16, 64, 128, False),
    (1, 64, 64, 64, 128, False),
    (2, 32, 32, 128, 64, False),     # 256x64 tile config
    (3, 8, 8, 256, 512, True),
    (2, 4, 4, 512, 512, True),       # 2x2 output, heavy padding
    (5, 2, 2, 512, 512, True),       # 1x1 output (bottleneck)
    (8, 128, 128, 64, 128, False),   # >= 128 tiles: the 256x128 LDS-DMA kernel (igemm3.hip) in fp16
    (7, 72, 72, 64, 512, False),     # igemm3 with a ragged last M tile (M = 9072) and non-power-of-two maps
    (8, 128, 128, 128, 64, False),   # igemm3, 64-column variant
    (8, 128, 128, 64, 512, False),   # igemm3, 256x256 two-stage variant (>= 256 such tiles)
    # use_ws = 2: scratch for one tile per split -> the split-K fix-up inside the GEMM kernel (last arriver per tile)
    (2, 64, 64, 64, 128, 2),         # M = 2048, one N tile
    (2, 32, 32, 128, 256, 2),        # M = 512
    (2, 16, 16, 256, 512, 2),        # M = 128: 128 x 64 tiles
    (3, 8, 8, 512, 512, 2),          # M = 48, ragged
    (5, 2, 2, 512, 512, 2),          # 1x1 output
    (32, 16, 16, 512, 512, 2),       # the generator's d5 at the headline configuration (M = 2048, K = 8192)
    (4, 32, 32, 64, 64, 2),          # 64 output channels: 256 x 64 tiles
    # rectangular maps (the C-ABI takes H and W separately; the networks of the reference are square)
    (16, 128, 256, 64, 128, False),  # 64 x 128 output maps, 512 tiles: igemm8's stride-2 gather on 8 x 32 patches
    (8, 256, 64, 64, 128, False),    # 128 x 32 output maps: 256 tiles (igemm6)
    (3, 16, 64, 128, 256, 2),        # 8 x 32 maps, split-K
    (2, 48, 16, 64, 128, False),     # 24 x 8 maps: not a power of two in one direction
]


@pytest.mark.parametrize("code", [B.GI_F32, B.GI_F16])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_s2_forward(code, case):
    n, H, W, cb, ca, use_ws = case
    x = quant(_rand((n, cb, H, W), 1), code)
    w = quant(_rand((ca, cb, 4, 4), 2, 0.05), code)
    ref = F.conv2d(x, w, None, stride=2, padding=1)
    packed, _ = pack(w, code)
    xd = nhwc_dev(x, code)
    out = torch.full((n, H // 2, W // 2, ca), float("nan"), dtype=tdt(code), device="cuda")
    ws = torch.zeros(n * (H // 2) * (W // 2) * ca, dtype=torch.float32, device="cuda") if use_ws else None
    if use_ws == 2:
        ws = torch.full((8 << 20,), float("nan"), dtype=torch.float32, device="cuda")
    B.check(B.lib().gi_conv_s2_forward(B.get_ctx(), code, B.ptr(xd), B.ptr(packed), B.ptr(out), n, H, W, cb, cb, ca, ca,
                                       0, 0, B.ptr(ws), ws.numel() * 4 if use_ws else 0))
    torch.cuda.synchronize()
    ok, msg = report(f"conv_s2 {case} dt={code}", from_nhwc(out), ref, TOL[code])
    assert ok, msg


@pytest.mark.parametrize("code", [B.GI_F32, B.GI_F16])
def test_conv_s2_relu_in_lrelu_out_strided(code):
    """relu on the input, LeakyReLU on the output, input read from / output written into wider
    buffers at channel offset 0 (the concat-buffer case)."""
    n, H, W, cb, ca = 2, 16, 16, 64, 128
    x = quant(_rand((n, cb, H, W), 3), code)
    w = quant(_rand((ca, cb, 4, 4), 4, 0.05), code)
    ref = F.leaky_relu(F.conv2d(F.relu(x), w, None, stride=2, padding=1), 0.2)
    packed, _ = pack(w, code)
    wide = torch.zeros((n, H, W, 2 * cb), dtype=tdt(code), device="cuda")
    wide[..., :cb] = nhwc_dev(x, code)
    wide[..., cb:] = 7.0   # must not be read
    out = torch.zeros((n, H // 2, W // 2, 2 * ca), dtype=tdt(code), device="cuda")
    B.check(B.lib().gi_conv_s2_forward(B.get_ctx(), code, B.ptr(wide), B.ptr(packed), B.ptr(out), n, H, W, cb, 2 * cb, ca,
                                       2 * ca, 1, B.ACT_LRELU, None, 0))
    torch.cuda.synchronize()
    ok, msg = report(f"conv_s2 relu/lrelu dt={code}", from_nhwc(out[..., :ca].contiguous()), ref, TOL[code])
    assert ok, msg
    assert float(out[..., ca:].abs().max()) == 0.0


CONVT_CASES = [
    # n, H, W, ca(in), cb(out), use_ws
    (2, 8, 8, 128, 64, False),      # 256x64 tile config
    (2, 8, 8, 256, 128, True),
    (1, 32, 32, 256, 128, False),
    (3, 1, 1, 512, 512, True),      # bottleneck 1x1 -> 2x2
    (2, 2, 2, 1024, 512, True),
    (8, 32, 32, 256, 128, False),   # igemm3 (fp16): 32 M tiles x 4 phases
    (3, 40, 40, 128, 256, False),   # igemm3, ragged M (4800), two N tiles
    (8, 32, 32, 128, 64, False),    # igemm3, 64-column variant (the generator's u2 shape family)
    (16, 32, 32, 128, 256, False),  # igemm3, 256x256 two-stage variant
    # use_ws = 2: split-K fix-up inside the GEMM kernel
    (2, 1, 1, 512, 512, 2),
    (2, 2, 2, 1024, 512, 2),
    (2, 4, 4, 1024, 512, 2),
    (2, 8, 8, 1024, 256, 2),
    (2, 16, 16, 512, 128, 2),
    # rectangular maps
    (16, 32, 64, 256, 128, False),  # 512 tiles x 4 phases: igemm8's phase mode on 8 x 32 patches
    (8, 64, 16, 128, 64, False),    # 64-channel layer: both px phases per workgroup (16-wide patches)
    (3, 4, 16, 512, 256, 2),        # split-K
    (2, 24, 8, 128, 128, False),    # 24 rows: not a power of two
    (2, 32, 32, 256, 64, 2),        # 64 output channels: 256 x 64 tiles
    (32, 4, 4, 1024, 512, 2),       # the generator's u6 at the headline configuration
]


@pytest.mark.parametrize("code", [B.GI_F32, B.GI_F16])
@pytest.mark.parametrize("case", CONVT_CASES)
def test_convT_s2_forward(code, case):
    n, H, W, ca, cb, use_ws = case
    x = quant(_rand((n, ca, H, W), 5), code)
    w = quant(_rand((ca, cb, 4, 4), 6, 0.05), code)   # ConvTranspose2d weight [in, out, 4, 4]
    ref = F.conv_transpose2d(x, w, None, stride=2, padding=1)
    _, phase = pack(w, code)
    xd = nhwc_dev(x, code)
    out = torch.full((n, 2 * H, 2 * W, cb), float("nan"), dtype=tdt(code), device="cuda")
    ws = torch.zeros(n * 4 * H * W * cb, dtype=torch.float32, device="cuda") if use_ws else None
    if use_ws == 2:
        ws = torch.full((8 << 20,), float("nan"), dtype=torch.float32, device="cuda")
    B.check(B.lib().gi_convT_s2_forward(B.get_ctx(), code, B.ptr(xd), B.ptr(phase), B.ptr(out), n, H, W, ca, ca, cb, cb,
                                        0, 0, B.ptr(ws), ws.numel() * 4 if use_ws else 0))
    torch.cuda.synchronize()
    ok, msg = report(f"convT_s2 {case} dt={code}", from_nhwc(out), ref, TOL[code])
    assert ok, msg


WGRAD_CASES = [
    # n, Hs, Ws, ca, cb, relu_S
    (2, 8, 8, 128, 64, 0),
    (2, 8, 8, 128, 64, 1),
    (1, 16, 16, 256, 128, 0),
    (3, 2, 2, 512, 256, 0),
    (2, 1, 1, 512, 512, 1),
    (4, 32, 32, 128, 64, 0),     # many K tiles -> split over pixel ranges
    (8, 16, 64, 128, 64, 0),     # rectangular: 4 x 16 pixel tiles of the halo kernels (wgrad3 / wgrad2)
    (8, 64, 8, 128, 64, 1),      # 8-wide maps: 8 x 8 pixel tiles
    (2, 32, 8, 256, 128, 0),     # 32 x 8 maps
    (2, 24, 8, 256, 128, 0),     # rows not a power of two: the register-staged kernel with divisions in its loader
    (3, 12, 20, 128, 64, 1),     # neither
]


@pytest.mark.parametrize("code", [B.GI_F32, B.GI_F16])
@pytest.mark.parametrize("case", WGRAD_CASES)
def test_wgrad_s2(code, case):
    n, Hs, Ws, ca, cb, relu_S = case
    S = quant(_rand((n, ca, Hs, Ws), 7), code)
    L = quant(_rand((n, cb, 2 * Hs, 2 * Ws), 8), code)
    ref = torch.nn.grad.conv2d_weight(L, (ca, cb, 4, 4), F.relu(S) if relu_S else S, stride=2, padding=1)
    Sd, Ld = nhwc_dev(S, code), nhwc_dev(L, code)
    dW = torch.zeros((ca, 4, 4, cb), dtype=torch.float32, device="cuda")
    B.check(B.lib().gi_wgrad_s2(B.get_ctx(), code, B.ptr(Sd), B.ptr(Ld), B.ptr(dW), n, Hs, Ws, ca, ca, cb, cb, relu_S, 0.5))
    # accumulate semantic: a second call adds again
    B.check(B.lib().gi_wgrad_s2(B.get_ctx(), code, B.ptr(Sd), B.ptr(Ld), B.ptr(dW), n, Hs, Ws, ca, ca, cb, cb, relu_S, 0.5))
    torch.cuda.synchronize()
    got = dW.cpu().permute(0, 3, 1, 2)
    ok, msg = report(f"wgrad {case} dt={code}", got, ref, 5e-5 if code == B.GI_F32 else 2e-3)
    assert ok, msg


@pytest.mark.parametrize("code", [B.GI_F32, B.GI_F16])
def test_wgrad_s2_scratch_path_is_deterministic_and_equal(code):
    """gi_wgrad_s2_ws: the pixel-range splits write partial tiles, a fixed-order pass adds them to dW. Same values
    as the atomics path (to rounding), accumulate semantic kept, and bit-identical from run to run."""
    n, Hs, Ws, ca, cb = 8, 32, 32, 128, 64          # 8192 pixels: many splits
    S = quant(_rand((n, ca, Hs, Ws), 17), code)
    L = quant(_rand((n, cb, 2 * Hs, 2 * Ws), 18), code)
    ref = torch.nn.grad.conv2d_weight(L, (ca, cb, 4, 4), S, stride=2, padding=1)
    Sd, Ld = nhwc_dev(S, code), nhwc_dev(L, code)
    lib, ctx = B.lib(), B.get_ctx()
    nbytes = lib.gi_wgrad_s2_scratch_bytes(code, n, Hs, Ws, ca, cb)
    assert nbytes > 0
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device="cuda")
    outs = []
    for _ in range(2):
        dW = torch.full((ca, 4, 4, cb), 1.0, dtype=torch.float32, device="cuda")      # pre-existing gradient: += semantic
        B.check(lib.gi_wgrad_s2_ws(ctx, code, B.ptr(Sd), B.ptr(Ld), B.ptr(dW), n, Hs, Ws, ca, ca, cb, cb, 0, 1.0, B.ptr(ws), nbytes))
        torch.cuda.synchronize()
        outs.append(dW.clone())
    assert torch.equal(outs[0], outs[1])
    got = (outs[0] - 1.0).cpu().permute(0, 3, 1, 2)
    ok, msg = report(f"wgrad scratch dt={code}", got, ref, 5e-5 if code == B.GI_F32 else 2e-3)
    assert ok, msg
    small = torch.empty(16, dtype=torch.float32, device="cuda")                       # too small: falls back to the atomics
    dW = torch.zeros((ca, 4, 4, cb), dtype=torch.float32, device="cuda")
    B.check(lib.gi_wgrad_s2_ws(ctx, code, B.ptr(Sd), B.ptr(Ld), B.ptr(dW), n, Hs, Ws, ca, ca, cb, cb, 0, 1.0, B.ptr(small), 64))
    ok, msg = report(f"wgrad fallback dt={code}", dW.cpu().permute(0, 3, 1, 2), ref, 5e-5 if code == B.GI_F32 else 2e-3)
    assert ok, msg


@pytest.mark.parametrize("code", [B.GI_F32, B.GI_F16])
def test_convT_s2_relu_strided_large(code):
    """Decoder form at igemm3 size: ReLU on the input, input read from the first half of a wider
    buffer (channel offset 0, ld = 2*ca), output written into a wider buffer."""
    n, H, W, ca, cb = 8, 32, 32, 128, 128
    x = quant(_rand((n, ca, H, W), 9), code)
    w = quant(_rand((ca, cb, 4, 4), 10, 0.05), code)
    ref = F.conv_transpose2d(F.relu(x), w, None, stride=2, padding=1)
    _, phase = pack(w, code)
    wide = torch.full((n, H, W, 2 * ca), 5.0, dtype=tdt(code), device="cuda")
    wide[..., :ca] = nhwc_dev(x, code)
    out = torch.zeros((n, 2 * H, 2 * W, 2 * cb), dtype=tdt(code), device="cuda")
    B.check(B.lib().gi_convT_s2_forward(B.get_ctx(), code, B.ptr(wide), B.ptr(phase), B.ptr(out), n, H, W, ca, 2 * ca, cb, 2 * cb,
                                        1, 0, None, 0))
    torch.cuda.synchronize()
    ok, msg = report(f"convT relu strided large dt={code}", from_nhwc(out[..., :cb].contiguous()), ref, TOL[code])
    assert ok, msg
    assert float(out[..., cb:].abs().max()) == 0.0


def test_igemm3_exact_integers():
    """Exact small-integer check at a size served by the LDS-DMA kernel (fragment / swizzle / ring order)."""
    g = torch.Generator().manual_seed(12)
    n, H, W, cb, ca = 8, 128, 128, 64, 128
    x = torch.randint(-2, 3, (n, cb, H, W), generator=g).float()
    w = torch.randint(-1, 2, (ca, cb, 4, 4), generator=g).float()
    ref = F.conv2d(x, w, None, stride=2, padding=1)
    packed, _ = pack(w, B.GI_F16)
    xd = nhwc_dev(x, B.GI_F16)
    out = torch.zeros((n, H // 2, W // 2, ca), dtype=torch.float16, device="cuda")
    B.check(B.lib().gi_conv_s2_forward(B.get_ctx(), B.GI_F16, B.ptr(xd), B.ptr(packed), B.ptr(out), n, H, W, cb, cb, ca, ca, 0, 0, None, 0))
    assert torch.equal(from_nhwc(out), ref)
    xs = torch.randint(-2, 3, (8, 256, 32, 32), generator=g).float()
    wt = torch.randint(-1, 2, (256, 128, 4, 4), generator=g).float()
    refT = F.conv_transpose2d(xs, wt, None, stride=2, padding=1)
    _, phase = pack(wt, B.GI_F16)
    xsd = nhwc_dev(xs, B.GI_F16)
    outT = torch.zeros((8, 64, 64, 128), dtype=torch.float16, device="cuda")
    B.check(B.lib().gi_convT_s2_forward(B.get_ctx(), B.GI_F16, B.ptr(xsd), B.ptr(phase), B.ptr(outT), 8, 32, 32, 256, 256, 128, 128, 0, 0,
                                        None, 0))
    assert torch.equal(from_nhwc(outT), refT)


@pytest.mark.parametrize("shape", [(8, 32, 32, 128, 64), (4, 64, 64, 64, 64), (16, 16, 16, 192, 64), (8, 32, 32, 64, 192)])
def test_halo_kernels_exact_integers_64_channel_tiles(shape):
    """ConvTranspose2d forward with 64-channel N tiles: the dual-px mode of the halo-resident kernel (both px
    sub-pixel phases per workgroup, shared halo of TW+2 columns), ReLU on the input, output written at a channel
    offset of a wider buffer, BatchNorm partial statistics rows per phase. Exact small integers."""
    n, H, W, ca, cb = shape
    g = torch.Generator().manual_seed(21)
    xs = torch.randint(-2, 3, (n, ca, H, W), generator=g).float()
    wt = torch.randint(-1, 2, (ca, cb, 4, 4), generator=g).float()
    refT = F.conv_transpose2d(F.relu(xs), wt, None, stride=2, padding=1)
    _, phase = pack(wt, B.GI_F16)
    xsd = nhwc_dev(xs, B.GI_F16)
    out = torch.full((n, 2 * H, 2 * W, cb + 64), 7.0, dtype=torch.float16, device="cuda")
    # gi_convT_s2_forward writes at channel offset 0 of an ld = cb + 64 buffer; the tail must stay untouched
    B.check(B.lib().gi_convT_s2_forward(B.get_ctx(), B.GI_F16, B.ptr(xsd), B.ptr(phase), B.ptr(out), n, H, W, ca, ca, cb, cb + 64, 1, 0,
                                        None, 0))
    torch.cuda.synchronize()
    assert torch.equal(from_nhwc(out[..., :cb].contiguous()), refT)
    assert float((out[..., cb:] - 7.0).abs().max()) == 0.0


def test_mfma_layout_exact_integers():
    """Asymmetric small-integer data: every product and sum is exact in fp16/fp32, so any fragment
    or C-layout mix-up shows as a non-zero error (cdna guide: 'A=I-check with asymmetric B')."""
    n, H, W, cb, ca = 1, 8, 8, 64, 128
    g = torch.Generator().manual_seed(11)
    x = torch.randint(-2, 3, (n, cb, H, W), generator=g).float()
    w = torch.randint(-2, 3, (ca, cb, 4, 4), generator=g).float()
    for code in (B.GI_F32, B.GI_F16):
        ref = F.conv2d(x, w, None, stride=2, padding=1)
        packed, phase = pack(w, code)
        out = torch.zeros((n, H // 2, W // 2, ca), dtype=tdt(code), device="cuda")
        xd = nhwc_dev(x, code)
        B.check(B.lib().gi_conv_s2_forward(B.get_ctx(), code, B.ptr(xd), B.ptr(packed), B.ptr(out), n, H, W, cb, cb, ca, ca, 0, 0,
                                           None, 0))
        assert torch.equal(from_nhwc(out), ref), f"conv exact-int mismatch dt={code}"
        xs = torch.randint(-2, 3, (n, ca, 4, 4), generator=g).float()
        refT = F.conv_transpose2d(xs, w, None, stride=2, padding=1)
        outT = torch.zeros((n, 8, 8, cb), dtype=tdt(code), device="cuda")
        xsd = nhwc_dev(xs, code)
        B.check(B.lib().gi_convT_s2_forward(B.get_ctx(), code, B.ptr(xsd), B.ptr(phase), B.ptr(outT), n, 4, 4, ca, ca, cb, cb, 0, 0,
                                            None, 0))
        assert torch.equal(from_nhwc(outT), refT), f"convT exact-int mismatch dt={code}"
        S = torch.randint(-2, 3, (2, ca, 4, 4), generator=g).float()
        L = torch.randint(-2, 3, (2, cb, 8, 8), generator=g).float()
        refW = torch.nn.grad.conv2d_weight(L, (ca, cb, 4, 4), S, stride=2, padding=1)
        dW = torch.zeros((ca, 4, 4, cb), dtype=torch.float32, device="cuda")
        Sd, Ld = nhwc_dev(S, code), nhwc_dev(L, code)   # keep alive until the kernel has run
        B.check(B.lib().gi_wgrad_s2(B.get_ctx(), code, B.ptr(Sd), B.ptr(Ld), B.ptr(dW), 2, 4, 4, ca, ca, cb, cb, 0, 1.0))
        assert torch.equal(dW.cpu().permute(0, 3, 1, 2), refW), f"wgrad exact-int mismatch dt={code}"
