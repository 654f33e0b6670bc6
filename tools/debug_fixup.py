"""Split-K fix-up debugging: op-level conv / convT on many shapes, back-to-back launches sharing one scratch buffer."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import torch.nn.functional as F
import gan_inpainting_amd  # noqa
from gan_inpainting_amd import backend as B
from gpu_util import quant, nhwc_dev, from_nhwc, pack, tdt

def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale

ws = torch.full((16 << 20,), float("nan"), dtype=torch.float32, device="cuda")
for code in (B.GI_F32, B.GI_F16):
    for (n, H, W, cb, ca) in [(2, 64, 64, 64, 256), (2, 64, 64, 64, 128), (2, 32, 32, 128, 512), (2, 16, 16, 256, 1024), (2, 8, 8, 512, 1024),
                              (2, 64, 64, 64, 512), (4, 32, 32, 64, 256)]:
        errs = []
        for rep in range(3):
            x = quant(rnd((n, cb, H, W), 10 + rep), code); w = quant(rnd((ca, cb, 4, 4), 20 + rep, 0.05), code)
            ref = F.conv2d(x, w, None, stride=2, padding=1)
            packed, _ = pack(w, code)
            out = torch.full((n, H // 2, W // 2, ca), float("nan"), dtype=tdt(code), device="cuda")
            B.check(B.lib().gi_conv_s2_forward(B.get_ctx(), code, B.ptr(nhwc_dev(x, code)), B.ptr(packed), B.ptr(out), n, H, W, cb, cb, ca, ca, 0, 0,
                                               B.ptr(ws), ws.numel() * 4))
            torch.cuda.synchronize()
            errs.append(float((from_nhwc(out) - ref).abs().max() / ref.abs().max()))
        print("conv", code, (n, H, W, cb, ca), ["%.2e" % e for e in errs], flush=True)
    for (n, H, W, ca, cb) in [(2, 16, 16, 512, 128), (2, 8, 8, 1024, 256), (2, 32, 32, 256, 64), (2, 4, 4, 1024, 512)]:
        errs = []
        for rep in range(3):
            x = quant(rnd((n, ca, H, W), 30 + rep), code); w = quant(rnd((ca, cb, 4, 4), 40 + rep, 0.05), code)
            ref = F.conv_transpose2d(x, w, None, stride=2, padding=1)
            _, phase = pack(w, code)
            out = torch.full((n, 2 * H, 2 * W, cb), float("nan"), dtype=tdt(code), device="cuda")
            B.check(B.lib().gi_convT_s2_forward(B.get_ctx(), code, B.ptr(nhwc_dev(x, code)), B.ptr(phase), B.ptr(out), n, H, W, ca, ca, cb, cb, 0, 0,
                                                B.ptr(ws), ws.numel() * 4))
            torch.cuda.synchronize()
            errs.append(float((from_nhwc(out) - ref).abs().max() / ref.abs().max()))
        print("convT", code, (n, H, W, ca, cb), ["%.2e" % e for e in errs], flush=True)
