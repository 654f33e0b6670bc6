"""Drop-in for the reference's lib/pytorch_ssim/__init__.py (`ssim()` :68-76, `SSIM` :42-66,
`gaussian` :10-12, `create_window` :14-18) on the fused HIP kernel `gi_ssim` (csrc/ssim.hip).

The reference evaluates the metric on detached tensors (experiment1_global_local_D.py:209,
experiment3_global_D.py:201); the kernel is forward-only and this module refuses inputs that
require grad instead of silently cutting the graph."""
from math import exp

import ctypes as C

import torch

from ... import backend as B


def gaussian(window_size, sigma):
    """1-D normalised Gaussian exactly as __init__.py:10-12 builds it (python-float exp, fp32 tensor, fp32 sum)."""
    gauss = torch.tensor([exp(-(x - window_size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(window_size)],
                         dtype=torch.float32)
    return gauss / gauss.sum()


def create_window(window_size, channel):
    """(channel,1,ws,ws) outer-product window (__init__.py:14-18). Kept for API compatibility; the
    kernel applies the 1-D window separably."""
    w1 = gaussian(window_size, 1.5).unsqueeze(1)
    w2 = w1.mm(w1.t()).float().unsqueeze(0).unsqueeze(0)
    return w2.expand(channel, 1, window_size, window_size).contiguous()


def _ssim(img1, img2, window_size, size_average):
    if img1.shape != img2.shape or img1.dim() != 4:
        raise ValueError("ssim: two (n,c,h,w) tensors of one shape expected, got %s and %s" % (tuple(img1.shape), tuple(img2.shape)))
    if img1.requires_grad or img2.requires_grad:
        raise B.BackendError("ssim is a forward-only metric here: detach() the inputs (as the reference's call sites do)")
    if not img1.is_cuda or img1.dtype != torch.float32 or img2.dtype != torch.float32 or img2.device != img1.device:
        raise B.BackendError("ssim takes float32 tensors on the gfx950 device")
    img1, img2 = img1.contiguous(), img2.contiguous()
    n, c, h, w = img1.shape
    lib, ctx = B.lib(), B.get_ctx(img1.device)
    ns = lib.gi_ssim_scratch_floats(n, c, h, w, window_size)
    if ns < 0:
        raise B.BackendError("ssim: unsupported size n=%d c=%d h=%d w=%d window_size=%d (odd window <= 31)" % (n, c, h, w, window_size))
    scratch = torch.empty((ns + 1) // 2, dtype=torch.float64, device=img1.device)
    win = gaussian(window_size, 1.5)
    hw = (C.c_float * window_size)(*win.tolist())
    if size_average:
        out = torch.empty(1, dtype=torch.float32, device=img1.device)
        B.check(lib.gi_ssim(ctx, B.ptr(img1), B.ptr(img2), n, c, h, w, window_size, C.cast(hw, C.c_void_p), None, B.ptr(out), B.ptr(scratch)))
        return out.view(())
    out = torch.empty(n, dtype=torch.float32, device=img1.device)
    B.check(lib.gi_ssim(ctx, B.ptr(img1), B.ptr(img2), n, c, h, w, window_size, C.cast(hw, C.c_void_p), B.ptr(out), None, B.ptr(scratch)))
    return out


class SSIM(torch.nn.Module):
    """__init__.py:42-66."""

    def __init__(self, window_size=11, size_average=True):
        super().__init__()
        self.window_size = window_size
        self.size_average = size_average

    def forward(self, img1, img2):
        return _ssim(img1, img2, self.window_size, self.size_average)


def ssim(img1, img2, window_size=11, size_average=True):
    """__init__.py:68-76."""
    return _ssim(img1, img2, window_size, size_average)
