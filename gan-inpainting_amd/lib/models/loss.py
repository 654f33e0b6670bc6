"""Drop-in for the losses of the reference's lib/models/loss.py (RMSELoss :11-19, LocalLoss :24-47)
and the torch built-ins the plugins call (nn.L1Loss, nn.MSELoss, nn.BCELoss, torch.mean), as fused
HIP reductions with their gradients. The module-level VGG wrapper of loss.py:4-5 (perceptual / style
losses) is not part of this path."""
import torch
from torch import nn

from ... import backend as B


def _scratch(dev):
    return torch.empty(4096, dtype=torch.float32, device=dev)


class _ReconFn(torch.autograd.Function):
    """loss(a, b[, mask]); gradient w.r.t. `a` only (b is the ground truth in every call site)."""

    @staticmethod
    def forward(ctx, a, b, mask, kind, eps):
        a, b = a.contiguous(), b.contiguous()
        if a.dtype != torch.float32 or not a.is_cuda:
            raise B.BackendError("HIP losses take float32 tensors on the gfx950 device")
        lib, c = B.lib(), B.get_ctx(a.device)
        out = torch.empty(1, dtype=torch.float32, device=a.device)
        grad = torch.empty_like(a) if a.requires_grad else None
        scr = _scratch(a.device)
        n = a.numel()
        if kind == "l1":
            B.check(lib.gi_loss_l1(c, B.ptr(a), B.ptr(b), n, B.ptr(out), B.ptr(grad), 1.0, B.ptr(scr)))
        elif kind == "mse":
            B.check(lib.gi_loss_mse(c, B.ptr(a), B.ptr(b), n, B.ptr(out), B.ptr(grad), 1.0, B.ptr(scr)))
        elif kind == "rmse":
            B.check(lib.gi_loss_rmse(c, B.ptr(a), B.ptr(b), n, eps, B.ptr(out), B.ptr(grad), 1.0, B.ptr(scr)))
        else:
            m = mask.contiguous()
            code = {"local_l1": 0, "local_mse": 1, "local_rmse": 2}[kind]
            B.check(lib.gi_loss_local(c, B.ptr(a), B.ptr(b), B.ptr(m), n, code, B.ptr(out), B.ptr(grad), 1.0, B.ptr(scr)))
        ctx.grad = grad
        return out.view(())

    @staticmethod
    def backward(ctx, g):
        return (ctx.grad * g if ctx.grad is not None else None), None, None, None, None


class L1Loss(nn.Module):
    """nn.L1Loss() as called in minimaxgan_l1.py:62,166: l1(ground, inpainted). The gradient flows
    to whichever argument requires it (the loss is symmetric)."""

    def forward(self, a, b):
        if b.requires_grad and not a.requires_grad:
            a, b = b, a
        return _ReconFn.apply(a, b, None, "l1", 0.0)


class MSELoss(nn.Module):
    def forward(self, a, b):
        if b.requires_grad and not a.requires_grad:
            a, b = b, a
        return _ReconFn.apply(a, b, None, "mse", 0.0)


class RMSELoss(nn.Module):
    """loss.py:11-19: sqrt(mse(yhat, y) + eps), eps = 1e-16."""

    def __init__(self, eps=1e-16):
        super().__init__()
        self.eps = eps

    def forward(self, yhat, y):
        if y.requires_grad and not yhat.requires_grad:
            yhat, y = y, yhat
        return _ReconFn.apply(yhat, y, None, "rmse", self.eps)


class LocalLoss(nn.Module):
    """loss.py:24-47: sum(base(y*m, yhat*m)) / count(m != 0). As in the reference the constructor
    takes the loss CLASS (nn.L1Loss / nn.MSELoss, or their HIP twins); the sqrt branch of the
    reference never executes (its isinstance test is on the replaced member), so passing RMSELoss
    selects the evident intent, sqrt(masked mean square + eps), as a labelled extension."""

    def __init__(self, baseloss, eps=1e-16):
        super().__init__()
        name = getattr(baseloss, "__name__", type(baseloss).__name__)
        self.kind = {"L1Loss": "local_l1", "MSELoss": "local_mse", "RMSELoss": "local_rmse"}[name]
        self.eps = eps

    def forward(self, yhat, y, mask):
        return _ReconFn.apply(yhat, y, mask, self.kind, self.eps)


class _AdvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, kind, target):
        p = pred.contiguous().view(-1)
        out = torch.empty(1, dtype=torch.float32, device=p.device)
        grad = torch.empty_like(p) if pred.requires_grad else None
        B.check(B.lib().gi_loss_adv(B.get_ctx(p.device), B.ptr(p), p.numel(), kind, float(target), B.ptr(out), B.ptr(grad), 1.0))
        ctx.grad, ctx.shape = grad, pred.shape
        return out.view(())

    @staticmethod
    def backward(ctx, g):
        return (ctx.grad.view(ctx.shape) * g if ctx.grad is not None else None), None, None


def _const_target(target):
    """The plugins always compare against torch.ones(n) / torch.zeros(n); take the constant."""
    if torch.is_tensor(target):
        t0 = float(target.reshape(-1)[0])
        return t0
    return float(target)


class BCELoss(nn.Module):
    """nn.BCELoss() against an all-ones / all-zeros target (minimaxgan_l1.py:135,141,162)."""

    def forward(self, pred, target):
        return _AdvFn.apply(pred, 0, _const_target(target))


class LSGANLoss(nn.Module):
    """nn.MSELoss() on (n,) critic outputs against a constant (experiment1_global_local_D.py:162)."""

    def forward(self, pred, target):
        return _AdvFn.apply(pred, 1, _const_target(target))


def critic_mean(pred):
    """torch.mean(d_pred).view(1) of the WGAN plugins (wgan_l1.py:137-143,177)."""
    return _AdvFn.apply(pred, 2, 0.0).view(1)


# ---- config-5 extras: perceptual / style / TV / weighted cross entropy ------------
vgg = None   # the reference builds VGG19Wrapper().to(device) at import (loss.py:4-5); here on first use or via set_vgg


def set_vgg(wrapper):
    """Install the feature network (e.g. after load_state_dict of torchvision's pretrained vgg19)."""
    global vgg
    vgg = wrapper
    return wrapper


def _vgg_for(x):
    global vgg
    if vgg is None:
        import warnings
        from . import networks
        warnings.warn("perceptual/style loss: no VGG-19 weights installed (loss.set_vgg); using a random He-initialised "
                      "feature network - the pretrained download of the reference (networks.py:371) is not available here")
        vgg = networks.VGG19Wrapper(max_pairs=max(8, x.shape[0])).to(x.device)
    return vgg


def perceptual_and_style_loss(output, target, weight_p=0.05, weight_s=100):
    """loss.py:93-115. Constants w.r.t. the generator, exactly like the reference (no_grad + detach)."""
    return _vgg_for(output).perceptual_and_style(output, target, weight_p, weight_s)


def perceptual_loss(output, target, weight=0.05):
    """loss.py:50-69."""
    return _vgg_for(output).perceptual_and_style(output, target, weight, 0.0)[0]


def style_loss(output, target, weight=0.1):
    """loss.py:71-90."""
    return _vgg_for(output).perceptual_and_style(output, target, 0.0, weight)[1]


class _TVFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, tv_weight):
        x = img.contiguous()
        if x.dtype != torch.float32 or not x.is_cuda or x.dim() != 4:
            raise B.BackendError("tv_loss takes a (n,c,H,W) float32 tensor on the gfx950 device")
        n, c, h, w = x.shape
        out = torch.empty(1, dtype=torch.float32, device=x.device)
        grad = torch.empty_like(x) if img.requires_grad else None
        scr = torch.empty(2048, dtype=torch.float64, device=x.device)
        B.check(B.lib().gi_loss_tv(B.get_ctx(x.device), B.ptr(x), n * c, h, w, float(tv_weight), B.ptr(out), B.ptr(grad), 1.0, B.ptr(scr)))
        ctx.grad = grad
        return out.view(())

    @staticmethod
    def backward(ctx, g):
        return (ctx.grad * g if ctx.grad is not None else None), None


def tv_loss(img, tv_weight):
    """loss.py:138-151, with its gradient (the only config-5 extra besides the face-parsing term that
    reaches the generator)."""
    return _TVFn.apply(img, tv_weight)


class _CEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, weight, ignore_index):
        z = logits.contiguous()
        y = labels.contiguous()
        if z.dtype != torch.float32 or not z.is_cuda or y.dtype != torch.int64 or y.device != z.device:
            raise B.BackendError("CrossEntropyLoss takes float32 logits and int64 labels on the gfx950 device")
        n, k = z.shape[0], z.shape[1]
        hw = z.numel() // (n * k)
        if y.numel() != n * hw:
            raise ValueError("labels %s do not match logits %s" % (tuple(y.shape), tuple(z.shape)))
        import ctypes as C
        hwt = (C.c_float * k)(*[float(v) for v in weight]) if weight is not None else None
        out = torch.empty(2, dtype=torch.float32, device=z.device)
        grad = torch.empty_like(z) if logits.requires_grad else None
        scr = torch.empty(2048, dtype=torch.float64, device=z.device)
        B.check(B.lib().gi_loss_cross_entropy(B.get_ctx(z.device), B.ptr(z), B.ptr(y), n, k, hw,
                                              C.cast(hwt, C.c_void_p) if hwt is not None else None, int(ignore_index),
                                              B.ptr(out), B.ptr(grad), 1.0, B.ptr(scr)))
        ctx.grad = grad
        return out[0]

    @staticmethod
    def backward(ctx, g):
        return (ctx.grad * g if ctx.grad is not None else None), None, None, None


class CrossEntropyLoss(nn.Module):
    """nn.CrossEntropyLoss(weight=w) on (n,K,H,W) logits / (n,H,W) labels
    (wgan_perceptual_style_faceparsing.py:67-68,212-213), K in {2,3,4,8,16}."""

    def __init__(self, weight=None, ignore_index=-100):
        super().__init__()
        self.weight = None if weight is None else [float(v) for v in torch.as_tensor(weight).tolist()]
        self.ignore_index = ignore_index

    def forward(self, logits, labels):
        return _CEFn.apply(logits, labels, self.weight, self.ignore_index)
