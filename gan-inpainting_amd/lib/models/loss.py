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
