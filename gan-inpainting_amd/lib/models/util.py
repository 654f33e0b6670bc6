"""Drop-in for the reference's lib/models/util.py (AverageMeter :2-17, set_requires_grad :19-22,
count_parameters :24-25) plus the fused helpers the plugins use on the HIP backend."""
import torch

from ... import backend as B


class AverageMeter(object):
    """Computes and stores the average and current value (util.py:2-17)."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = 0
        self.avg = 0
        self.sum = 0
        self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


def set_requires_grad(nets, requires_grad):
    """util.py:19-22. A frozen net's backward produces input gradients only (no wgrad kernels)."""
    for net in nets:
        for param in net.parameters():
            param.requires_grad = requires_grad


def count_parameters(model):
    return sum(p.numel() for p in model.parameters() if p.requires_grad)


def clamp_parameters(net, lo=-0.01, hi=0.01):
    """WGAN weight clipping over every parameter of `net` (wgan_l1.py:151-153) as one kernel."""
    flat = net.flat_params()
    B.check(B.lib().gi_clamp(B.get_ctx(flat.device), B.ptr(flat), flat.numel(), float(lo), float(hi)))
    net.mark_dirty()


class GradFlow:
    """mean(|grad|) of every parameter whose name has no 'bias' (minimaxgan_l1.py:103-108,
    :180-182), computed by one kernel and read back with one copy instead of one .item() per tensor."""

    def __init__(self, net, scale=1.0):
        """scale: factor applied to the measured means - 1/world_size under data parallelism, where the flat gradient buffer
        holds the SUM over the ranks until the optimizer kernel divides (parallel.GradSync.grad_scale), so that the logged
        values are those of the averaged gradient a single-process run would log."""
        self.net = net
        self.scale = float(scale)
        inv = [t for t in net._inv if t["kind"] <= 1 and "bias" not in t["name"]]
        self.names = [t["name"] for t in inv]
        dev = net.device
        self.off = torch.tensor([t["offset"] for t in inv], dtype=torch.int64, device=dev)
        self.len = torch.tensor([t["numel"] for t in inv], dtype=torch.int64, device=dev)
        self.out = torch.zeros(len(inv), dtype=torch.float32, device=dev)

    def measure(self):
        """Device tensor of per-parameter means (no host sync)."""
        g = self.net.flat_grads()
        B.check(B.lib().gi_grad_absmean(B.get_ctx(g.device), B.ptr(g), B.ptr(self.off), B.ptr(self.len), len(self.names),
                                        B.ptr(self.out)))
        return self.out if self.scale == 1.0 else self.out * self.scale

    def as_dict(self):
        vals = self.measure().tolist()
        return dict(zip(self.names, vals))
