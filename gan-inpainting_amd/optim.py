"""Fused optimizers with the torch.optim call surface the plugins use:
    optim.Adam(net.parameters(), lr=0.0002, betas=(0.5, 0.999))     minimaxgan_l1.py:64-65
    optim.RMSprop(net.parameters(), lr=0.00005)                     wgan_l1.py:64-65
One kernel per network over its flat fp32 parameter / gradient / state buffers."""
import itertools

import torch

from . import backend as B


class _FlatOptimizer:
    def __init__(self, params):
        params = list(params)
        if not params:
            raise ValueError("optimizer got an empty parameter list")
        owners = []
        for p in params:
            owner = getattr(p, "_gi_owner", None)
            if owner is None:
                raise B.BackendError("parameter does not belong to a HIP-backend network on a gfx950 device "
                                     "(move the network with .to('cuda') before building the optimizer)")
            if all(owner is not o for o in owners):
                owners.append(owner)
        for o in owners:
            mine = sum(1 for p in params if getattr(p, "_gi_owner", None) is o)
            if mine != sum(1 for _ in o.parameters()):
                raise B.BackendError("the fused optimizers update whole networks: pass all of net.parameters()")
        self.nets = owners
        self.param_groups = [dict(params=params)]
        self.grad_scale = 1.0   # multiplies every gradient (e.g. 1/world_size after a SUM all-reduce)
        # fp16 overflow guard: scan the gradients before every update and skip the whole update when they hold
        # inf/NaN (like torch.cuda.amp.GradScaler.step). On by default for networks computing in fp16.
        self.guard = any(getattr(n, "_dtype", None) == B.GI_F16 for n in owners)
        self._flags = None
        self._seen = 0

    def _guard_ptr(self):
        """Run the finite check over the gradients of EVERY network of this optimizer into one shared verdict (all of
        them skip the update or none does, and a skipped update counts once whatever the number of networks); returns
        (device flag pointer, scan word) - (None, 0) when the guard is off. No finish launch: the update's first
        optimizer kernel records the verdict (gi_*_step_scan, include/ganinpaint.h); updates alternate between two scan words."""
        if not self.guard:
            return None, 0
        lib = B.lib()
        if self._flags is None:
            self._flags = torch.zeros(4, dtype=torch.int32, device=self.nets[0].flat_params().device)
            self._word = 3
        self._word = 5 - self._word      # 2, 3, 2, ...
        for net in self.nets:
            g = net.flat_grads()
            B.check(lib.gi_check_finite_scan_word(B.get_ctx(g.device), B.ptr(g), g.numel(), B.ptr(self._flags), self._word))
        return B.ptr(self._flags), self._word

    def poll_skipped(self):
        """Number of updates skipped since the last poll (one device read-back: call it at logging cadence)."""
        if not self.guard or self._flags is None:
            return 0
        total = int(self._flags[0].item())
        new, self._seen = total - self._seen, total
        return new

    def zero_grad(self, set_to_none=False):
        for n in self.nets:
            n.zero_grad()

    def _done(self):
        for n in self.nets:
            n.mark_dirty()


class Adam(_FlatOptimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params)
        self.lr, self.betas, self.eps = lr, betas, eps
        self.state = [dict(m=torch.zeros_like(n.flat_params()), v=torch.zeros_like(n.flat_params())) for n in self.nets]
        self.t = 0

    @torch.no_grad()
    def step(self):
        self.t += 1
        lib = B.lib()
        guard, word = self._guard_ptr()
        for k, (n, st) in enumerate(zip(self.nets, self.state)):
            p, g = n.flat_params(), n.flat_grads()
            # self.t still counts the updates skipped since the last poll_skipped(); the kernel subtracts them (device count - _seen)
            B.check(lib.gi_adam_step_scan(B.get_ctx(p.device), B.ptr(p), B.ptr(g), B.ptr(st["m"]), B.ptr(st["v"]), p.numel(),
                                          self.lr, self.betas[0], self.betas[1], self.eps, self.t, self._seen if guard else -1,
                                          self.grad_scale, guard, word, 1 if k == 0 else 0))
        self._done()

    def poll_skipped(self):
        new = super().poll_skipped()
        self.t -= new     # one per skipped UPDATE (shared verdict): skipped updates do not advance the bias correction (the kernel
        #                   already left them out - device count minus _seen - so nothing changes for the updates in between)
        return new


class RMSprop(_FlatOptimizer):
    """torch.optim.RMSprop defaults (alpha=0.99, eps=1e-8, no momentum, not centered). `clamp` > 0
    fuses the WGAN weight clipping p.data.clamp_(-clamp, clamp) into the same pass."""

    def __init__(self, params, lr=1e-2, alpha=0.99, eps=1e-8, clamp=0.0):
        super().__init__(params)
        self.lr, self.alpha, self.eps, self.clamp = lr, alpha, eps, clamp
        self.state = [dict(sq=torch.zeros_like(n.flat_params())) for n in self.nets]

    @torch.no_grad()
    def step(self):
        lib = B.lib()
        guard, word = self._guard_ptr()
        for k, (n, st) in enumerate(zip(self.nets, self.state)):
            p, g = n.flat_params(), n.flat_grads()
            B.check(lib.gi_rmsprop_step_scan(B.get_ctx(p.device), B.ptr(p), B.ptr(g), B.ptr(st["sq"]), p.numel(), self.lr,
                                             self.alpha, self.eps, self.clamp, self.grad_scale, guard, word, 1 if k == 0 else 0))
        self._done()


def chain(*its):
    return itertools.chain(*its)
