"""Per-batch step schedules of the reference's training loops, driven straight through the C-ABI
(no autograd graph, preallocated buffers, no host synchronisation inside a step):

    MinimaxStep   experiment_list/minimaxgan_l1.py:110-173 (recon='rmse': minimaxgan_rmse.py)
    WGANStep      experiment_list/wgan_l1.py:110-186 / wgan_rmse.py  (weight clipping :151-153)
    DualDStep     experiment_list/experiment1_global_local_D.py:139-200

Scalar losses stay on the device (1-element tensors); the plugins read them back at their own
logging cadence. With a GradSync (parallel.py) every optimizer step is preceded by a bucketed
all-reduce of that network's flat gradient buffer.
"""
import torch

from . import backend as B
from . import optim
from .lib.models import util


class _Ops:
    """Thin allocation-free wrappers over the elementwise / loss entry points."""

    def __init__(self, device):
        self.dev = torch.device(device)
        self.scratch = torch.empty(2048, dtype=torch.float64, device=self.dev).view(torch.float32)   # 4096 floats, 8-byte aligned

    @property
    def ctx(self):
        return B.get_ctx(self.dev)

    def mask_apply(self, ground, mask, mask_c, masked, do_ceil):
        B.check(B.lib().gi_mask_apply(self.ctx, B.ptr(ground), B.ptr(mask), B.ptr(mask_c), B.ptr(masked), ground.numel(),
                                      1 if do_ceil else 0))

    def composite(self, masked, gen, mask_c, out):
        B.check(B.lib().gi_mask_composite(self.ctx, B.ptr(masked), B.ptr(gen), B.ptr(mask_c), B.ptr(out), out.numel()))

    def mul(self, a, b, out):
        B.check(B.lib().gi_mul(self.ctx, B.ptr(a), B.ptr(b), B.ptr(out), out.numel()))

    def add(self, a, b, out, alpha=1.0):
        B.check(B.lib().gi_add(self.ctx, B.ptr(a), B.ptr(b), B.ptr(out), out.numel(), float(alpha)))

    def adv(self, pred, kind, target, loss_out, grad, gscale=1.0):
        B.check(B.lib().gi_loss_adv(self.ctx, B.ptr(pred), pred.numel(), kind, float(target), B.ptr(loss_out), B.ptr(grad),
                                    float(gscale)))

    def adv_pair(self, pred2, n, kind, t_a, t_b, loss_a, loss_b, grad2, gs_a=1.0, gs_b=1.0):
        """adv() of both halves of a stacked [a | b] prediction vector in one launch (gi_loss_adv_pair)."""
        if not hasattr(B.lib(), "gi_loss_adv_pair"):      # an older build loaded through GI_LIB_PATH for an A/B run
            self.adv(pred2[:n], kind, t_a, loss_a, grad2[:n], gs_a)
            self.adv(pred2[n:], kind, t_b, loss_b, grad2[n:], gs_b)
            return
        B.check(B.lib().gi_loss_adv_pair(self.ctx, B.ptr(pred2), int(n), kind, float(t_a), float(t_b), B.ptr(loss_a), B.ptr(loss_b),
                                         B.ptr(grad2), float(gs_a), float(gs_b)))

    def recon(self, kind, a, b, loss_out, grad, gscale=1.0):
        lib = B.lib()
        if kind == "l1":
            B.check(lib.gi_loss_l1(self.ctx, B.ptr(a), B.ptr(b), a.numel(), B.ptr(loss_out), B.ptr(grad), float(gscale), B.ptr(self.scratch)))
        elif kind == "rmse":
            B.check(lib.gi_loss_rmse(self.ctx, B.ptr(a), B.ptr(b), a.numel(), 1e-16, B.ptr(loss_out), B.ptr(grad), float(gscale),
                                     B.ptr(self.scratch)))
        else:
            raise ValueError(kind)

    def local(self, kind, yhat, y, mask, loss_out, grad, gscale=1.0):
        """LocalLoss (loss.py:24-47): kind 'l1' | 'mse' | 'rmse' (the sqrt extension); gradient w.r.t. yhat."""
        code = {"l1": 0, "mse": 1, "rmse": 2}[kind]
        B.check(B.lib().gi_loss_local(self.ctx, B.ptr(yhat), B.ptr(y), B.ptr(mask), yhat.numel(), code, B.ptr(loss_out), B.ptr(grad),
                                      float(gscale), B.ptr(self.scratch)))

    def tv(self, img, weight, loss_out, grad, gscale=1.0):
        n, c, h, w = img.shape
        B.check(B.lib().gi_loss_tv(self.ctx, B.ptr(img), n * c, h, w, float(weight), B.ptr(loss_out), B.ptr(grad), float(gscale),
                                   B.ptr(self.scratch)))

    def cross_entropy(self, logits, labels, class_weight, loss_out2, grad, gscale=1.0):
        import ctypes as C
        n, k = logits.shape[0], logits.shape[1]
        hw = logits.numel() // (n * k)
        cw = (C.c_float * k)(*[float(v) for v in class_weight]) if class_weight is not None else None
        B.check(B.lib().gi_loss_cross_entropy(self.ctx, B.ptr(logits), B.ptr(labels), n, k, hw, C.cast(cw, C.c_void_p) if cw is not None else None,
                                              -100, B.ptr(loss_out2), B.ptr(grad), float(gscale), B.ptr(self.scratch)))


BCE, LSGAN, MEAN = 0, 1, 2


class _StepBase:
    recon_weight = 1.0      # largest multiplier on a mean-reduced reconstruction loss (lambda)
    auto_loss_scale = True  # fp16 only: pick per-network loss scales from the batch geometry

    def __init__(self, net_G, nets_D, device, sync=None):
        self.G, self.Ds = net_G, list(nets_D)
        self.ops = _Ops(device)
        self.sync = sync
        self._shape = None
        for n in [net_G] + self.Ds:
            n.train()
            # inside a step object parameters only change through the package's optimizers / clamp, which
            # mark the network dirty: re-derive packed weights once per update instead of once per forward
            n.always_sync = False
        if sync is not None:
            sync.prepare(net_G.device)
            sync.broadcast_parameters([net_G] + self.Ds)    # replicas start identical whatever the ranks' RNG state
            if hasattr(net_G, "set_dropout_seed"):          # ... but draw their own dropout masks
                net_G.set_dropout_seed(0x5EED0000 + 1000 * sync.rank)

    def _bind_optimizers(self, *opts):
        self._opts = opts
        if self.sync is not None:
            for o in opts:
                o.grad_scale = self.sync.grad_scale()   # SUM all-reduce -> mean gradient

    def _buffers(self, ground):
        if self._shape != tuple(ground.shape):
            self._shape = tuple(ground.shape)
            mk = lambda: torch.empty_like(ground)  # noqa: E731
            self.mask_c, self.masked, self.inpainted = mk(), mk(), mk()
            self.g_adv, self.g_rec, self.g_gen, self.tmp1, self.tmp2 = mk(), mk(), mk(), mk(), mk()
            n = ground.shape[0]
            self.dpred = torch.empty((n, 1), dtype=torch.float32, device=ground.device)
            self.L = {}
            if self.auto_loss_scale:
                self._pick_loss_scales(ground)

    def _pick_loss_scales(self, ground):
        """fp16 backward: the gradient entering the generator is ~ lambda / (N*H*W) per pixel (mean-
        reduced L1/RMSE), the one entering a discriminator ~ 1/N per sample. Scale each network so
        these land around 2^-5 in fp16 (normal range 6e-5 .. 65504); the library removes the scale
        when it writes the fp32 parameter gradients / the input gradient."""
        import math
        n, _, h, w = ground.shape
        if self.G._dtype == B.GI_F16:
            g0 = self.recon_weight / float(n * h * w)
            self.G.set_loss_scale(min(65536.0, max(1.0, 2.0 ** round(math.log2(0.03 / g0)))))
        for d in self.Ds:
            if d._dtype == B.GI_F16:
                d.set_loss_scale(min(65536.0, 64.0 * n))

    # ---- stream contract for readers of the step's results (plugins: loss accumulation, gradient-flow statistics) ----
    def side_stream(self):
        """The HIP stream the discriminator side of the step runs on, or None when everything is on the caller's."""
        return None

    def side_keys(self):
        """Keys of the loss dict whose values (and the discriminators' gradient buffers) are produced on side_stream()."""
        return ()

    def sync_for_logging(self):
        """Make the caller's stream wait for everything the step has issued on its side stream, so that losses,
        discriminator gradients and overflow flags can be read from the caller's stream."""

    def poll_overflow(self, logger=None):
        """fp16 only, call at logging cadence: when an optimizer skipped updates because its gradients held
        inf/NaN, halve the loss scale of its networks (they re-derive it from the batch geometry otherwise)."""
        self.sync_for_logging()
        out = 0
        for opt in getattr(self, "_opts", ()):
            new = opt.poll_skipped()
            if new:
                out += new
                for net in opt.nets:
                    if net._dtype == B.GI_F16:
                        net.set_loss_scale(max(1.0, net._loss_scale / 2.0))
                        if logger:
                            logger.info("fp16 overflow: %d update(s) of %s skipped, loss scale -> %g", new, type(net).__name__, net._loss_scale)
        return out

    def _loss(self, name):
        if name not in self.L:
            self.L[name] = torch.zeros(1, dtype=torch.float32, device=self.ops.dev)
        return self.L[name]

    def _reduce(self, net):
        """Start the all-reduce of a network's whole gradient buffer (no overlap with its backward); _step waits for it."""
        if self.sync is not None:
            self.sync.launch(net.flat_grads())

    def _step(self, opt):
        """Optimizer update of the networks of `opt`. The gradient exchange is awaited HERE, as late as possible: whatever
        the stream issued between the backward and this call (on critic-only batches of the overlapped WGAN step: nothing
        on the critic's stream, while the main stream is already in the next batch's generator forward) overlaps it."""
        if self.sync is not None and self.sync.world > 1:
            for net in opt.nets:
                flat = net.flat_grads()
                self.sync.wait(flat.device if flat.is_cuda else None, flat=flat)
        opt.step()

    def _bwd_G(self, gtok, dy):
        """Generator backward; with a GradSync the gradients start their all-reduce in completion order while the
        rest of the backward is still being computed: decoder, then the inner encoder levels (d7 .. d5, 82 % of the
        encoder's gradients), then the outer ones."""
        if self.sync is None or self.sync.world == 1:
            return self._bwd(self.G, gtok, dy, False, True)
        G, lib = self.G, B.lib()
        if G._slot_gen[gtok[0]] != gtok[1]:
            raise B.BackendError("generator activations were overwritten before their backward")
        flat = G.flat_grads()
        split, split2 = lib.gi_net_phase_split(G._handle), lib.gi_net_phase_split2(G._handle)
        B.check(lib.gi_net_backward_phase(G._handle, gtok[0], B.ptr(dy), None, 1, 1))
        self.sync.launch(flat, split, flat.numel())
        B.check(lib.gi_net_backward_phase(G._handle, gtok[0], B.ptr(dy), None, 1, 3))
        self.sync.launch(flat, split2, split)
        B.check(lib.gi_net_backward_phase(G._handle, gtok[0], B.ptr(dy), None, 1, 4))
        self.sync.launch(flat, 0, split2)       # awaited by _step(optG)

    def _d_pair(self, net, real, fake, kind, t_real, t_fake, name_real, name_fake, gs_real=1.0, gs_fake=1.0, synced=False, x2=None,
                real_done=False):
        """A discriminator's two calls of a batch, D(real) and D(fake), as ONE [real | fake] batch whose BatchNorm
        layers are evaluated per half (gi_net_set_bn_groups): per image the reference's arithmetic, half the
        launches, one weight-gradient GEMM over both halves. Losses go to L[name_*], gradients accumulate."""
        o, n = self.ops, real.shape[0]
        bufs = self.__dict__.setdefault("_pair_bufs", {})
        key = (id(net), tuple(real.shape))
        if key not in bufs:
            bufs[key] = (torch.empty((2 * n,) + tuple(real.shape[1:]), dtype=torch.float32, device=real.device),
                         torch.empty((2 * n, 1), dtype=torch.float32, device=real.device))
        own, dp2 = bufs[key]
        if x2 is None:
            x2 = own
        if not real_done:                             # (the caller may have copied the real half already, ahead of a wait)
            x2[:n].copy_(real)
        if fake.data_ptr() != x2[n:].data_ptr():      # the caller may have produced `fake` in the pair buffer already
            x2[n:].copy_(fake)
        p, t = self._fwd(net, x2, bn_groups=2)
        o.adv_pair(p, n, kind, t_real, t_fake, self._loss(name_real), self._loss(name_fake), dp2, gs_real, gs_fake)
        if synced:
            self._bwd_D_synced(net, t, dp2)
        else:
            self._bwd(net, t, dp2, False, True)

    def _bwd_D_synced(self, net, tok, dy):
        """Critic backward + gradient all-reduce: the conv4 block and the head (8.5 of 11 MB) are reduced while
        conv3 .. conv1 are still in their backward; the optimizer step (_step) waits for both ranges."""
        lib = B.lib()
        if net._slot_gen[tok[0]] != tok[1]:
            raise B.BackendError("critic activations were overwritten before their backward")
        dy = dy.contiguous()
        if net._slot_groups[tok[0]] != net._groups_set:
            B.check(lib.gi_net_set_bn_groups(net._handle, net._slot_groups[tok[0]]))
            net._groups_set = net._slot_groups[tok[0]]
        flat = net.flat_grads()
        split = lib.gi_net_phase_split(net._handle)
        B.check(lib.gi_net_backward_phase(net._handle, tok[0], B.ptr(dy), None, 1, 1))
        self.sync.launch(flat, split, flat.numel())
        B.check(lib.gi_net_backward_phase(net._handle, tok[0], B.ptr(dy), None, 1, 2))
        self.sync.launch(flat, 0, split)        # awaited by _step(optD)

    @staticmethod
    def _fwd(net, x, bn_groups=1):
        y, slot, gen = net._forward_raw(x, bn_groups) if bn_groups != 1 else net._forward_raw(x)
        return y, (slot, gen)

    @staticmethod
    def _bwd(net, tok, dy, need_dx, need_wgrad):
        return net._backward_raw(tok[0], tok[1], dy, need_dx, need_wgrad)


class MinimaxStep(_StepBase):
    def __init__(self, net_G, net_D, opt_G, opt_D, recon="l1", sync=None, stacked=True):
        super().__init__(net_G, [net_D], net_G.device, sync)
        self.D, self.optG, self.optD, self.recon = net_D, opt_G, opt_D, recon
        self.stacked = stacked      # D(ground) | D(inpainted) as one batch with per-half BatchNorm (see _d_pair)
        self._bind_optimizers(opt_G, opt_D)

    @torch.no_grad()
    def __call__(self, ground, mask):
        o = self.ops
        self._buffers(ground)
        o.mask_apply(ground, mask, self.mask_c, self.masked, True)             # :114-117
        gen, gtok = self._fwd(self.G, self.masked)                              # :119
        o.composite(self.masked, gen, self.mask_c, self.inpainted)             # :122
        # ---- D step :129-148
        self.optD.zero_grad()
        if self.stacked:
            self._d_pair(self.D, ground, self.inpainted, BCE, 1.0, 0.0, "d_loss_real", "d_loss_fake")
        else:
            p, t = self._fwd(self.D, ground)
            o.adv(p, BCE, 1.0, self._loss("d_loss_real"), self.dpred)
            self._bwd(self.D, t, self.dpred, False, True)
            p, t = self._fwd(self.D, self.inpainted)
            o.adv(p, BCE, 0.0, self._loss("d_loss_fake"), self.dpred)
            self._bwd(self.D, t, self.dpred, False, True)
        self._reduce(self.D)
        self._step(self.optD)
        # ---- G step :154-173 (D frozen: input gradient only)
        self.optG.zero_grad()
        p, t = self._fwd(self.D, self.inpainted)
        o.adv(p, BCE, 1.0, self._loss("g_adv"), self.dpred)
        d_adv = self._bwd(self.D, t, self.dpred, True, False)
        o.recon(self.recon, self.inpainted, ground, self._loss("recon"), self.g_rec)
        o.add(d_adv, self.g_rec, self.tmp1)
        o.mul(self.tmp1, self.mask_c, self.g_gen)                              # d(inpainted)/d(gen) = mask
        self._bwd_G(gtok, self.g_gen)
        self._step(self.optG)
        return self.L


class WGANStep(_StepBase):
    """clip > 0 applies the reference's weight clipping after the critic update. If opt_D is this
    package's RMSprop its `clamp` is set so the clip is fused into the optimizer kernel."""

    def __init__(self, net_G, net_D, opt_G, opt_D, recon="l1", clip=0.01, sync=None, gp_lambda=0.0, overlap=False, stacked=True):
        super().__init__(net_G, [net_D], net_G.device, sync)
        self.D, self.optG, self.optD, self.recon, self.clip = net_D, opt_G, opt_D, recon, clip
        # stacked=True: the critic's two calls of a batch, D(ground) and D(inpainted) (wgan_l1.py:134-135), run as ONE
        # 2n batch with independent BatchNorm statistics per half (gi_net_set_bn_groups): same arithmetic per
        # image, half the launches, and one weight-gradient GEMM over both halves instead of two accumulating ones.
        self.stacked = stacked
        # overlap=True: the critic runs on a side HIP stream. Its real-image forward/backward does not depend
        # on the generator forward, and in critic-only batches the next batch's generator forward does not
        # depend on this batch's critic update, so the two networks' small / latency-bound kernels fill each
        # other's idle CUs. Results are identical (same kernels, same order per network).
        self.overlap = overlap
        if overlap:
            self._sD = torch.cuda.Stream(device=net_G.device)
            net_D._release_handle()          # the handle launches on the stream that is current when it is created
            self._inp2 = [None, None]
            self._evD = [None, None]
            self._k = 0
        # True: `ground` / `mask` were produced before the previous step was issued (resident in HBM, the
        # benchmark's contract), so the side stream need not wait for the main stream's pending work (the
        # previous batch's generator backward) before the critic's real-image pass.
        self.inputs_resident = False
        # gp_lambda > 0: WGAN-GP extension (BASELINE config 2) instead of the reference's weight clipping
        self.gp_lambda = gp_lambda
        if gp_lambda > 0:
            self.clip = clip = 0.0
        self._fused_clip = isinstance(opt_D, optim.RMSprop)
        if self._fused_clip:
            opt_D.clamp = clip
        self._bind_optimizers(opt_G, opt_D)

    @torch.no_grad()
    def _call_overlapped(self, ground, mask, update_g):
        o = self.ops
        self._d_synced = False
        self._buffers(ground)
        main = torch.cuda.current_stream(ground.device)
        sD = self._sD
        k = self._k
        self._k ^= 1
        if self._inp2[k] is None or self._inp2[k].shape[0] != 2 * ground.shape[0] or self._inp2[k].shape[1:] != ground.shape[1:]:
            # [ground | inpainted] of the stacked critic pass: the composite writes its half in place
            self._inp2[k] = torch.empty((2 * ground.shape[0],) + tuple(ground.shape[1:]), dtype=ground.dtype, device=ground.device)
        inp = self._inp2[k][ground.shape[0]:]
        # the side stream reads the caller's `ground` (critic real half, gradient penalty): tell the caching allocator,
        # or a per-batch `ground` freed by the caller could be handed to the next batch's H2D copy while the critic
        # still reads it (critic-only batches never make the main stream wait for the side stream)
        ground.record_stream(sD)
        if self._evD[k] is not None:
            main.wait_event(self._evD[k])        # the critic pass that read this buffer two batches ago is done
        e0 = main.record_event()
        o.mask_apply(ground, mask, self.mask_c, self.masked, True)
        gen, gtok = self._fwd(self.G, self.masked)
        o.composite(self.masked, gen, self.mask_c, inp)
        e_inp = main.record_event()
        self.inpainted = inp
        d_adv = None
        with torch.cuda.stream(sD):
            if not self.inputs_resident:
                sD.wait_event(e0)
            self.optD.zero_grad()
            if self.stacked:
                self._inp2[k][:ground.shape[0]].copy_(ground)   # the real half of the pair: before the wait for the inpainted one
                sD.wait_event(e_inp)
                self._critic_stacked(ground, inp, x2=self._inp2[k], real_done=True)
            else:
                pr, tr = self._fwd(self.D, ground)
                o.adv(pr, MEAN, 0.0, self._loss("d_loss_real"), self.dpred, +1.0)
                self._bwd(self.D, tr, self.dpred, False, True)
                sD.wait_event(e_inp)
                pf, tf = self._fwd(self.D, inp)
                o.adv(pf, MEAN, 0.0, self._loss("d_loss_fake"), self.dpred, -1.0)
                self._bwd(self.D, tf, self.dpred, False, True)
            if self.gp_lambda > 0:
                self.L["gp"] = self.D.gradient_penalty(ground, inp, getattr(self, "gp_eps", None), self.gp_lambda).view(1)
            if not self._d_synced:
                self._reduce(self.D)
            self._step(self.optD)
            if self.clip > 0 and not self._fused_clip:
                util.clamp_parameters(self.D, -self.clip, self.clip)
            if update_g:
                p, t = self._fwd(self.D, inp)
                o.adv(p, MEAN, 0.0, self._loss("g_adv"), self.dpred, +1.0)
                d_adv = self._bwd(self.D, t, self.dpred, True, False)
            self._evD[k] = sD.record_event()
        if update_g:
            # before the wait - the main stream has nothing else to do while the critic runs: clearing the generator's gradients
            # and every loss term that depends on (inpainted, ground) only, i.e. all but the adversarial one
            self.optG.zero_grad()
            g_other = self._g_losses(inp, ground)
            main.wait_event(self._evD[k])
            d_adv.record_stream(main)
            o.add(d_adv, g_other, self.tmp1)
            o.mul(self.tmp1, self.mask_c, self.g_gen)
            self._bwd_G(gtok, self.g_gen)
            self._step(self.optG)
        return self.L

    def side_stream(self):
        return self._sD if self.overlap else None

    def side_keys(self):
        return ("d_loss_real", "d_loss_fake", "gp", "g_adv") if self.overlap else ()

    def sync_for_logging(self):
        if self.overlap:
            torch.cuda.current_stream(self.G.device).wait_stream(self._sD)

    @torch.no_grad()
    def __call__(self, ground, mask, update_g):
        if self.overlap:
            return self._call_overlapped(ground, mask, update_g)
        o = self.ops
        self._d_synced = False
        self._buffers(ground)
        o.mask_apply(ground, mask, self.mask_c, self.masked, True)
        gen, gtok = self._fwd(self.G, self.masked)                              # every batch, :119
        o.composite(self.masked, gen, self.mask_c, self.inpainted)
        self.optD.zero_grad()
        if self.stacked:
            self._critic_stacked(ground, self.inpainted)
        else:
            pr, tr = self._fwd(self.D, ground)                                      # :134
            pf, tf = self._fwd(self.D, self.inpainted)                              # :135
            o.adv(pr, MEAN, 0.0, self._loss("d_loss_real"), self.dpred, +1.0)       # backward(one)  :137-138
            self._bwd(self.D, tr, self.dpred, False, True)
            o.adv(pf, MEAN, 0.0, self._loss("d_loss_fake"), self.dpred, -1.0)       # backward(mone) :140-141
            self._bwd(self.D, tf, self.dpred, False, True)
        if self.gp_lambda > 0:
            self.L["gp"] = self.D.gradient_penalty(ground, self.inpainted, getattr(self, "gp_eps", None), self.gp_lambda).view(1)
        if not self._d_synced:
            self._reduce(self.D)
        self._step(self.optD)                                                   # :147
        if self.clip > 0 and not self._fused_clip:
            util.clamp_parameters(self.D, -self.clip, self.clip)                # :151-153
        if update_g:                                                            # cadence :157-163 is the caller's
            self.optG.zero_grad()
            p, t = self._fwd(self.D, self.inpainted)
            o.adv(p, MEAN, 0.0, self._loss("g_adv"), self.dpred, +1.0)
            d_adv = self._bwd(self.D, t, self.dpred, True, False)
            o.add(d_adv, self._g_losses(self.inpainted, ground), self.tmp1)
            o.mul(self.tmp1, self.mask_c, self.g_gen)
            self._bwd_G(gtok, self.g_gen)
            self._step(self.optG)
        return self.L

    def _critic_stacked(self, ground, inp, x2=None, real_done=False):
        """D(ground) and D(inpainted) of wgan_l1.py:134-141 as one [ground | inpainted] batch, BatchNorm per half;
        backward(one) / backward(mone) are the gradient scales +1 / -1."""
        synced = self.sync is not None and self.sync.world > 1 and self.gp_lambda <= 0
        self._d_pair(self.D, ground, inp, MEAN, 0.0, 0.0, "d_loss_real", "d_loss_fake", +1.0, -1.0, synced=synced, x2=x2,
                     real_done=real_done)
        self._d_synced = synced   # the all-reduce already overlapped the tail of the backward

    def _g_losses(self, inp, ground):
        """Every non-adversarial generator loss: records the scalars, returns d(sum)/d(inpainted)."""
        self.ops.recon(self.recon, inp, ground, self._loss("recon"), self.g_rec)
        return self.g_rec


class WGANPerceptualStep(WGANStep):
    """experiment_list/wgan_perceptual_style_faceparsing.py:136-232 (BASELINE config 5): the WGAN schedule with
        g_loss = g_adv + recon_global + recon_local + perceptual + style + face_parsing + tv          (:222)
    recon_global = RMSELoss(ground, inpainted) (:206), recon_local = LocalLoss RMSE on the mask (:207; the file adds
    an undefined `recon_loss`: the sum of the two terms it computes is the evident intent), perceptual / style =
    loss.perceptual_and_style_loss(weight_p = weight_s = 0.01) (:216; constants, the reference runs them under
    no_grad), face_parsing = 0.01 * CrossEntropy(w=[0,1.2,0.7,0.7])(segment_model(inpainted), segment) (:212-213,
    frozen eval network: input gradient only), tv = tv_loss(inpainted, 1) (:219).
    `vgg` (networks.VGG19Wrapper) and `segment_model` (frozen UnetGenerator(1,4,7,ngf=32)) are optional: a missing
    one drops its term (the pretrained weights of both are not redistributable / not available here)."""

    def __init__(self, net_G, net_D, opt_G, opt_D, vgg=None, segment_model=None, weight_p=0.01, weight_s=0.01, weight_fp=0.01,
                 tv_weight=1.0, ce_weight=(0, 1.2, 0.7, 0.7), **kw):
        super().__init__(net_G, net_D, opt_G, opt_D, recon="rmse", **kw)
        self.recon_weight = 2.0
        self.vgg, self.weight_p, self.weight_s = vgg, weight_p, weight_s
        self.tv_weight, self.weight_fp, self.ce_weight = tv_weight, weight_fp, list(ce_weight)
        self.seg = getattr(segment_model, "phys", segment_model)      # EmbeddedUnetGenerator -> the physical network
        self.segment = None
        if self.seg is not None:
            self.seg.eval()
            for p in self.seg.parameters():
                p.requires_grad_(False)
            self._ce2 = torch.zeros(2, dtype=torch.float32, device=net_G.device)
            self._zero1 = torch.zeros(1, dtype=torch.float32, device=net_G.device)
            self._dseg = None

    def __call__(self, ground, mask, update_g, segment=None):
        self.segment = segment
        return super().__call__(ground, mask, update_g)

    def _g_losses(self, inp, ground):
        o = self.ops
        o.recon("rmse", inp, ground, self._loss("recon_global"), self.g_rec)                   # :206
        o.local("rmse", inp, ground, self.mask_c, self._loss("recon_local"), self.tmp2)        # :207
        o.add(self.g_rec, self.tmp2, self.g_rec)
        o.tv(inp, self.tv_weight, self._loss("tv"), self.tmp2)                                 # :219
        o.add(self.g_rec, self.tmp2, self.g_rec)
        if self.vgg is not None:                                                               # :216 (no gradient)
            p, s = self.vgg.perceptual_and_style(inp, ground, self.weight_p, self.weight_s)
            self.L["perceptual"], self.L["style"] = p.view(1), s.view(1)
        if self.seg is not None and self.segment is not None:                                  # :212-213
            import math
            n, _, h, w = inp.shape
            if self.seg._dtype == B.GI_F16:   # gradient entering the frozen network ~ weight_fp / (n*h*w) per logit
                self.seg.set_loss_scale(min(2.0 ** 24, 2.0 ** round(math.log2(0.03 * n * h * w / self.weight_fp))))
            y, tok = self._fwd(self.seg, inp)
            if self._dseg is None or self._dseg.shape != y.shape:
                self._dseg = torch.empty_like(y)
            o.cross_entropy(y, self.segment, self.ce_weight, self._ce2, self._dseg, self.weight_fp)
            o.add(self._zero1, self._ce2[:1], self._loss("face_parsing"), self.weight_fp)
            dx = self._bwd(self.seg, tok, self._dseg, True, False)
            o.add(self.g_rec, dx, self.g_rec)
        return self.g_rec


def wgan_update_g(batch_index, g_iter_count, update_g_every=5):
    """G-update cadence of wgan_l1.py:157-163."""
    period = 140 if (g_iter_count < 25 or g_iter_count % 500 == 0) else update_g_every
    return batch_index % period == 0 and batch_index > 0


class DualDStep(_StepBase):
    """G step first, LSGAN losses, global + local (mask * x) discriminators, mask not ceil-ed,
    output not composited, one optimizer over both discriminators (:123)."""

    def __init__(self, net_G, net_Dg, net_Dl, opt_G, opt_D, lam1=300.0, lam2=300.0, sync=None, stacked=True):
        super().__init__(net_G, [net_Dg, net_Dl], net_G.device, sync)
        self.stacked = stacked      # each discriminator's real | fake calls as one batch with per-half BatchNorm
        self.Dg, self.Dl, self.optG, self.optD, self.lam1, self.lam2 = net_Dg, net_Dl, opt_G, opt_D, lam1, lam2
        self.recon_weight = max(lam1, lam2)
        self._bind_optimizers(opt_G, opt_D)

    @torch.no_grad()
    def __call__(self, ground, mask):
        o = self.ops
        self._buffers(ground)
        o.mask_apply(ground, mask, None, self.masked, False)                    # :144
        self.optG.zero_grad()
        gen, gtok = self._fwd(self.G, self.masked)                              # :154
        self.gen = gen                                                          # the reference's `inpainted` (:154, metric at :209)
        # D_global(inpainted), D_local(mask*inpainted) with frozen Ds :156-157
        p, t = self._fwd(self.Dg, gen)
        o.adv(p, LSGAN, 1.0, self._loss("g_adv_global"), self.dpred)
        d1 = self._bwd(self.Dg, t, self.dpred, True, False)
        o.mul(mask, gen, self.inpainted)                                        # inpainted buffer = mask*gen
        p, t = self._fwd(self.Dl, self.inpainted)
        o.adv(p, LSGAN, 1.0, self._loss("g_adv_local"), self.dpred)
        d2 = self._bwd(self.Dl, t, self.dpred, True, False)
        o.mul(d2, mask, self.tmp1)
        o.add(d1, self.tmp1, self.g_adv)
        o.recon("rmse", gen, ground, self._loss("rmse_global"), self.g_rec, self.lam1)        # :166
        o.add(self.g_adv, self.g_rec, self.g_adv)
        o.mul(mask, ground, self.tmp2)
        o.recon("rmse", self.inpainted, self.tmp2, self._loss("rmse_local"), self.g_rec, self.lam2)  # :167
        o.mul(self.g_rec, mask, self.tmp1)
        o.add(self.g_adv, self.tmp1, self.g_gen)
        self._bwd_G(gtok, self.g_gen)
        self._step(self.optG)                                                        # :172
        # ---- D step :178-200
        self.optD.zero_grad()
        for net, real, fake, tag in ((self.Dg, ground, gen, "global"), (self.Dl, self.tmp2, self.inpainted, "local")):
            if self.stacked:
                self._d_pair(net, real, fake, LSGAN, 1.0, 0.0, f"d_real_{tag}", f"d_fake_{tag}")
                continue
            p, t = self._fwd(net, real)
            o.adv(p, LSGAN, 1.0, self._loss(f"d_real_{tag}"), self.dpred)
            self._bwd(net, t, self.dpred, False, True)
            p, t = self._fwd(net, fake)
            o.adv(p, LSGAN, 0.0, self._loss(f"d_fake_{tag}"), self.dpred)
            self._bwd(net, t, self.dpred, False, True)
        self._reduce(self.Dg)
        self._reduce(self.Dl)
        self._step(self.optD)
        return self.L
