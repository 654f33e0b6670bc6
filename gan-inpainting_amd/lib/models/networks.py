"""Drop-in for the reference's lib/models/networks.py (get_network :13-28, UnetGenerator :216-253,
PatchGANDiscriminator :331-363) on the HIP backend.

The classes are torch.nn.Modules whose parameters/buffers are zero-copy views into flat fp32
device buffers owned by the module and bound to a libganinpaint handle:
  * state_dict() keys, shapes, dtypes and named_parameters() order equal the reference's
    (conv weights are logical [a,b,4,4] tensors with channels_last strides);
  * `.grad` of every parameter is a view into the flat gradient buffer the kernels accumulate into;
  * __call__ takes/returns fp32 (N,1,H,W) / (N,1) tensors and takes part in torch autograd through
    one autograd.Function per network (whole-net forward / backward in the library).
Nothing here computes on the CPU: calling a module that is not on a gfx950 device raises.
"""
import ctypes as C
import functools
import math

import torch
from torch import nn

from ... import backend as B

DEFAULT_DTYPE = "fp16"


def set_default_dtype(dtype):
    """Compute/storage type of networks created afterwards: 'fp16' (MFMA f16, fp32 accumulate,
    fp32 master weights, loss-scaled backward) or 'fp32' (exact-fp32 MFMA)."""
    global DEFAULT_DTYPE
    B.dtype_code(dtype)
    DEFAULT_DTYPE = dtype


class Identity(nn.Module):
    def forward(self, x):
        return x


def get_norm_layer(norm_type="instance"):
    """networks.py:30-45. get_network only ever builds 'batch' (networks.py:18); UnetGenerator accepts all three."""
    if norm_type == "batch":
        return functools.partial(nn.BatchNorm2d, affine=True, track_running_stats=True)
    if norm_type == "instance":
        return functools.partial(nn.InstanceNorm2d, affine=False, track_running_stats=False)
    if norm_type == "none":
        return lambda x: Identity()
    raise NotImplementedError("normalization layer [%s] is not found" % norm_type)


def get_network(type, name, **kw):
    """networks.py:13-28. vgg19 / dcgan are not on the accelerated path."""
    if type == "generator":
        if name == "unet":
            return UnetGenerator(1, 1, 7, ngf=64, norm_layer=get_norm_layer(norm_type="batch"),
                                 use_dropout="False", **kw)
        if name == "vgg19":
            raise NotImplementedError("VGG19Generator is outside the HIP backend's hot path")
        raise Exception("Invalid generator network name")
    if type == "discriminator":
        if name == "patchgan":
            return PatchGANDiscriminator(**kw)
        if name == "dcgan":
            raise NotImplementedError("DCGANDiscriminator is outside the HIP backend's hot path")
        raise Exception("Invalid discriminator network name")
    raise Exception("Invalid network type")


class _Holder(nn.Module):
    """Structural node: reproduces the reference's module nesting so state_dict keys match."""


class _NetFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, anchor, net):
        # an eval-mode forward (running-statistics BatchNorm) is differentiable w.r.t. its input only
        need_wgrad = net.training and any(p.requires_grad for p in net.parameters())
        y, slot, gen = net._forward_raw(x)
        ctx.net, ctx.slot, ctx.gen = net, slot, gen
        ctx.need_dx = x.requires_grad
        ctx.need_wgrad = need_wgrad
        return y

    @staticmethod
    def backward(ctx, dy):
        net = ctx.net
        dx = net._backward_raw(ctx.slot, ctx.gen, dy, ctx.need_dx, ctx.need_wgrad)
        return dx, None, None


class HipNet(nn.Module):
    """Common machinery: inventory, flat buffers, handle (re)creation, raw forward/backward."""

    n_slots = 3

    def __init__(self, dtype=None):
        super().__init__()
        self._dtype = B.dtype_code(dtype or DEFAULT_DTYPE)
        self._handle = None          # sized handle (device)
        self._geom = None            # (H, W, max_n)
        self._flat = None            # dict of flat device tensors
        self._dirty = True
        self._next_slot = 0
        self._slot_gen = [0] * self.n_slots
        self._bn_count_buffers = []
        self._anchor = None
        self._loss_scale = 65536.0 if self._dtype == B.GI_F16 else 1.0
        self._nbt_pending = 0
        self.always_sync = False
        self._slot_groups = [1] * self.n_slots

    # ---- subclass hooks ----------------------------------------------------------------------
    def _create_handle(self, ctx, H, W, max_n):
        raise NotImplementedError

    def _output_shape(self, n, H, W):
        raise NotImplementedError

    # ---- inventory ---------------------------------------------------------------------------
    def _inventory(self, handle):
        lib = B.lib()
        out = []
        name = C.create_string_buffer(256)
        kind, ndim = C.c_int(), C.c_int()
        shape = (C.c_int64 * 4)()
        off, numel = C.c_int64(), C.c_int64()
        for i in range(lib.gi_net_tensor_count(handle)):
            B.check(lib.gi_net_tensor_desc(handle, i, name, 256, C.byref(kind), shape, C.byref(ndim), C.byref(off),
                                           C.byref(numel)))
            out.append(dict(name=name.value.decode(), kind=kind.value, shape=tuple(shape[j] for j in range(ndim.value)),
                            offset=off.value, numel=numel.value))
        return out

    def _build_tree(self, inv, n_params, n_buffers):
        """Create the nested holders and register parameters/buffers as views of CPU flat buffers."""
        self._inv = inv
        self._n_params, self._n_buffers = n_params, n_buffers
        flat = dict(params=torch.zeros(n_params), grads=None, buffers=torch.zeros(n_buffers))
        self._flat = flat
        self._tensor_refs = {}
        for t in inv:
            parts = t["name"].split(".")
            mod = self
            for p in parts[:-1]:
                if p not in mod._modules:
                    mod.add_module(p, _Holder())
                mod = mod._modules[p]
            view = self._view(flat["params" if t["kind"] <= 1 else "buffers"], t)
            if t["kind"] <= 1:
                prm = nn.Parameter(view, requires_grad=True)
                mod.register_parameter(parts[-1], prm)
                self._tensor_refs[t["name"]] = (mod, parts[-1], True)
            else:
                mod.register_buffer(parts[-1], view)
                self._tensor_refs[t["name"]] = (mod, parts[-1], False)
                if t["kind"] == 3:  # after running_var: the reference's num_batches_tracked buffer
                    mod.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
                    self._bn_count_buffers.append(mod)

    @staticmethod
    def _view(flat, t):
        v = flat[t["offset"]: t["offset"] + t["numel"]]
        if t["kind"] == 0:  # physical [a][ky][kx][b] -> logical [a,b,4,4] (channels_last strides)
            a, b = t["shape"][0], t["shape"][1]
            return v.view(a, 4, 4, b).permute(0, 3, 1, 2)
        return v.view(*t["shape"])

    def _reset_batchnorm(self):
        for t in self._inv:
            if t["kind"] == 2:
                prefix = t["name"][:-len("running_mean")]
                self._tensor(prefix + "weight").fill_(1.0)
                self._tensor(prefix + "bias").zero_()
                self._tensor(prefix + "running_mean").zero_()
                self._tensor(prefix + "running_var").fill_(1.0)
        for mod in self._bn_count_buffers:
            mod._buffers["num_batches_tracked"].zero_()

    def _tensor(self, name):
        mod, leaf, is_param = self._tensor_refs[name]
        return mod._parameters[leaf] if is_param else mod._buffers[leaf]

    # ---- device placement --------------------------------------------------------------------
    def _apply(self, fn, recurse=True):
        # .to()/.cuda()/.cpu()/.float(): move the FLAT buffers and re-point every view at them
        old = self._flat
        new_params = fn(old["params"])
        new_buffers = fn(old["buffers"])
        if new_params.dtype != torch.float32:
            raise TypeError("HIP backend keeps fp32 master weights; choose the compute type with dtype='fp16'")
        moved = new_params.device != old["params"].device
        self._flat = dict(params=new_params.contiguous(), buffers=new_buffers.contiguous(), grads=None)
        with torch.no_grad():
            for t in self._inv:
                mod, leaf, is_param = self._tensor_refs[t["name"]]
                view = self._view(self._flat["params" if t["kind"] <= 1 else "buffers"], t)
                if is_param:
                    mod._parameters[leaf].data = view
                    mod._parameters[leaf].grad = None
                else:
                    mod._buffers[leaf] = view
            for mod in self._bn_count_buffers:
                mod._buffers["num_batches_tracked"] = fn(mod._buffers["num_batches_tracked"])
        if moved:
            self._release_handle()
        self._dirty = True
        if self._flat["params"].is_cuda:
            self._attach_grads()
            self._anchor = torch.zeros(1, device=self._flat["params"].device, requires_grad=True)
        return self

    def _attach_grads(self):
        if self._flat["grads"] is None:
            self._flat["grads"] = torch.zeros_like(self._flat["params"])
        for t in self._inv:
            if t["kind"] <= 1:
                p = self._tensor(t["name"])
                g = self._view(self._flat["grads"], t)
                if p.grad is None or p.grad.data_ptr() != g.data_ptr():
                    p.grad = g
                p._gi_owner = self

    def _release_handle(self):
        if self._handle is not None:
            B.lib().gi_net_destroy(self._handle)
            self._handle = None
            self._geom = None
            self._ws = None

    def __del__(self):
        try:
            self._release_handle()
        except Exception:
            pass

    @property
    def device(self):
        return self._flat["params"].device

    def flat_params(self):
        return self._flat["params"]

    def flat_grads(self):
        self._attach_grads()
        return self._flat["grads"]

    def mark_dirty(self):
        """Parameters were written outside the library: packed fp16/phase copies are stale."""
        self._dirty = True

    def set_loss_scale(self, scale):
        self._loss_scale = float(scale)
        if self._handle is not None:
            B.check(B.lib().gi_net_set_loss_scale(self._handle, self._loss_scale))

    def zero_grad(self, set_to_none=False):
        if self._flat["grads"] is not None:
            self._flat["grads"].zero_()
        self._attach_grads() if self._flat["params"].is_cuda else None

    def _flush_nbt(self):
        if self._nbt_pending:
            for mod in self._bn_count_buffers:
                mod._buffers["num_batches_tracked"] += self._nbt_pending
            self._nbt_pending = 0

    def state_dict(self, *args, **kw):
        self._flush_nbt()
        return super().state_dict(*args, **kw)

    def load_state_dict(self, state_dict, strict=True, **kw):
        self._nbt_pending = 0
        res = super().load_state_dict(state_dict, strict=strict, **kw)
        self._dirty = True
        return res

    # ---- handle -------------------------------------------------------------------------------
    def _ensure_handle(self, n, H, W):
        if not self._flat["params"].is_cuda:
            raise B.BackendError(f"{type(self).__name__} lives on {self.device}: the HIP backend runs on a gfx950 "
                                 "device only (no CPU fallback). Call .to('cuda') first.")
        if self._handle is not None and self._geom[0] == H and self._geom[1] == W and n <= self._geom[2]:
            return
        self._release_handle()
        ctx = B.get_ctx(self.device)
        h = self._create_handle(ctx, H, W, n)
        lib = B.lib()
        assert lib.gi_net_param_floats(h) == self._n_params, "parameter inventory changed with geometry"
        nbytes = lib.gi_net_workspace_bytes(h)
        self._ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=self.device)
        off = (-self._ws.data_ptr()) % 256
        self._attach_grads()
        B.check(lib.gi_net_bind(h, B.ptr(self._flat["params"]), B.ptr(self._flat["grads"]), B.ptr(self._flat["buffers"]),
                                self._ws.data_ptr() + off, nbytes))
        B.check(lib.gi_net_set_loss_scale(h, self._loss_scale))
        self._handle, self._geom = h, (H, W, n)
        self._groups_set = 1
        self._dirty = True

    def _forward_raw(self, x, bn_groups=1, inference=False):
        """bn_groups = 2 (discriminators only): x stacks two equally sized batches that are normalised with
        independent BatchNorm statistics, i.e. two reference forward calls in one launch sequence.
        inference (eval mode only): this forward will never be differentiated; generators then run with their
        BatchNorm layers folded into the convolutions (gi_net_set_inference)."""
        if x.dim() != 4 or x.shape[1] != 1:
            raise ValueError(f"expected (N,1,H,W) input, got {tuple(x.shape)}")
        if x.dtype != torch.float32 or not x.is_cuda:
            raise B.BackendError("input must be a float32 tensor on the module's gfx950 device")
        x = x.contiguous()
        n, _, H, W = x.shape
        self._ensure_handle(n, H, W)
        lib = B.lib()
        if self._dirty or self.always_sync:
            B.check(lib.gi_net_sync_weights(self._handle))
            self._dirty = False
        B.check(lib.gi_net_set_train(self._handle, 1 if self.training else 0))
        inference = bool(inference) and not self.training
        if getattr(self, "_inference_set", (None, False)) != (self._handle, inference):
            B.check(lib.gi_net_set_inference(self._handle, 1 if inference else 0))
            self._inference_set = (self._handle, inference)
        if bn_groups != 1 or self._groups_set != 1:
            B.check(lib.gi_net_set_bn_groups(self._handle, bn_groups))
            self._groups_set = bn_groups
        slot = self._next_slot
        self._next_slot = (slot + 1) % self.n_slots
        self._slot_gen[slot] += 1
        self._slot_groups[slot] = bn_groups
        y = torch.empty(self._output_shape(n, H, W), dtype=torch.float32, device=x.device)
        B.check(lib.gi_net_forward(self._handle, slot, B.ptr(x), B.ptr(y), n))
        if self.training:
            self._nbt_pending += bn_groups   # num_batches_tracked is bookkeeping only: materialised lazily
        self._last_slot = slot
        return y, slot, self._slot_gen[slot]

    def _backward_raw(self, slot, gen, dy, need_dx, need_wgrad):
        if self._slot_gen[slot] != gen:
            raise B.BackendError(f"activations of this forward were overwritten: more than {self.n_slots} forwards of "
                                 f"{type(self).__name__} were live before their backward")
        dy = dy.contiguous().float()
        n = dy.shape[0]
        dx = torch.empty((n, 1, self._geom[0], self._geom[1]), dtype=torch.float32, device=dy.device) if need_dx else None
        self._attach_grads()
        if self._slot_groups[slot] != self._groups_set:
            B.check(B.lib().gi_net_set_bn_groups(self._handle, self._slot_groups[slot]))
            self._groups_set = self._slot_groups[slot]
        B.check(B.lib().gi_net_backward(self._handle, slot, B.ptr(dy), B.ptr(dx), 1 if need_wgrad else 0))
        return dx

    def saved_activation(self, kind, level, shape, slot=None):
        """fp32 (N,C,h,w) copy of an activation the last forward saved for its backward (gi_net_saved_activation): its sign is
        the side of the ReLU / LeakyReLU kink that forward took. Parity tests only."""
        slot = self._last_slot if slot is None else slot
        out = torch.empty(tuple(shape), dtype=torch.float32, device=self.device)
        B.check(B.lib().gi_net_saved_activation(self._handle, slot, kind, level, B.ptr(out), out.numel()))
        return out

    def nonzero_tickets(self):
        """Debug: split-K tile tickets that are not zero between launches (gi_net_debug_nonzero_tickets; must be 0)."""
        import ctypes as C
        n = C.c_int()
        B.check(B.lib().gi_net_debug_nonzero_tickets(self._handle, C.byref(n)))
        return n.value

    def forward(self, x):
        if torch.is_grad_enabled() and (self.training or x.requires_grad):
            return _NetFunction.apply(x, self._anchor, self)
        return self._forward_raw(x, inference=not self.training)[0]   # no autograd node: nothing can be differentiated


def _init_conv(cout, cin, transposed=False, bias=False):
    """Draw from torch's global RNG exactly as the reference's constructors do
    (nn.Conv2d / nn.ConvTranspose2d default init: kaiming_uniform(a=sqrt(5)), then bias)."""
    m = (nn.ConvTranspose2d(cin, cout, 4, 2, 1, bias=bias) if transposed else nn.Conv2d(cin, cout, 4, 2, 1, bias=bias))
    return m.weight.detach(), (m.bias.detach() if bias else None)


def _norm_kind(norm_layer):
    """0 BatchNorm2d, 1 InstanceNorm2d (parameter-free, instance statistics in train and eval), 2 no norm: the three
    results of get_norm_layer (networks.py:29-45). use_bias follows the reference's test `func == nn.InstanceNorm2d`."""
    fn = norm_layer.func if isinstance(norm_layer, functools.partial) else norm_layer
    kw = norm_layer.keywords if isinstance(norm_layer, functools.partial) else {}
    if fn is nn.BatchNorm2d:
        if not kw.get("affine", True) or not kw.get("track_running_stats", True):
            raise NotImplementedError("HIP backend: BatchNorm2d generators are built with affine=True, track_running_stats=True")
        return 0
    if fn is nn.InstanceNorm2d:
        if kw.get("affine", False) or kw.get("track_running_stats", False):
            raise NotImplementedError("HIP backend: InstanceNorm2d generators are built with affine=False, track_running_stats=False")
        return 1
    try:
        probe = norm_layer(8)
    except Exception as e:      # noqa: BLE001
        raise NotImplementedError(f"HIP backend: norm_layer {norm_layer!r} is not one of get_norm_layer's results") from e
    if isinstance(probe, (Identity, nn.Identity)):
        return 2
    raise NotImplementedError(f"HIP backend: norm_layer {norm_layer!r} is not one of get_norm_layer's results")


class UnetGenerator(HipNet):
    """networks.py:216-253 (+ UnetSkipConnectionBlock :255-324). input_nc = 1; output_nc = 1 (the inpainting
    generator) or up to 64 (the frozen face-parsing network UnetGenerator(1,4,7,ngf=32), train.py:171-172:
    forward + input gradient only). ngf must be a multiple of 64 on the device; other widths (ngf=32) are
    served by EmbeddedUnetGenerator, which this constructor returns transparently."""

    def __new__(cls, input_nc=1, output_nc=1, num_downs=7, ngf=64, *args, **kw):
        if cls is UnetGenerator and ngf % 64 != 0:
            return EmbeddedUnetGenerator(input_nc, output_nc, num_downs, ngf, *args, **kw)
        return super().__new__(cls)

    _ch1 = 0     # physical width of level 1 (0: ngf); set by _PaddedUnetGenerator

    def __init__(self, input_nc=1, output_nc=1, num_downs=7, ngf=64, norm_layer=nn.BatchNorm2d, use_dropout=False,
                 use_sigmoid_output=False, dtype=None):
        super().__init__(dtype)
        if input_nc != 1 or not (1 <= output_nc <= 64):
            raise NotImplementedError("HIP backend: UnetGenerator takes 1-channel images and emits 1..64 channels")
        self.output_nc = output_nc
        self.norm_kind = _norm_kind(norm_layer)
        if use_sigmoid_output:
            raise NotImplementedError("use_sigmoid_output is never enabled by the reference (networks.py:250)")
        self.num_downs, self.ngf = num_downs, ngf
        # any truthy value enables dropout, including the string 'False' (networks.py:18-19, :313)
        self.dropout_p = 0.5 if use_dropout else 0.0
        lib = B.lib()
        h = C.c_void_p()
        size0 = 1 << max(num_downs, 7)
        B.check(lib.gi_unet_create_padded(None, num_downs, ngf, self._ch1, output_nc, self.norm_kind, self.dropout_p, size0, size0, 1,
                                          self._dtype, 1, C.byref(h)))
        self._build_tree(self._inventory(h), lib.gi_net_param_floats(h), lib.gi_net_buffer_floats(h))
        lib.gi_net_destroy(h)
        self.reset_parameters()

    def _level_names(self):
        """(down conv, up conv) weight names per level 1..num_downs (nesting of networks.py:296-318)."""
        out, p = [], "model.model"
        for k in range(1, self.num_downs + 1):
            outer, inner = k == 1, k == self.num_downs
            out.append((p + (".0" if outer else ".1") + ".weight", p + (".3" if (outer or inner) else ".5") + ".weight"))
            p = p + (".1" if outer else ".3") + ".model"
        return out

    def reset_parameters(self):
        """Same RNG consumption order as the reference constructor: blocks are built innermost
        first (networks.py:236-243), each drawing downconv(+bias) then upconv(+bias) (:285-309; with InstanceNorm
        every convolution has a bias, :270-273); norm layers draw nothing (BatchNorm: weight 1, bias 0, running
        stats 0/1)."""
        with torch.no_grad():
            for down, up in reversed(self._level_names()):
                a, b = self._tensor(down).shape[:2]
                bias_name = down[:-len("weight")] + "bias"
                has_bias = bias_name in self._tensor_refs
                w, bias = _init_conv(a, b, bias=has_bias)                 # Conv2d weight [out=a, in=b, 4, 4]
                self._tensor(down).copy_(w)
                if has_bias:
                    self._tensor(bias_name).copy_(bias)
                a, b = self._tensor(up).shape[:2]
                bias_name = up[:-len("weight")] + "bias"
                has_bias = bias_name in self._tensor_refs
                w, bias = _init_conv(b, a, transposed=True, bias=has_bias)  # ConvTranspose2d weight [in=a, out=b, 4, 4]
                self._tensor(up).copy_(w)
                if has_bias:
                    self._tensor(bias_name).copy_(bias)
            self._reset_batchnorm()
        self._dirty = True

    def _create_handle(self, ctx, H, W, max_n):
        h = C.c_void_p()
        B.check(B.lib().gi_unet_create_padded(ctx, self.num_downs, self.ngf, self._ch1, self.output_nc, self.norm_kind, self.dropout_p, H, W,
                                              max_n, self._dtype, self.n_slots, C.byref(h)))
        if getattr(self, "_drop_seed", None) is not None:      # a seed chosen before the handle existed
            B.check(B.lib().gi_net_set_dropout_seed(h, self._drop_seed))
        return h

    def _output_shape(self, n, H, W):
        return (n, self.output_nc, H, W)

    # dropout control (parity tests feed the device-generated masks to the oracle, or impose masks)
    def set_dropout_seed(self, seed):
        self._drop_seed = int(seed)
        if self._handle is not None:
            B.check(B.lib().gi_net_set_dropout_seed(self._handle, self._drop_seed))

    def dropout_masks(self, slot=None):
        """{level: uint8 keep-mask (N,C,H,W)} used by the last train-mode forward."""
        slot = self._last_slot if slot is None else slot
        out = {}
        H, W, _ = self._geom
        n = None
        for lvl in range(5, self.num_downs):
            c = self.ngf * min(2 ** (lvl - 2), 8)
            h, w = H >> (lvl - 1), W >> (lvl - 1)
            n = self._last_n if n is None else n
            m = torch.empty((n, c, h, w), dtype=torch.uint8, device=self.device)
            B.check(B.lib().gi_net_dropout_mask(self._handle, slot, lvl, B.ptr(m), m.numel()))
            out[lvl] = m
        return out

    def _forward_raw(self, x, bn_groups=1, inference=False):
        self._last_n = x.shape[0]
        if getattr(self, "_pending_masks", None):
            self._ensure_handle(x.shape[0], x.shape[2], x.shape[3])
            slot = self._next_slot
            self._mask_keep = {k: v.to(self.device).contiguous() for k, v in self._pending_masks.items()}
            for lvl, m in self._mask_keep.items():
                B.check(B.lib().gi_net_set_dropout_mask(self._handle, slot, lvl, B.ptr(m)))
            out = super()._forward_raw(x, inference=inference)
            for lvl in self._mask_keep:
                B.check(B.lib().gi_net_set_dropout_mask(self._handle, slot, lvl, None))
            self._pending_masks = None
            return out
        return super()._forward_raw(x, inference=inference)

    def impose_dropout_masks(self, masks):
        """Use these keep-masks ({level: uint8 (N,C,H,W)}) in the NEXT forward (parity tests)."""
        self._pending_masks = dict(masks)


class _PaddedUnetGenerator(UnetGenerator):
    """UnetGenerator(ngf=32) on the device: level 1 is 64 channels wide (upper 32 zero), every other level has its true
    width (64, 128, 256, ...): gi_unet_create_padded. The tensors that touch level 1 have the padded shapes."""
    _ch1 = 64


class EmbeddedUnetGenerator(nn.Module):
    """UnetGenerator whose width is not a multiple of 64 (the reference's face-parsing network has ngf=32,
    train.py:171-172) on kernels whose channel counts are multiples of 64, by zero-embedding: padded channel axes get zero
    weights, BatchNorm of the padded channels has weight = bias = running_mean = 0 and running_var = 1, and the two halves
    of a skip concatenation are embedded separately. Padded channels carry exact zeros through convolutions, BatchNorm,
    LeakyReLU/ReLU and their gradients, so outputs and input gradients equal the narrow network's
    (tests/test_segnet_gpu.py).
      * ngf = 32: only level 1 (32 channels) is below the granularity - that level alone is padded to 64, levels 2..7 run
        at their true widths 64, 128, 256, ... (gi_unet_create_padded): 1.15x the narrow network's FLOPs;
      * other widths: every level is widened to ngf' = 64*ceil(ngf/64) ((ngf'/ngf)^2 x the FLOPs).
    state_dict()/load_state_dict() speak the NARROW shapes and the reference's key names."""

    def __init__(self, input_nc=1, output_nc=1, num_downs=7, ngf=32, norm_layer=nn.BatchNorm2d, use_dropout=False,
                 use_sigmoid_output=False, dtype=None):
        super().__init__()
        if _norm_kind(norm_layer) != 0:
            raise NotImplementedError("HIP backend: generators narrower than 64 channels are built with BatchNorm2d only")
        self.ngf, self.num_downs, self.output_nc = ngf, num_downs, output_nc
        self.ngf_phys = 64 * ((ngf + 63) // 64)
        lib = B.lib()
        h = C.c_void_p()
        size0 = 1 << max(num_downs, 7)
        dcode = B.dtype_code(dtype or DEFAULT_DTYPE)
        B.check(lib.gi_unet_create_ex(None, num_downs, ngf, output_nc, 0.0, size0, size0, 1, dcode, 1, C.byref(h)))
        self._narrow = {t["name"]: t for t in HipNet._inventory(self, h)}
        lib.gi_net_destroy(h)
        # draw the narrow network's initial weights exactly like the reference constructor, then embed
        probe = _NarrowInit(self._narrow, num_downs)
        if ngf == 32:     # only level 1 is narrower than the kernels' 64-channel granularity: pad that level alone (1.15x the FLOPs, not 4x)
            self.phys = _PaddedUnetGenerator(input_nc, output_nc, num_downs, ngf, norm_layer, use_dropout, use_sigmoid_output, dtype)
        else:
            self.phys = UnetGenerator(input_nc, output_nc, num_downs, self.ngf_phys, norm_layer, use_dropout, use_sigmoid_output, dtype)
        self._phys_shapes = {k: tuple(v.shape) for k, v in self.phys.state_dict().items()}
        self.load_state_dict(probe.state, strict=True)

    # channel index maps narrow -> physical per tensor axis
    def _axis_maps(self, name, shape):
        pshape = self._phys_shapes[name]                        # the physical tensor: each axis at least as long
        def plain(c):
            return torch.arange(c)
        def halves(c, cphys):                                   # [skip | decoder] concat of two c/2 blocks (physical: cphys/2 each)
            h = c // 2
            return torch.cat([torch.arange(h), cphys // 2 + torch.arange(h)])
        is_up_conv = name.endswith(".weight") and len(shape) == 4 and name in self._up_names()
        maps = []
        for ax, c in enumerate(shape[:2] if len(shape) == 4 else shape):
            if len(shape) == 4 and is_up_conv and ax == 0 and name != self._up_names()[-1]:
                maps.append(halves(c, pshape[ax]))              # ConvTranspose2d in-channels = concat (all but the innermost)
            else:
                maps.append(plain(c))
        return maps

    def _up_names(self):
        out, p = [], "model.model"
        for k in range(1, self.num_downs + 1):
            outer, inner = k == 1, k == self.num_downs
            out.append(p + (".3" if (outer or inner) else ".5") + ".weight")
            p = p + (".1" if outer else ".3") + ".model"
        return out

    def state_dict(self, *args, **kw):
        phys = self.phys.state_dict()
        out = {}
        for name, t in self._narrow.items():
            v = phys[name]
            maps = self._axis_maps(name, t["shape"])
            if len(t["shape"]) == 4:
                v = v[maps[0].to(v.device)][:, maps[1].to(v.device)]
            else:
                v = v[maps[0].to(v.device)]
            out[name] = v.clone()
            if name.endswith("running_var"):
                out[name[:-len("running_var")] + "num_batches_tracked"] = phys[name[:-len("running_var")] + "num_batches_tracked"].clone()
        return out

    def load_state_dict(self, sd, strict=True):
        phys = self.phys.state_dict()
        missing = [k for k in self._narrow if k not in sd]
        if strict and missing:
            raise RuntimeError("EmbeddedUnetGenerator.load_state_dict: missing keys %s" % missing)
        for name, t in self._narrow.items():
            if name not in sd:
                continue
            src = torch.as_tensor(sd[name]).float()
            if tuple(src.shape) != tuple(t["shape"]):
                raise RuntimeError("size mismatch for %s: %s vs %s" % (name, tuple(src.shape), tuple(t["shape"])))
            dst = phys[name]
            full = torch.ones_like(dst) if name.endswith("running_var") else torch.zeros_like(dst)
            maps = self._axis_maps(name, t["shape"])
            if len(t["shape"]) == 4:
                full[maps[0].to(dst.device)[:, None], maps[1].to(dst.device)[None, :]] = src.to(dst.device)
            else:
                full[maps[0].to(dst.device)] = src.to(dst.device)
            phys[name] = full
        self.phys.load_state_dict(phys, strict=True)
        return missing

    def forward(self, x):
        return self.phys(x)


class _NarrowInit:
    """Initial weights of a narrow UnetGenerator drawn from torch's RNG in the reference constructor's order."""

    def __init__(self, inv, num_downs):
        self.state = {}
        names, p = [], "model.model"
        for k in range(1, num_downs + 1):
            outer, inner = k == 1, k == num_downs
            names.append((p + (".0" if outer else ".1") + ".weight", p + (".3" if (outer or inner) else ".5") + ".weight"))
            p = p + (".1" if outer else ".3") + ".model"
        for down, up in reversed(names):
            a, b = inv[down]["shape"][:2]
            self.state[down] = _init_conv(a, b)[0]
            a, b = inv[up]["shape"][:2]
            bias_name = up[:-len("weight")] + "bias"
            w, bias = _init_conv(b, a, transposed=True, bias=bias_name in inv)
            self.state[up] = w
            if bias is not None:
                self.state[bias_name] = bias
        for name, t in inv.items():
            if name.endswith("running_mean"):
                pre = name[:-len("running_mean")]
                c = t["shape"][0]
                self.state[pre + "weight"], self.state[pre + "bias"] = torch.ones(c), torch.zeros(c)
                self.state[pre + "running_mean"], self.state[pre + "running_var"] = torch.zeros(c), torch.ones(c)


class Flatten(nn.Module):
    def forward(self, input):
        return input.view(input.size(0), -1)


class PatchGANDiscriminator(HipNet):
    """networks.py:331-363. `image_size` generalises the hard-coded Linear(25,1) (valid for
    128x128 only) to Linear((H/16-3)*(W/16-3), 1); n_d_channel is ignored as in the reference."""

    def __init__(self, c=1, n_d_channel=64, sigmoid=True, image_size=128, dtype=None):
        super().__init__(dtype)
        if c != 1:
            raise NotImplementedError("HIP backend: PatchGANDiscriminator is built for 1-channel images")
        self.sigmoid = bool(sigmoid)
        self.image_size = (image_size, image_size) if isinstance(image_size, int) else tuple(image_size)
        lib = B.lib()
        h = C.c_void_p()
        B.check(lib.gi_patchgan_create(None, self.image_size[0], self.image_size[1], int(self.sigmoid), 1, self._dtype, 1,
                                       C.byref(h)))
        self._build_tree(self._inventory(h), lib.gi_net_param_floats(h), lib.gi_net_buffer_floats(h))
        lib.gi_net_destroy(h)
        self.always_sync = True   # 2.8 M params: re-deriving the packed copies costs microseconds and makes
        #                           raw `p.data.clamp_()` loops (wgan_l1.py:151-153) safe
        self.reset_parameters()

    def reset_parameters(self):
        with torch.no_grad():
            chans = [1, 64, 128, 256, 512]
            for i, idx in enumerate((0, 2, 5, 8)):
                w, _ = _init_conv(chans[i + 1], chans[i])
                self._tensor(f"model.{idx}.weight").copy_(w)
            m = nn.Conv2d(512, 1, 4, 1, 0, bias=False)
            self._tensor("model.11.weight").copy_(m.weight.detach())
            lin = nn.Linear(self._tensor("model.13.weight").shape[1], 1)
            self._tensor("model.13.weight").copy_(lin.weight.detach())
            self._tensor("model.13.bias").copy_(lin.bias.detach())
            self._reset_batchnorm()
        self._dirty = True

    def _create_handle(self, ctx, H, W, max_n):
        if (H, W) != self.image_size:
            raise ValueError(f"PatchGANDiscriminator was built for {self.image_size} images (its Linear head has "
                             f"{self._tensor('model.13.weight').shape[1]} inputs), got {(H, W)}")
        h = C.c_void_p()
        B.check(B.lib().gi_patchgan_create(ctx, H, W, int(self.sigmoid), max_n, self._dtype, self.n_slots, C.byref(h)))
        return h

    def _output_shape(self, n, H, W):
        return (n, 1)

    def gradient_penalty(self, real, fake, eps=None, lam=10.0):
        """WGAN-GP extension (not in the reference): adds d/dtheta of
        lam * mean_n (||grad_x D(xhat)_n||_2 - 1)^2, xhat = eps*real + (1-eps)*fake, to the parameter
        gradients and returns the penalty (0-d device tensor). eps: (n,) in [0,1), drawn here if None.
        fp32 critics built with sigmoid=False."""
        real, fake = real.contiguous(), fake.detach().contiguous()
        n, _, H, W = real.shape
        if eps is None:
            eps = torch.rand(n, device=real.device)
        eps = eps.to(real.device, torch.float32).contiguous().view(-1)
        self._ensure_handle(n, H, W)
        lib, ctx = B.lib(), B.get_ctx(real.device)
        if self._dirty or self.always_sync:
            B.check(lib.gi_net_sync_weights(self._handle))
            self._dirty = False
        B.check(lib.gi_net_set_train(self._handle, 1))
        xhat = torch.empty_like(real)
        B.check(lib.gi_interpolate(ctx, B.ptr(real), B.ptr(fake), B.ptr(eps), n, H * W, B.ptr(xhat)))
        out = torch.zeros(1, dtype=torch.float32, device=real.device)
        self._attach_grads()
        B.check(lib.gi_patchgan_gradient_penalty(self._handle, B.ptr(xhat), n, float(lam), B.ptr(out)))
        self._nbt_pending += 1   # the penalty's own forward runs the BatchNorm layers in train mode
        return out.view(())


class VGG19Wrapper(nn.Module):
    """The feature network behind the perceptual / style losses (reference networks.py:367-389:
    torchvision `vgg19(pretrained=True)` with hooks on features[1,6,11,20,29]; loss.py:4 builds one at
    import). Here: the 13 convolutions up to features.28 as one flat fp32 parameter buffer driven by
    `gi_vgg19_*` (fp16 MFMA 3x3 implicit GEMM, forward only - the reference calls it under no_grad).

    The pretrained weights cannot be downloaded in this environment: the constructor initialises like
    torchvision does for `pretrained=False` (He-normal fan_out, zero bias); `load_state_dict` accepts a
    torchvision vgg19 state_dict (keys `features.<i>.weight|bias`, optionally prefixed `vgg19.`) and
    ignores the classifier / unused tail."""

    CONV_IDX = (0, 2, 5, 7, 10, 12, 14, 16, 19, 21, 23, 25, 28)
    CHANNELS = (64, 64, 128, 128, 256, 256, 256, 256, 512, 512, 512, 512, 512)

    def __init__(self, max_pairs=8):
        super().__init__()
        self.max_pairs = max_pairs
        sizes, cin = [], 3
        for cout in self.CHANNELS:
            sizes.append((cout, cin))
            cin = cout
        self._shapes = sizes
        flat = []
        for cout, ci in sizes:
            w = torch.empty(cout, ci, 3, 3)
            nn.init.kaiming_normal_(w, mode="fan_out", nonlinearity="relu")
            flat += [w.reshape(-1), torch.zeros(cout)]
        self.register_buffer("flat", torch.cat(flat))
        self._handles = {}
        self._dirty = True

    # ---- state_dict in torchvision's naming --------------------------------------------------------
    def _views(self):
        off, out = 0, {}
        for i, (cout, ci) in zip(self.CONV_IDX, self._shapes):
            out[f"features.{i}.weight"] = self.flat[off: off + cout * ci * 9].view(cout, ci, 3, 3)
            off += cout * ci * 9
            out[f"features.{i}.bias"] = self.flat[off: off + cout]
            off += cout
        return out

    def state_dict(self, *args, **kw):
        return {k: v.detach().clone() for k, v in self._views().items()}

    def load_state_dict(self, sd, strict=True):
        views = self._views()
        seen = set()
        for k, v in sd.items():
            k = k[len("vgg19."):] if k.startswith("vgg19.") else k
            if k in views:
                views[k].copy_(torch.as_tensor(v).to(views[k].device, torch.float32))
                seen.add(k)
        missing = [k for k in views if k not in seen]
        if strict and missing:
            raise RuntimeError("VGG19Wrapper.load_state_dict: missing keys %s" % missing)
        self._dirty = True
        return missing

    def _handle(self, h, w, dev):
        key = (h, w, dev.index, torch.cuda.current_stream(dev).cuda_stream)
        if key not in self._handles:
            lib, ctx = B.lib(), B.get_ctx(dev)
            hd = C.c_void_p()
            B.check(lib.gi_vgg19_create(ctx, h, w, self.max_pairs, C.byref(hd)))
            if lib.gi_vgg19_param_floats(hd) != self.flat.numel():
                raise B.BackendError("VGG19Wrapper: parameter layout mismatch")
            ws = torch.empty(lib.gi_vgg19_workspace_bytes(hd) + 256, dtype=torch.uint8, device=dev)
            off = (-ws.data_ptr()) % 256
            B.check(lib.gi_vgg19_bind(hd, B.ptr(self.flat), ws.data_ptr() + off, ws.numel() - off))
            self._handles[key] = (hd, ws)
            self._dirty = True
        hd = self._handles[key][0]
        if self._dirty:
            for h2, _ in self._handles.values():
                B.check(B.lib().gi_vgg19_sync_weights(h2))
            self._dirty = False
        return hd

    def _check(self, x):
        if not x.is_cuda or x.dtype != torch.float32 or x.dim() != 4 or x.shape[1] != 1:
            raise B.BackendError("VGG19Wrapper takes (n,1,H,W) float32 tensors on the gfx950 device")
        if self.flat.device != x.device:
            raise B.BackendError("VGG19Wrapper parameters live on %s, input on %s: call .to(device)" % (self.flat.device, x.device))
        if x.shape[0] > self.max_pairs:
            raise B.BackendError("VGG19Wrapper(max_pairs=%d) got a batch of %d" % (self.max_pairs, x.shape[0]))

    @torch.no_grad()
    def perceptual_and_style(self, output, target, weight_p, weight_s, per_tap=False):
        """(weight_p * sum_taps mse(F_o, F_t), weight_s * sum_taps mse(G_o, G_t)) as device scalars."""
        self._check(output)
        self._check(target)
        output, target = output.detach().contiguous(), target.detach().contiguous()
        hd = self._handle(output.shape[2], output.shape[3], output.device)
        out = torch.empty(2, dtype=torch.float32, device=output.device)
        taps = torch.empty(10, dtype=torch.float32, device=output.device) if per_tap else None
        B.check(B.lib().gi_vgg19_perceptual_style(hd, B.ptr(output), B.ptr(target), output.shape[0], float(weight_p), float(weight_s),
                                                  B.ptr(out), B.ptr(taps)))
        return (out[0], out[1], taps) if per_tap else (out[0], out[1])

    @torch.no_grad()
    def features(self, x, tap):
        """Feature map of tap 0..4 (relu1_1 .. relu5_1) as (n,C,h,w) float32."""
        self._check(x)
        x = x.detach().contiguous()
        hd = self._handle(x.shape[2], x.shape[3], x.device)
        c = (64, 128, 256, 512, 512)[tap]
        s = 1 << tap
        out = torch.empty(x.shape[0], c, x.shape[2] // s, x.shape[3] // s, dtype=torch.float32, device=x.device)
        B.check(B.lib().gi_vgg19_features(hd, B.ptr(x), x.shape[0], tap, B.ptr(out)))
        return out

    def forward(self, x):
        raise NotImplementedError("the classifier head of vgg19 is not part of the loss path (reference loss.py uses features only)")

    def __del__(self):
        try:
            for hd, _ in self._handles.values():
                B.lib().gi_vgg19_destroy(hd)
        except Exception:
            pass
