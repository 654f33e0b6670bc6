"""Functional CPU restatement of the reference hot path on torch-CPU primitives.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py). fp32 (or fp64 on request), NCHW,
torch.nn.functional ops + autograd for the backward passes.

Every function cites the reference lines it follows (paths under the upstream repo).
Pinned against the imported reference by tests/golden (tests/test_oracle_golden.py).
"""
import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

from . import params as _p

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


def to_torch(P, dtype=torch.float32, requires_grad=True):
    """numpy state dict -> torch leaf tensors (float params require grad, buffers do not)."""
    out = OrderedDict()
    for k, v in P.items():
        t = torch.from_numpy(np.array(v))
        if k.endswith("num_batches_tracked"):
            out[k] = t.clone()
            continue
        t = t.to(dtype).clone()
        if requires_grad and not (k.endswith("running_mean") or k.endswith("running_var")):
            t.requires_grad_(True)
        out[k] = t
    return out


def named_parameter_keys(P):
    """Keys that nn.Module.named_parameters() would yield, in order."""
    return [k for k in P if not (k.endswith("running_mean") or k.endswith("running_var")
                                 or k.endswith("num_batches_tracked"))]


def _bn(P, prefix, x, train):
    # nn.BatchNorm2d(affine=True, track_running_stats=True): networks.py:38, :288-290, :341
    if train:
        P[prefix + ".num_batches_tracked"] += 1
    return F.batch_norm(x, P[prefix + ".running_mean"], P[prefix + ".running_var"],
                        P[prefix + ".weight"], P[prefix + ".bias"], train, BN_MOMENTUM, BN_EPS)


def _norm(P, prefix, x, train, norm):
    """The norm_layer of a block (networks.py:30-45): BatchNorm2d(affine, running statistics), InstanceNorm2d(affine=False,
    track_running_stats=False: instance statistics in train AND eval mode, eps 1e-5) or Identity."""
    if norm == "batch":
        return _bn(P, prefix, x, train)
    if norm == "instance":
        return F.instance_norm(x, eps=BN_EPS)
    return x


def _act(z, slope, flip=None):
    """LeakyReLU(slope) / ReLU (slope 0). flip (bool tensor like z, tools/pick_kink_safe_seeds.py only): the units whose
    kink decision is inverted - the value is continuous there (z is within rounding distance of 0), the derivative is
    the other branch's. flip=None is the reference's own call."""
    if flip is None:
        return F.leaky_relu(z, slope) if slope else F.relu(z)
    pos = (z > 0) ^ flip
    return torch.where(pos, z, slope * z)


def unet_forward(P, x, num_downs=7, train=True, dropout_masks=None, norm="batch", taps=None, flips=None):
    """UnetGenerator.forward (networks.py:246-253) through the recursive
    UnetSkipConnectionBlock.forward (networks.py:320-324), unrolled.

    Quirks reproduced:
      * downrelu is LeakyReLU(0.2, inplace=True) applied to the block input, so the tensor that
        is concatenated as the skip is leaky_relu(x), not x (networks.py:287, :324);
      * uprelu (in place) then acts on the whole concat (networks.py:289);
      * use_dropout='False' is truthy: Dropout(0.5) after upnorm of levels 5..num_downs-1
        (networks.py:18-19, :313-314).
    dropout_masks: {level: keep-mask tensor (N,C,H,W) of 0/1}; required in train mode for
    those levels (the oracle never draws its own random numbers).
    taps: optional dict; receives every tensor that feeds a LeakyReLU / ReLU kink, keyed 'd<k>' (input of level k+1's
    downrelu = output of level k's down path) and 'u<k>' (level k's up-path output before dropout, one half of the next
    uprelu's input; the other half is the skip, whose kinks are those of 'd<k-1>'), together with the keep-mask under
    'u<k>.keep' where dropout follows (tools/pick_kink_safe_seeds.py). flips: {tap name: bool tensor}, see _act.
    """
    flips = flips or {}
    keys = _p.unet_key_layout(num_downs)
    drops = _p.dropout_levels(num_downs)
    skips = {}
    h = F.conv2d(x, P[keys[0]["down"] + ".weight"], P.get(keys[0]["down"] + ".bias"), stride=2, padding=1)
    for k in range(2, num_downs + 1):
        if taps is not None:
            taps[f"d{k - 1}"] = h
        s = _act(h, 0.2, flips.get(f"d{k - 1}"))
        skips[k - 1] = s
        h = F.conv2d(s, P[keys[k - 1]["down"] + ".weight"], P.get(keys[k - 1]["down"] + ".bias"), stride=2, padding=1)
        if keys[k - 1]["dnorm"]:
            h = _norm(P, keys[k - 1]["dnorm"], h, train, norm)
    u = h
    if taps is not None:
        taps[f"d{num_downs}"] = h
    for k in range(num_downs, 0, -1):
        if flips:   # the skip's ReLU decides like its LeakyReLU did (same sign); the decoder half has its own kinks
            fu = flips.get(f"d{num_downs}") if k == num_downs else flips.get(f"u{k + 1}")
            inp = _act(u, 0.0, fu) if k == num_downs else torch.cat([_act(skips[k], 0.0, flips.get(f"d{k}")), _act(u, 0.0, fu)], 1)
        else:
            inp = F.relu(u) if k == num_downs else F.relu(torch.cat([skips[k], u], 1))
        bias = P.get(keys[k - 1]["up"] + ".bias")
        u = F.conv_transpose2d(inp, P[keys[k - 1]["up"] + ".weight"], bias, stride=2, padding=1)
        if keys[k - 1]["unorm"]:
            u = _norm(P, keys[k - 1]["unorm"], u, train, norm)
        if taps is not None and k > 1:
            taps[f"u{k}"] = u
        if train and k in drops:
            if taps is not None:
                taps[f"u{k}.keep"] = dropout_masks[k]
            u = u * dropout_masks[k].to(u.dtype) * 2.0
    return torch.tanh(u)


def patchgan_forward(P, x, sigmoid=True, train=True, taps=None, flips=None):
    """PatchGANDiscriminator.forward (networks.py:335-363); Linear(25,1) generalised to
    Linear((H/16-3)*(W/16-3),1) for sizes other than 128x128."""
    h = F.conv2d(x, P["model.0.weight"], None, stride=2, padding=1)
    if taps is not None:   # the tensors that feed a LeakyReLU kink (tools/pick_kink_safe_seeds.py)
        taps["c1"] = h
    flips = flips or {}
    h = _act(h, 0.2, flips.get("c1"))
    for i, (conv, bn) in enumerate(((2, 3), (5, 6), (8, 9))):
        h = F.conv2d(h, P[f"model.{conv}.weight"], None, stride=2, padding=1)
        h = _bn(P, f"model.{bn}", h, train)
        if taps is not None:
            taps[f"c{i + 2}"] = h
        h = _act(h, 0.2, flips.get(f"c{i + 2}"))
    h = F.conv2d(h, P["model.11.weight"], None, stride=1, padding=0)
    h = h.reshape(h.shape[0], -1)
    h = F.linear(h, P["model.13.weight"], P["model.13.bias"])
    if sigmoid:
        h = torch.sigmoid(h)
    return h.view(-1, 1)


# ---------------------------------------------------------------------------------------------
# mask pipeline and losses
# ---------------------------------------------------------------------------------------------

def mask_pipeline(ground, mask, ceil=True):
    """experiment_list/minimaxgan_l1.py:114-117: mask=ceil(mask); masked = ground*(1-mask)."""
    m = torch.ceil(mask) if ceil else mask
    return m, ground * (1 - m)


def composite(masked, gen_out, mask):
    """minimaxgan_l1.py:122: inpainted = masked + inpainted*mask."""
    return masked + gen_out * mask


def l1_loss(a, b):
    return torch.mean(torch.abs(a - b))  # nn.L1Loss, minimaxgan_l1.py:62,166


def rmse_loss(yhat, y, eps=1e-16):
    return torch.sqrt(torch.mean((yhat - y) ** 2) + eps)  # lib/models/loss.py:11-19


def local_loss(yhat, y, mask, base="l1"):
    """lib/models/loss.py:24-47 as literally coded: sum(base(y*m, yhat*m)) / count(m != 0);
    the sqrt branch never fires because self.loss is never an RMSELoss instance (:30-31,:44).
    base='rmse' is the evident intent (sqrt of the masked mean square + eps), kept as a
    labelled extension."""
    d = y * mask - yhat * mask
    cnt = (mask != 0).to(d.dtype).sum()
    if base == "l1":
        return torch.abs(d).sum() / cnt
    if base == "mse":
        return (d * d).sum() / cnt
    if base == "rmse":
        return torch.sqrt((d * d).sum() / cnt + 1e-16)
    raise ValueError(base)


def ssim_window(window_size=11, sigma=1.5):
    """1-D normalised Gaussian, fp32 (lib/pytorch_ssim/__init__.py:10-12: python-float exp, fp32 tensor,
    divided by its fp32 sum)."""
    import math
    g = torch.tensor([math.exp(-(x - window_size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(window_size)],
                     dtype=torch.float32)
    return g / g.sum()


def ssim(img1, img2, window_size=11, size_average=True):
    """SSIM of two (n,c,h,w) batches (lib/pytorch_ssim/__init__.py:14-40,68-76): depthwise
    window_size^2 Gaussian (outer product of the 1-D window, zero padding window_size//2) of x, y,
    x^2, y^2, xy; C1=0.01^2, C2=0.03^2; mean over everything or per sample."""
    n, c, h, w = img1.shape
    g = ssim_window(window_size).to(img1.dtype).unsqueeze(1)
    win = g.mm(g.t()).unsqueeze(0).unsqueeze(0).expand(c, 1, window_size, window_size).contiguous()
    p = window_size // 2
    conv = lambda t: F.conv2d(t, win, padding=p, groups=c)
    mu1, mu2 = conv(img1), conv(img2)
    mu1_sq, mu2_sq, mu1_mu2 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    s1 = conv(img1 * img1) - mu1_sq
    s2 = conv(img2 * img2) - mu2_sq
    s12 = conv(img1 * img2) - mu1_mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    m = ((2 * mu1_mu2 + C1) * (2 * s12 + C2)) / ((mu1_sq + mu2_sq + C1) * (s1 + s2 + C2))
    return m.mean() if size_average else m.mean(1).mean(1).mean(1)


def eval_recon_batch(ground, gen, mask, is_flip_mask=False):
    """One batch of lib/models/evaluate.py:127-158 after the network call: m = ceil(mask) (flipped if
    asked), out = gen*m + ground*(1-m); returns (out, rmse_global, l1_global, rmse_local, l1_local) with
    RMSELoss (loss.py:11-19), nn.L1Loss, LocalLoss(nn.L1Loss) (loss.py:24-47) and, for the local RMSE,
    the evident intent sqrt(masked mean square + eps) (the reference's `LocalLoss(loss.RMSELoss())`
    raises at construction: it passes an instance where the class is called)."""
    m = torch.ceil(mask)
    if is_flip_mask:
        m = 1 - m
    masked = ground * (1 - m)
    out = gen * m + masked
    return (out, rmse_loss(ground, out), l1_loss(ground, out), local_loss(ground, out, m, base="rmse"),
            local_loss(ground, out, m, base="l1"))


def segmentation_eval_metric(labels, outputs, unique_labels):
    """lib/models/evaluate.py:179-224: prediction = argmax over dim 1; per class u and per sample
    intersection / union / |p| / |g| pixel counts -> iou, precision, recall (eps 1e-32), mean over the
    batch; across-class = mean over the classes. A label value of -1 matches every class (the reference
    marks matches with -1 in place, :193-195)."""
    prediction = torch.argmax(outputs, 1)
    eps = 1e-32
    batch = labels.shape[0]
    per_class, across = {}, {"precision": 0, "recall": 0, "iou": 0}
    for u in unique_labels:
        g = ((labels.view(batch, -1) == u) | (labels.view(batch, -1) == -1)).long()
        p = (prediction.view(batch, -1) == u).long()
        inter = (p & g).sum(1).float()
        union = (p | g).sum(1).float()
        iou = inter / (union + eps)
        precision = inter / (p.sum(1) + eps)
        recall = inter / (g.sum(1) + eps)
        per_class[u] = {"precision": precision.mean(), "recall": recall.mean(), "iou": iou.mean()}
        for k in across:
            across[k] = across[k] + per_class[u][k]
    for k in across:
        across[k] = across[k] / len(unique_labels)
    return per_class, across


VGG_TAPS = (1, 6, 11, 20, 29)     # relu1_1, relu2_1, relu3_1, relu4_1, relu5_1 (lib/models/loss.py:53,73,93)


def vgg19_tap_features(P, x, last=29):
    """Feature maps after features[1,6,11,20,29] of a VGG-19 `features` stack with parameters P
    (torchvision keys) applied to the grey batch x repeated over 3 channels (loss.py:54-61)."""
    from . import params as op
    h = x.repeat(1, 3, 1, 1)
    taps, i = [], 0
    for v in op.VGG19_CFG:
        if v == "M":
            h = F.max_pool2d(h, 2, 2)
            i += 1
        else:
            h = F.relu(F.conv2d(h, P[f"features.{i}.weight"], P[f"features.{i}.bias"], padding=1))
            if i + 1 in VGG_TAPS:
                taps.append(h)
            i += 2
        if i > last:
            break
    return taps


def gram_matrix(features, normalize=True):
    """loss.py:117-136."""
    N, C, H, W = features.shape
    f = features.reshape(N, C, -1)
    g = torch.bmm(f, f.transpose(1, 2))
    return g / (H * W * C) if normalize else g


def perceptual_and_style_loss(P, output, target, weight_p=0.05, weight_s=100):
    """loss.py:93-115 (no gradient: the reference runs it under no_grad on detached inputs). Also
    returns the per-tap terms."""
    with torch.no_grad():
        fo, ft = vgg19_tap_features(P, output.detach()), vgg19_tap_features(P, target.detach())
        p_terms = [torch.mean(torch.pow(a - b, 2)) for a, b in zip(fo, ft)]
        s_terms = [torch.mean(torch.pow(gram_matrix(a) - gram_matrix(b), 2)) for a, b in zip(fo, ft)]
        return weight_p * sum(p_terms), weight_s * sum(s_terms), p_terms, s_terms


def tv_loss(img, tv_weight):
    """loss.py:138-151."""
    w_variance = torch.mean(torch.pow(img[:, :, :, :-1] - img[:, :, :, 1:], 2))
    h_variance = torch.mean(torch.pow(img[:, :, :-1, :] - img[:, :, 1:, :], 2))
    return tv_weight * (h_variance + w_variance)


def weighted_cross_entropy(logits, labels, weight):
    """nn.CrossEntropyLoss(weight=w)(logits, labels) as called at
    wgan_perceptual_style_faceparsing.py:67-68,212-213."""
    return F.cross_entropy(logits, labels, weight=weight)


class _BCE(torch.autograd.Function):
    """nn.BCELoss as ATen computes it: forward with both log terms clamped at -100, backward
    (p - t) / max(p (1 - p), 1e-12) / n (so p = 0 or 1 gives finite gradients where differentiating the clamped
    logs would give 0 * inf)."""

    @staticmethod
    def forward(ctx, p, target):
        ctx.save_for_backward(p, target)
        lp = torch.clamp(torch.log(p), min=-100.0)
        l1p = torch.clamp(torch.log(1 - p), min=-100.0)
        return -(target * lp + (1 - target) * l1p).mean()

    @staticmethod
    def backward(ctx, g):
        p, target = ctx.saved_tensors
        return g * (p - target) / torch.clamp(p * (1 - p), min=1e-12) / p.numel(), None


def bce_loss(p, target):
    """nn.BCELoss (minimaxgan_l1.py:61,135,141,162): log terms clamped at -100."""
    return _BCE.apply(p, target)


def mse_loss(p, target):
    return torch.mean((p - target) ** 2)  # LSGAN, experiment1_global_local_D.py:119,162


def gradient_penalty(PD, real, fake, eps, lam=10.0, sigmoid=False):
    """WGAN-GP extension (NOT in the reference): lam*E[(||grad_x D(x_hat)||_2-1)^2],
    x_hat = eps*real + (1-eps)*fake, eps (N,1,1,1)."""
    xhat = (eps * real + (1 - eps) * fake).detach().requires_grad_(True)
    d = patchgan_forward(PD, xhat, sigmoid=sigmoid, train=True)
    (g,) = torch.autograd.grad(d.sum(), xhat, create_graph=True)
    norm = g.reshape(g.shape[0], -1).norm(2, dim=1)
    return lam * ((norm - 1) ** 2).mean()


# ---------------------------------------------------------------------------------------------
# optimizers (same update rules as the torch built-ins the plugins call)
# ---------------------------------------------------------------------------------------------

class Adam:
    """torch.optim.Adam(lr=2e-4, betas=(0.5,0.999), eps=1e-8): minimaxgan_l1.py:64-65."""

    def __init__(self, params, lr=2e-4, betas=(0.5, 0.999), eps=1e-8):
        self.params, self.lr, self.b1, self.b2, self.eps = list(params), lr, betas[0], betas[1], eps
        self.m = [torch.zeros_like(p) for p in self.params]
        self.v = [torch.zeros_like(p) for p in self.params]
        self.t = 0

    def zero_grad(self):
        for p in self.params:
            p.grad = None

    @torch.no_grad()
    def step(self):
        self.t += 1
        bc1 = 1 - self.b1 ** self.t
        bc2 = 1 - self.b2 ** self.t
        for p, m, v in zip(self.params, self.m, self.v):
            if p.grad is None:
                continue
            g = p.grad
            m.mul_(self.b1).add_(g, alpha=1 - self.b1)
            v.mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
            denom = (v.sqrt() / math.sqrt(bc2)).add_(self.eps)
            p.addcdiv_(m, denom, value=-self.lr / bc1)


class RMSprop:
    """torch.optim.RMSprop(lr=5e-5) defaults alpha=0.99, eps=1e-8: wgan_l1.py:64-65."""

    def __init__(self, params, lr=5e-5, alpha=0.99, eps=1e-8):
        self.params, self.lr, self.alpha, self.eps = list(params), lr, alpha, eps
        self.sq = [torch.zeros_like(p) for p in self.params]

    def zero_grad(self):
        for p in self.params:
            p.grad = None

    @torch.no_grad()
    def step(self):
        for p, sq in zip(self.params, self.sq):
            if p.grad is None:
                continue
            g = p.grad
            sq.mul_(self.alpha).addcmul_(g, g, value=1 - self.alpha)
            p.addcdiv_(g, sq.sqrt().add_(self.eps), value=-self.lr)


def clamp_params(params, lo=-0.01, hi=0.01):
    """wgan_l1.py:151-153: every D parameter (BN affine and Linear included)."""
    with torch.no_grad():
        for p in params:
            p.clamp_(lo, hi)


def trainable(P):
    return [P[k] for k in named_parameter_keys(P)]


def set_requires_grad(P, flag):
    """lib/models/util.py:19-22."""
    for k in named_parameter_keys(P):
        P[k].requires_grad_(flag)


def grad_absmean(P):
    """minimaxgan_l1.py:180-182: mean(|grad|) of every parameter whose name has no 'bias'."""
    return OrderedDict((k, float(P[k].grad.abs().mean())) for k in named_parameter_keys(P)
                       if "bias" not in k and P[k].grad is not None)


# ---------------------------------------------------------------------------------------------
# step sequences (re-typed from the plugin sources, read as text)
# ---------------------------------------------------------------------------------------------

def minimax_step(PG, PD, optG, optD, ground, mask, num_downs, dropout_masks, recon="l1"):
    """One batch of experiment_list/minimaxgan_l1.py:110-173 (recon='rmse': minimaxgan_rmse.py)."""
    m, masked = mask_pipeline(ground, mask)
    inpainted = composite(masked, unet_forward(PG, masked, num_downs, True, dropout_masks), m)
    set_requires_grad(PD, True)
    optD.zero_grad()
    n = ground.shape[0]
    d_real = patchgan_forward(PD, ground, True, True).view(-1)
    d_loss_real = bce_loss(d_real, torch.ones(n, dtype=d_real.dtype))
    d_loss_real.backward()
    d_fake = patchgan_forward(PD, inpainted.detach(), True, True).view(-1)
    d_loss_fake = bce_loss(d_fake, torch.zeros(n, dtype=d_real.dtype))
    d_loss_fake.backward()
    d_grad_stats = grad_absmean(PD)
    optD.step()
    set_requires_grad(PD, False)
    optG.zero_grad()
    d_fake2 = patchgan_forward(PD, inpainted, True, True).view(-1)
    g_adv = bce_loss(d_fake2, torch.ones(n, dtype=d_real.dtype))
    rec = l1_loss(ground, inpainted) if recon == "l1" else rmse_loss(ground, inpainted)
    g_loss = g_adv + rec
    g_loss.backward()
    g_grad_stats = grad_absmean(PG)
    optG.step()
    return dict(d_loss_real=float(d_loss_real.detach()), d_loss_fake=float(d_loss_fake.detach()), g_adv=float(g_adv.detach()),
                recon=float(rec.detach()), g_loss=float(g_loss.detach()), inpainted=inpainted.detach(),
                g_grad_absmean=g_grad_stats, d_grad_absmean=d_grad_stats)


def wgan_step(PG, PD, optG, optD, ground, mask, num_downs, dropout_masks, update_g, recon="l1",
              clip=0.01, extra=None):
    """One batch of experiment_list/wgan_l1.py:110-186 with one=+1, mone=-1 (the file's
    torch.FloatTensor(1) is uninitialised memory). update_g is the
    cadence decision of wgan_l1.py:157-163, taken by the caller."""
    m, masked = mask_pipeline(ground, mask)
    inpainted = composite(masked, unet_forward(PG, masked, num_downs, True, dropout_masks), m)
    set_requires_grad(PD, True)
    optD.zero_grad()
    d_real = patchgan_forward(PD, ground, False, True)
    d_fake = patchgan_forward(PD, inpainted.detach(), False, True)
    d_loss_real = d_real.mean().view(1)
    d_loss_real.backward(torch.ones(1, dtype=d_real.dtype))
    d_loss_fake = d_fake.mean().view(1)
    d_loss_fake.backward(-torch.ones(1, dtype=d_real.dtype))
    d_grad_stats = grad_absmean(PD)
    optD.step()
    clamp_params(trainable(PD), -clip, clip)
    out = dict(d_loss_real=float(d_loss_real), d_loss_fake=float(d_loss_fake),
               d_loss=float(d_loss_real - d_loss_fake), inpainted=inpainted.detach(),
               d_grad_absmean=d_grad_stats)
    if update_g:
        set_requires_grad(PD, False)
        optG.zero_grad()
        d_fake2 = patchgan_forward(PD, inpainted, False, True).view(-1)
        g_adv = d_fake2.mean().view(1)
        rec = l1_loss(ground, inpainted) if recon == "l1" else rmse_loss(ground, inpainted)
        g_loss = g_adv + rec
        if extra is not None:      # config-5 terms: callable(inpainted, ground, mask_c) -> (loss tensor, {name: float})
            ex, named = extra(inpainted, ground, m)
            g_loss = g_loss + ex
            out.update(named)
        g_loss.backward()
        out.update(g_adv=float(g_adv), recon=float(rec), g_loss=float(g_loss),
                   g_grad_absmean=grad_absmean(PG))
        optG.step()
    return out


def config5_extra(PV, PS, segment, weight_p=0.01, weight_s=0.01, weight_fp=0.01, tv_weight=1.0):
    """The generator-loss terms wgan_perceptual_style_faceparsing.py adds to g_adv + recon_global (:206-222):
    recon_local (LocalLoss RMSE, :207), face parsing (:212-213; PS = parameters of the frozen eval-mode
    UnetGenerator(1,4,7,ngf=32)), perceptual + style (:216; PV = VGG-19 parameters; constants) and tv (:219).
    Use with wgan_step(recon='rmse', extra=...)."""
    def fn(inpainted, ground, m):
        rl = local_loss(ground, inpainted, m, base="rmse")
        tv = tv_loss(inpainted, tv_weight)
        named = dict(recon_local=float(rl), tv=float(tv))
        tot = rl + tv
        if PS is not None:
            fp = weight_fp * weighted_cross_entropy(unet_forward(PS, inpainted, 7, False, None), segment,
                                                    torch.tensor([0, 1.2, 0.7, 0.7], dtype=inpainted.dtype))
            tot = tot + fp
            named["face_parsing"] = float(fp)
        if PV is not None:
            p, s, _, _ = perceptual_and_style_loss(PV, inpainted, ground, weight_p, weight_s)
            tot = tot + p + s
            named["perceptual"], named["style"] = float(p), float(s)
        return tot, named
    return fn


def wgan_update_g(batch_index, g_iter_count, update_g_every=5):
    """wgan_l1.py:157-163."""
    period = 140 if (g_iter_count < 25 or g_iter_count % 500 == 0) else update_g_every
    return batch_index % period == 0 and batch_index > 0


def dual_d_step(PG, PDg, PDl, optG, optD, ground, mask, num_downs, dropout_masks, lam1=300.0,
                lam2=300.0):
    """One batch of experiment_list/experiment1_global_local_D.py:139-200: G step first, LSGAN
    losses, mask NOT ceil-ed, output NOT composited; one Adam over both discriminators (:123)."""
    masked = ground * (1 - mask)
    n = ground.shape[0]
    set_requires_grad(PDg, False)
    set_requires_grad(PDl, False)
    optG.zero_grad()
    inpainted = unet_forward(PG, masked, num_downs, True, dropout_masks)
    pg = patchgan_forward(PDg, inpainted, True, True).view(-1)
    pl = patchgan_forward(PDl, mask * inpainted, True, True).view(-1)
    ones, zeros = torch.ones(n, dtype=pg.dtype), torch.zeros(n, dtype=pg.dtype)
    g_adv_g, g_adv_l = mse_loss(pg, ones), mse_loss(pl, ones)
    rec_g = rmse_loss(ground, inpainted)
    rec_l = rmse_loss(mask * ground, mask * inpainted)
    g_loss = g_adv_g + g_adv_l + lam1 * rec_g + lam2 * rec_l
    g_loss.backward()
    g_grad_stats = grad_absmean(PG)
    optG.step()
    set_requires_grad(PDg, True)
    set_requires_grad(PDl, True)
    optD.zero_grad()
    fake = inpainted.detach()
    d_loss = (mse_loss(patchgan_forward(PDl, ground * mask, True, True).view(-1), ones)
              + mse_loss(patchgan_forward(PDl, fake * mask, True, True).view(-1), zeros)
              + mse_loss(patchgan_forward(PDg, ground, True, True).view(-1), ones)
              + mse_loss(patchgan_forward(PDg, fake, True, True).view(-1), zeros))
    d_loss.backward()
    optD.step()
    return dict(g_loss=float(g_loss), d_loss=float(d_loss), rmse_global=float(rec_g),
                rmse_local=float(rec_l), g_adv_global=float(g_adv_g), g_adv_local=float(g_adv_l),
                inpainted=fake, g_grad_absmean=g_grad_stats)
