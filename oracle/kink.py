"""Kink-aware gradient references (test infrastructure, see oracle/__init__.py).

Why. The gradient tests compare whole gradient tensors of UnetGenerator / PatchGANDiscriminator
(lib/models/networks.py:216-363) at 1e-3 of max|ref| per tensor. Two correct fp32 evaluations of these graphs agree to
~1e-6 in the forward, but a ReLU / LeakyReLU input within that distance of zero can land on either side of its kink, and
with the tests' random-sign objective ONE such unit moves gradient tensors by 1e-3 .. 4e-1 of their max (measured unit by
unit with `flip_impacts` below: every single at-risk unit of a 64x64 case is worth more than 1e-3 somewhere). A case
with ~5e5 .. 3e7 activations has 0.3 .. 3 such units between ANY two fp32 evaluations, so no seed list can make a strict
comparison hold "by luck" on the larger sizes, and margins of 10 x the measured forward error exist on no seed at all
(counting argument in DESIGN.md section 2).

What is decidable from the oracle alone is WHERE a correct fp32 forward may disagree about a kink:
  eps(z)   = max |z32 - z64| over a tapped pre-activation tensor, the oracle's own fp32 forward error of that tensor;
  at risk  = |z64| < BAND * eps(z)  (never: exact zeros of both evaluations, and units removed by dropout).
The reference gradient of a case is then the fp64 oracle evaluated with, on the at-risk units ONLY, the side of the kink
the implementation under test actually took in its own forward (read back through gi_net_saved_activation; the value is
continuous there, only the derivative branch differs: torch_ref._act). Outside the band the oracle's own decisions
stand and the tests assert that the implementation took the same ones (a kink disagreement at |z| >= BAND * eps is a
wrong forward, not rounding). The comparison itself is strict on every tensor and runs on ONE fixed seed per case: no
retry, no seed chosen by looking at a result.

fp16 runs: the implementation's forward error is the half-precision one, so the band is FP16_BAND * max|z64| of the
tensor (the forward tolerance of those tests) instead of the measured fp32 error.
"""
import numpy as np
import torch

from . import params as _p
from . import torch_ref as _o

BAND = 16.0          # fp32: at risk when |z64| < BAND * max|z32 - z64| (BAND * eps ~ 1e-5 of the tensor's largest value)
FP16_BAND = 2e-2     # fp16: at risk when |z64| < FP16_BAND * max|z64|


def unet_channels(nd, ngf=64):
    return [0] + [ngf * min(2 ** (k - 1), 8) for k in range(1, nd + 1)]


def unet_case(seed, nd, N, HW, norm="batch"):
    """Inputs of one generator parity case: weights, masked image, objective weights R (loss = sum(y * R)), dropout masks."""
    P = _p.make_unet_params(seed, num_downs=nd, ngf=64, norm=norm)
    ground, mask = _p.synth_batch(seed + 7, N, HW, HW)
    x = torch.from_numpy(ground * (1 - mask))
    R = torch.from_numpy(np.random.Generator(np.random.PCG64(seed)).standard_normal(size=(N, 1, HW, HW), dtype=np.float32))
    masks = {k: torch.from_numpy(v) for k, v in _p.synth_dropout_masks(seed + 13, nd, N, HW, HW).items()}
    return {"kind": "unet", "P": P, "x": x, "R": R, "masks": masks, "nd": nd, "norm": norm, "N": N, "HW": HW}


def patchgan_case(seed, HW, N, sigmoid, groups=1):
    """groups = 2: the N images are two consecutive BatchNorm populations of N / 2, i.e. two critic calls on the same parameters
    (wgan_l1.py:134-135 D(ground), D(inpainted)) whose gradients accumulate; the second half is shifted and scaled so that the
    two populations have different statistics."""
    P = _p.make_patchgan_params(seed, HW, HW)
    ground, _ = _p.synth_batch(seed + 3, N, HW, HW)
    if groups == 2:
        ground[N // 2:] = ground[N // 2:] * 0.5 + 0.2
    r = torch.from_numpy(np.random.Generator(np.random.PCG64(seed + 5)).standard_normal(size=(N, 1), dtype=np.float32))
    return {"kind": "patchgan", "P": P, "x": torch.from_numpy(ground), "R": r, "sigmoid": sigmoid, "N": N, "HW": HW, "groups": groups}


def tap_shapes(case):
    """{tap name: ((N,C,h,w), kind, level)}: the tensors that feed a kink and where gi_net_saved_activation keeps the
    activation behind each (generator: d<k> -> kind 0 level k; u<k> -> kind 1 level k-1; critic: c<i> -> kind 0 level i)."""
    N, HW = case["N"], case["HW"]
    out = {}
    if case["kind"] == "unet":
        nd = case["nd"]
        ch = unet_channels(nd)
        for k in range(1, nd + 1):
            out[f"d{k}"] = ((N, ch[k], HW >> k, HW >> k), 0, k)
        for k in range(2, nd + 1):
            out[f"u{k}"] = ((N, ch[k - 1], HW >> (k - 1), HW >> (k - 1)), 1, k - 1)
    else:
        for i, c in ((1, 64), (2, 128), (3, 256), (4, 512)):
            out[f"c{i}"] = ((N, c, HW >> i, HW >> i), 0, i)
    return out


def run(case, dtype, taps=None, flips=None, backward=True):
    """One oracle evaluation of loss = sum(y * R) -> (y, {name: parameter gradient}, input gradient, parameters)."""
    OP = _o.to_torch(case["P"], dtype=dtype)
    x = case["x"].to(dtype).clone().requires_grad_(backward)
    with torch.set_grad_enabled(backward):
        if case["kind"] == "unet":
            y = _o.unet_forward(OP, x, case["nd"], True, case["masks"], norm=case["norm"], taps=taps, flips=flips)
        elif case.get("groups", 1) == 2:   # two calls, one per population, in the reference's order (running statistics move twice)
            h = case["N"] // 2
            ys, tt = [], [{}, {}]
            for g in range(2):
                fl = {k: v[g * h:(g + 1) * h] for k, v in flips.items()} if flips else None
                ys.append(_o.patchgan_forward(OP, x[g * h:(g + 1) * h], case["sigmoid"], True, taps=tt[g] if taps is not None else None, flips=fl))
            y = torch.cat(ys)
            if taps is not None:
                for k in tt[0]:
                    taps[k] = torch.cat([tt[0][k], tt[1][k]])
        else:
            y = _o.patchgan_forward(OP, x, case["sigmoid"], True, taps=taps, flips=flips)
        if backward:
            (y * case["R"].to(dtype)).sum().backward()
    grads = {k: OP[k].grad for k in _o.named_parameter_keys(case["P"])} if backward else None
    return y.detach(), grads, (x.grad if backward else None), OP


def survey(case, band=BAND, fp16=False):
    """The oracle's two forwards with taps -> {tap: dict(z64, live, at_risk, eps)} (no gradients)."""
    t32, t64 = {}, {}
    run(case, torch.float32, t32, None, backward=False)
    y64, _, _, _ = run(case, torch.float64, t64, None, backward=False)
    out = {}
    for name, z64 in t64.items():
        if name.endswith(".keep"):
            continue
        z64 = z64.detach()
        z32 = t32[name].detach().double()
        eps = float((z32 - z64).abs().max())
        live = ~((z64 == 0) & (z32 == 0))
        if name + ".keep" in t64:
            live &= t64[name + ".keep"].bool()
        width = FP16_BAND * float(z64.abs().max()) if fp16 else band * eps
        out[name] = {"z64": z64, "live": live, "eps": eps, "width": width, "at_risk": live & (z64.abs() < width)}
    return out


def kink_reference(case, decisions, band=BAND, fp16=False):
    """decisions: {tap: bool tensor (N,C,h,w), True where the implementation's saved activation is > 0}.
    -> (y64, grads64, dx64, report): the fp64 oracle with the implementation's decisions on the at-risk units; report
    counts the at-risk units, the decisions taken from the implementation and the disagreements OUTSIDE the band (with
    the largest |z64| / width among them), which a correct forward does not have."""
    sv = survey(case, band, fp16)
    flips, rep = {}, {"units": 0, "at_risk": 0, "flipped": 0, "outside": 0, "outside_worst": 0.0, "eps": {}}
    for name, s in sv.items():
        dec = decisions[name].to(s["z64"].device)
        disagree = (dec != (s["z64"] > 0)) & s["live"]
        inside = disagree & s["at_risk"]
        outside = disagree & ~s["at_risk"]
        rep["units"] += int(s["live"].sum())
        rep["at_risk"] += int(s["at_risk"].sum())
        rep["flipped"] += int(inside.sum())
        rep["eps"][name] = s["eps"]
        if outside.any():
            rep["outside"] += int(outside.sum())
            rep["outside_worst"] = max(rep["outside_worst"], float(s["z64"].abs()[outside].max()) / max(s["width"], 1e-300))
        if inside.any():
            flips[name] = inside
    y, g, dx, _ = run(case, torch.float64, None, flips or None)
    return y, g, dx, rep


def flip_impacts(case, band=BAND, limit=8):
    """Diagnostic: for up to `limit` at-risk units per tapped tensor, the largest relative change (max-norm over a gradient
    tensor / max|that tensor|) caused by inverting that ONE unit's kink decision. -> [(tap, index, |z64|/eps, impact, where)]"""
    sv = survey(case, band)
    _, g0, dx0, _ = run(case, torch.float64)
    out = []
    for name, s in sv.items():
        for idx in s["at_risk"].nonzero()[:limit]:
            f = torch.zeros_like(s["at_risk"])
            f[tuple(idx)] = True
            _, g, dx, _ = run(case, torch.float64, None, {name: f})
            worst, where = float((dx - dx0).abs().max() / dx0.abs().max()), "dx"
            for k in g0:
                v = float((g[k] - g0[k]).abs().max() / (g0[k].abs().max() + 1e-300))
                if v > worst:
                    worst, where = v, k
            out.append((name, tuple(idx.tolist()), float(s["z64"].abs()[tuple(idx)]) / max(s["eps"], 1e-300), worst, where))
    return out
