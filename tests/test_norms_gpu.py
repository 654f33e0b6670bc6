"""GPU tier: the generator built with get_norm_layer('instance') and ('none') (lib/models/networks.py:29-45) - the two
norm layers besides BatchNorm2d that UnetGenerator's constructor accepts. InstanceNorm2d(affine=False,
track_running_stats=False): statistics per image and channel in train and eval mode, and every convolution carries a
bias (use_bias, networks.py:270-273); 'none': Identity layers. Against the fixtures recorded from the reference class
(tests/golden/unet128_instance.npz, unet128_none.npz: the reference's own dropout masks are imposed) and against the
oracle in fp32 and fp64 at a second size."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import gan_inpainting_amd  # noqa: F401,E402
from gan_inpainting_amd.lib.models import networks  # noqa: E402
from oracle import params as op  # noqa: E402
from oracle import torch_ref as orc  # noqa: E402
from gpu_util import close_to_either, rel_l2, report  # noqa: E402
from util_golden import load, unpack_masks  # noqa: E402

TOL_OUT = {"fp32": 1e-4, "fp16": 2e-2}
TOL_GRAD = {"fp32": 1e-3, "fp16": 6e-2}
TOL_GRAD_L2 = {"fp32": None, "fp16": 0.15}      # see tests/test_nets_gpu.py
# a bias in front of an InstanceNorm has the gradient sum_p dx_p = 0 exactly; what either implementation returns is the
# rounding residue of that sum (fp16: of a sum of fp16-rounded values; measured up to 2e-2 where a channel has 8 pixels).
# Bound: this fraction of the same convolution's weight-gradient magnitude
CANCEL = {"fp32": 1e-5, "fp16": 5e-2}


def sd(P):
    return {k: torch.from_numpy(np.array(v)) for k, v in P.items()}


def make(P, nd, norm, dtype):
    net = networks.UnetGenerator(1, 1, nd, ngf=64, norm_layer=networks.get_norm_layer(norm_type=norm), use_dropout="False", dtype=dtype)
    assert list(net.state_dict().keys()) == list(P.keys()), "state_dict layout differs from the reference's"
    net.load_state_dict(sd(P))
    net.set_loss_scale(1.0)
    return net.to("cuda").train()


def cancelled_biases(net, nd, norm):
    """names of the biases whose convolution is followed by an InstanceNorm (all but the outermost and innermost down
    convolutions and the outermost up convolution)"""
    if norm != "instance":
        return set()
    keep = set()
    lv = net._level_names()
    keep.add(lv[0][0][:-len("weight")] + "bias")       # d1: no norm behind it
    keep.add(lv[nd - 1][0][:-len("weight")] + "bias")  # innermost down convolution: no norm
    keep.add(lv[0][1][:-len("weight")] + "bias")       # head
    return {n for n, _ in net.named_parameters() if n.endswith(".bias")} - keep


@pytest.mark.parametrize("dtype", ["fp32", "fp16"])
@pytest.mark.parametrize("norm", ["instance", "none"])
def test_unet_other_norms_vs_golden(norm, dtype):
    fx = load(f"unet128_{norm}")
    seed, N, HW, nd = int(fx["seed"]), int(fx["N"]), int(fx["HW"]), int(fx["num_downs"])
    net = make(op.make_unet_params(seed, num_downs=nd, ngf=64, norm=norm), nd, norm, dtype)
    net.impose_dropout_masks(unpack_masks(fx))
    ground, mask = op.synth_batch(seed + 7, N, HW, HW)
    x = torch.from_numpy(ground * (1 - mask)).cuda().requires_grad_(True)
    R = torch.from_numpy(np.random.Generator(np.random.PCG64(seed + 99)).standard_normal(size=(N, 1, HW, HW), dtype=np.float32))
    y = net(x)
    (y * R.cuda()).sum().backward()
    ok, msg = report(f"unet128_{norm} {dtype} out vs reference", y.detach().cpu(), torch.from_numpy(fx["out"]), TOL_OUT[dtype])
    assert ok, msg
    dxr = torch.from_numpy(fx["dx"])
    if TOL_GRAD_L2[dtype] is None:
        ok, msg = report(f"unet128_{norm} {dtype} dx vs reference", x.grad.cpu(), dxr, TOL_GRAD[dtype])
        assert ok, msg
    else:
        assert rel_l2(x.grad.cpu(), dxr) <= TOL_GRAD_L2[dtype], rel_l2(x.grad.cpu(), dxr)
    names = [str(s) for s in fx["grad_names"]]
    prm = dict(net.named_parameters())
    assert names == list(prm.keys()), "named_parameters() order differs from the reference"
    zero = cancelled_biases(net, nd, norm)
    bad = []
    for i, n in enumerate(names):
        got, ref = float(prm[n].grad.abs().mean()), float(fx["grad_absmean"][i])
        if n in zero:
            wref = float(fx["grad_absmean"][names.index(n[:-len("bias")] + "weight")])
            print(f"{n}: cancelled bias gradient |got| {got:.3e} |ref| {ref:.3e} (weight gradient {wref:.3e})")
            if got > CANCEL[dtype] * wref + 10 * ref:
                bad.append(f"{n}: absmean {got:.3e} should vanish (weight gradient {wref:.3e})")
            continue
        if abs(got - ref) > TOL_GRAD[dtype] * abs(ref) + 1e-9:
            bad.append(f"{n}: absmean {got:.6g} vs {ref:.6g}")
        head, ref_head = prm[n].grad.reshape(-1)[:32].cpu(), torch.from_numpy(fx[f"ghead_{i}"])
        if dtype == "fp32" and float(ref_head.abs().max()) > 1e-6 and rel_l2(head, ref_head) > 5e-3:
            bad.append(f"{n}: first-32 gradient entries relL2 {rel_l2(head, ref_head):.3e}")
    assert not bad, "\n".join(bad)
    net.eval()
    with torch.no_grad():
        ev = net(x.detach())
    ok, msg = report(f"unet128_{norm} {dtype} eval out vs reference", ev.cpu(), torch.from_numpy(fx["eval_out"]), TOL_OUT[dtype])
    assert ok, msg


def _norm_case(norm, cfg, dtype, seed):
    """one seeded forward + backward (+ eval forward) against the oracle; returns the strict-tolerance violations"""
    nd, N, HW = cfg
    P = op.make_unet_params(seed, num_downs=nd, ngf=64, norm=norm)
    net = make(P, nd, norm, dtype)
    net.set_dropout_seed(seed)
    ground, mask = op.synth_batch(seed + 1, N, HW, HW)
    x0 = torch.from_numpy(ground * (1 - mask))
    R = torch.from_numpy(np.random.Generator(np.random.PCG64(seed + 2)).standard_normal(size=(N, 1, HW, HW), dtype=np.float32))
    x = x0.detach().clone().cuda().requires_grad_(True)
    y = net(x)
    (y * R.cuda()).sum().backward()
    masks = {k: v.cpu() for k, v in net.dropout_masks().items()}
    res = {}
    for dt in (torch.float32, torch.float64):
        OP = orc.to_torch(P, dtype=dt)
        xo = x0.detach().clone().to(dt).requires_grad_(True)
        out = orc.unet_forward(OP, xo, nd, True, masks, norm=norm)
        (out * R.to(dt)).sum().backward()
        res[dt] = (OP, out.detach(), xo.grad)
    ok, msg = close_to_either(f"{norm} {cfg} {dtype} out", y.detach().cpu(), res[torch.float32][1], res[torch.float64][1], TOL_OUT[dtype])
    assert ok, msg
    zero = cancelled_biases(net, nd, norm)
    prm = dict(net.named_parameters())
    for name, p in prm.items():      # never kink-sensitive: a loose bound on every gradient
        if name not in zero:
            l2 = rel_l2(p.grad.detach().cpu(), res[torch.float64][0][name].grad)
            assert l2 <= 0.25, f"{norm} {cfg} {dtype} seed {seed} grad {name}: relative L2 error {l2:.3e}"
    tol = TOL_GRAD[dtype]
    if nd == 8:
        # InstanceNorm over the 4 values of a 2x2 map (levels 7 and 8) is ill-conditioned: the oracle's own fp32 and fp64
        # evaluations differ by 2.5e-2 in dx and up to 2.9e-2 per weight gradient here (measured); bound = 3x that spread
        sp = lambda a, b: float((a.double() - b).abs().max() / (b.abs().max() + 1e-30))   # noqa: E731
        worst = max([sp(res[torch.float32][2], res[torch.float64][2])] +
                    [sp(res[torch.float32][0][n].grad, res[torch.float64][0][n].grad) for n in prm if n not in zero])
        tol = max(tol, 3.0 * worst)
        print(f"largest oracle fp32-vs-fp64 spread {worst:.3e} -> max-norm tol {tol:.3e}")
    bad = []
    ok, msg = close_to_either(f"{norm} {cfg} {dtype} dx", x.grad.cpu(), res[torch.float32][2], res[torch.float64][2], tol, TOL_GRAD_L2[dtype])
    if not ok:
        bad.append(msg)
    for name, p in prm.items():
        g = p.grad.detach().cpu()
        if name in zero:
            wmag = float(prm[name[:-len("bias")] + "weight"].grad.abs().mean())
            if float(g.abs().mean()) > CANCEL[dtype] * wmag + 10 * float(res[torch.float32][0][name].grad.abs().mean()):
                bad.append(f"{name}: {float(g.abs().mean()):.3e} should vanish (weight gradient {wmag:.3e})")
            continue
        ok, msg = close_to_either(f"{norm} {cfg} {dtype} grad {name}", g, res[torch.float32][0][name].grad, res[torch.float64][0][name].grad,
                                  tol, TOL_GRAD_L2[dtype])
        if not ok:
            bad.append(msg)
    # eval mode: InstanceNorm keeps using instance statistics, dropout is off
    net.eval()
    with torch.no_grad():
        ev = net(x0.cuda()).cpu()
        oev = orc.unet_forward(orc.to_torch(P, dtype=torch.float64), x0.double(), nd, False, None, norm=norm)
    ok, msg = report(f"{norm} {cfg} {dtype} eval out", ev, oev, TOL_OUT[dtype])
    assert ok, msg
    return bad


@pytest.mark.parametrize("dtype", ["fp32", "fp16"])
@pytest.mark.parametrize("norm,cfg", [("instance", (6, 3, 64)), ("none", (6, 3, 64)), ("instance", (7, 2, 256)), ("instance", (8, 5, 256))])
def test_unet_other_norms_vs_oracle(norm, cfg, dtype):
    """(8, 5, 256): two dropout levels drawn on the device (the masks are read back and handed to the oracle), odd batch.
    The strict gradient tolerance must hold on one of up to four seeds (a single ReLU / LeakyReLU kink flip between two
    correct fp32 evaluations moves gradient tensors by more than 1e-3 - even the oracle's own fp32 and fp64 evaluations
    differ by 4e-3 in dx on some seeds -, see tests/test_nets_gpu.py::test_unet_forward_backward_vs_oracle); the forward
    outputs and a loose bound on every gradient are asserted on every seed."""
    nd, N, HW = cfg
    failures = []
    for attempt in range(4):
        seed = 4100 + nd + N + 1000 * attempt
        bad = _norm_case(norm, cfg, dtype, seed)
        if not bad:
            return
        failures.append((seed, bad))
    assert False, "\n".join(f"seed {s}: " + " | ".join(b)[:3000] for s, b in failures)
