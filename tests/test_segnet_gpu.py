"""GPU tier: the frozen face-parsing network of BASELINE config 5 (UnetGenerator(1,4,7,ngf=32).eval(),
train.py:171-175) - multi-channel Tanh head, running-statistics BatchNorm, input gradient only, ngf=32
zero-embedded in the ngf=64 kernels - against the fixture recorded from the imported reference
(tests/golden/make_golden.py case_segnet) and against the oracle for a wide (ngf=64) 4-channel net."""
import functools

import numpy as np
import pytest
import torch

from oracle import params as op
from oracle import torch_ref as orc
from util_golden import load
from gpu_util import rel_l2

pytestmark = pytest.mark.gpu

W_CE = [0, 1.2, 0.7, 0.7]


def _mods():
    import gan_inpainting_amd  # noqa: F401
    from gan_inpainting_amd.lib.models import loss, networks
    return loss, networks


@pytest.mark.parametrize("dtype", ["fp32", "fp16"])
def test_face_parsing_net_vs_reference_fixture(dtype):
    loss, networks = _mods()
    fx = load("segnet")
    seed, N, HW = int(fx["seed"]), int(fx["N"]), int(fx["HW"])
    P = op.make_unet_params(seed, num_downs=7, ngf=32, in_c=1, out_c=4)
    net = networks.UnetGenerator(1, 4, 7, ngf=32, norm_layer=functools.partial(torch.nn.BatchNorm2d, affine=True, track_running_stats=True),
                                 use_dropout='False', dtype=dtype)
    assert isinstance(net, networks.EmbeddedUnetGenerator)
    net.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in P.items()}, strict=True)
    sd = net.state_dict()
    assert set(sd) == set(P) and all(torch.equal(sd[k].cpu(), torch.from_numpy(np.array(P[k]))) for k in P)   # narrow shapes round-trip
    net = net.cuda().eval()
    for p in net.parameters():
        p.requires_grad_(False)
    if dtype == "fp16":
        net.phys.set_loss_scale(1024.0)
    g, _ = op.synth_batch(seed + 1, N, HW, HW)
    labels, _ = op.synth_segmentation(seed + 2, N, 4, HW, HW)
    x = torch.from_numpy(g).cuda().requires_grad_(True)
    y = net(x)
    assert tuple(y.shape) == (N, 4, HW, HW)
    ce = loss.CrossEntropyLoss(weight=W_CE)(y, torch.from_numpy(labels).cuda())
    (0.01 * ce).backward()
    ref_y = torch.from_numpy(fx["out_full"].astype(np.float32))
    ref_g = torch.from_numpy(fx["xgrad_full"])
    ey, eg = rel_l2(y.detach().cpu(), ref_y), rel_l2(x.grad.cpu(), ref_g)
    print(f"{dtype}: out rel L2 {ey:.2e}  ce {float(ce):.6f} vs {float(fx['ce']):.6f}  input-grad rel L2 {eg:.2e}")
    # the fixture stores the output in fp16 (size): 5e-4 floor; fp16 compute: 11-bit activations through 14 layers
    assert ey <= (1e-3 if dtype == "fp32" else 5e-3)
    assert abs(float(ce) - float(fx["ce"])) <= (1e-4 if dtype == "fp32" else 2e-3) * float(fx["ce"])
    assert eg <= (1e-3 if dtype == "fp32" else 3e-2)
    if dtype == "fp32":
        assert np.abs(x.grad[0, 0, :6, :8].cpu().numpy() - fx["xgrad_head"]).max() <= 2e-3 * np.abs(fx["xgrad_head"]).max()


def test_wide_multichannel_eval_and_train_forward_vs_oracle():
    """ngf=64, 3 output channels, num_downs=6 at 64x64: eval forward + input gradient, and a train-mode forward
    (batch statistics) of the multi-channel head, against the oracle."""
    _, networks = _mods()
    P = op.make_unet_params(123, num_downs=6, ngf=64, in_c=1, out_c=3)
    net = networks.UnetGenerator(1, 3, 6, ngf=64, use_dropout=False, dtype="fp32")
    net.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in P.items()}, strict=True)
    net = net.cuda().eval()
    for p in net.parameters():
        p.requires_grad_(False)
    g, _ = op.synth_batch(124, 2, 64, 64)
    R = torch.from_numpy(np.random.Generator(np.random.PCG64(125)).standard_normal((2, 3, 64, 64)).astype(np.float32))
    x = torch.from_numpy(g).cuda().requires_grad_(True)
    y = net(x)
    (y * R.cuda()).sum().backward()
    TP = orc.to_torch(P, requires_grad=False)
    xo = torch.from_numpy(g).requires_grad_(True)
    yo = orc.unet_forward(TP, xo, 6, False, None)
    (yo * R).sum().backward()
    assert rel_l2(y.detach().cpu(), yo.detach()) <= 1e-5
    assert rel_l2(x.grad.cpu(), xo.grad) <= 1e-4
    net.train()
    with torch.no_grad():
        yt = net(torch.from_numpy(g).cuda())
    TP2 = orc.to_torch(P, requires_grad=False)
    # the oracle always applies the dropout of level 5 in train mode (u * mask * 2): a constant 0.5 "mask" disables it
    yto = orc.unet_forward(TP2, torch.from_numpy(g), 6, True, {5: torch.full((2, 512, 4, 4), 0.5)})
    assert rel_l2(yt.cpu(), yto) <= 1e-4


def test_multichannel_and_narrow_inference_forward_vs_oracle():
    """Inference path (BatchNorm folded into the convolutions) of the multi-channel head, wide (ngf=64, 3 channels) and
    embedded narrow (ngf=32, 4 classes: the face-parsing network called under no_grad by the evaluation pass)."""
    _, networks = _mods()
    for ngf, out_c, nd, hw, dtype, tol in ((64, 3, 6, 64, "fp32", 1e-5), (32, 4, 7, 128, "fp16", 2e-2)):
        P = op.make_unet_params(321 + ngf, num_downs=nd, ngf=ngf, in_c=1, out_c=out_c)
        net = networks.UnetGenerator(1, out_c, nd, ngf=ngf, use_dropout=False, dtype=dtype)
        net.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in P.items()}, strict=True)
        net = net.cuda().eval()
        g, _ = op.synth_batch(322 + ngf, 2, hw, hw)
        with torch.no_grad():
            y = net(torch.from_numpy(g).cuda())                       # eval mode, no autograd node: inference
        yo = orc.unet_forward(orc.to_torch(P, requires_grad=False), torch.from_numpy(g), nd, False, None)
        assert rel_l2(y.cpu(), yo) <= tol, (ngf, rel_l2(y.cpu(), yo))


def test_parameter_gradients_of_multichannel_net_are_refused():
    _, networks = _mods()
    from gan_inpainting_amd import backend as B
    net = networks.UnetGenerator(1, 2, 6, ngf=64, use_dropout=False, dtype="fp32").cuda().train()
    x = torch.rand(1, 1, 64, 64, device="cuda")
    y = net(x)
    with pytest.raises(B.BackendError):
        y.sum().backward()
