"""GPU tier: the fused SSIM kernel (gi_ssim through gan_inpainting_amd.lib.pytorch_ssim) against the
fixtures recorded from the imported reference's lib/pytorch_ssim, the oracle at the benchmark size,
and the metric's own properties (ssim(x,x) = 1, symmetry, per-sample mean = overall mean)."""
import numpy as np
import pytest
import torch

from oracle import params as op
from oracle import torch_ref as orc
from util_golden import load

pytestmark = pytest.mark.gpu

# fp32 reference vs its own fp64 evaluation differ by <= 2e-7 on these cases (fixture `mean64_*`); the
# kernel applies the window separably and sums tiles in fp64: 2e-6 absolute on a value in [0,1].
TOL = 2e-6


def _mod():
    import gan_inpainting_amd  # noqa: F401
    from gan_inpainting_amd.lib import pytorch_ssim
    return pytorch_ssim


def test_ssim_matches_reference_fixtures():
    ps = _mod()
    fx = load("ssim")
    for i, (seed, n, c, h, w, ws) in enumerate(fx["cases"].tolist()):
        x, y = (torch.from_numpy(a).cuda() for a in op.synth_ssim_pair(seed, n, c, h, w))
        m = ps.ssim(x, y, window_size=ws)
        per = ps.ssim(x, y, window_size=ws, size_average=False)
        m2 = ps.SSIM(window_size=ws)(x, y)
        assert m.shape == () and per.shape == (n,)
        assert abs(float(m) - float(fx[f"mean_{i}"])) <= TOL, (i, float(m), float(fx[f"mean_{i}"]))
        assert abs(float(m) - float(fx[f"mean64_{i}"])) <= TOL
        assert np.abs(per.cpu().numpy() - fx[f"per_sample_{i}"]).max() <= TOL, i
        assert float(m2) == float(m)                                       # deterministic
        assert abs(float(ps.ssim(x, x, window_size=ws)) - 1.0) <= TOL
        assert abs(float(ps.ssim(y, x, window_size=ws)) - float(m)) <= TOL  # symmetric


@pytest.mark.parametrize("shape,ws", [((32, 1, 256, 256), 11), ((4, 1, 512, 512), 11), ((3, 2, 100, 37), 31), ((5, 1, 7, 5), 11)])
def test_ssim_vs_oracle_sizes(shape, ws):
    """BASELINE sizes (256^2 bs=32; 512^2), the largest window, an image smaller than the window."""
    ps = _mod()
    x, y = op.synth_ssim_pair(77, *shape)
    tx, ty = torch.from_numpy(x), torch.from_numpy(y)
    ref = orc.ssim(tx.double(), ty.double(), ws, size_average=False)
    got = ps.ssim(tx.cuda(), ty.cuda(), window_size=ws, size_average=False).cpu().double()
    assert (got - ref).abs().max().item() <= TOL
    assert abs(float(ps.ssim(tx.cuda(), ty.cuda(), window_size=ws)) - float(ref.mean())) <= TOL


def test_ssim_rejects_bad_input():
    ps = _mod()
    from gan_inpainting_amd import backend as B
    x = torch.rand(1, 1, 16, 16, device="cuda")
    with pytest.raises(B.BackendError):
        ps.ssim(x, x, window_size=10)          # even window: output size would differ (reference pads ws//2)
    with pytest.raises(B.BackendError):
        ps.ssim(x.requires_grad_(), x.detach())
    with pytest.raises(B.BackendError):
        ps.ssim(x.detach().cpu(), x.detach().cpu())
