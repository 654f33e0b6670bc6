"""CPU tier: oracle/kink.py, the kink-aware gradient reference the GPU parity tests use, checked on the oracle itself: the
fp32 oracle plays the implementation under test, its own kink decisions are read from its taps."""
import torch

from oracle import kink


def _worst(g_a, dx_a, g_b, dx_b):
    rel = lambda a, b: float((a.double() - b).abs().max() / (b.abs().max() + 1e-300))   # noqa: E731
    return max([rel(dx_a, dx_b)] + [rel(g_a[k], g_b[k]) for k in g_b])


def test_one_flipped_unit_breaks_the_plain_comparison_and_the_kink_reference_repairs_it():
    """Seed 101 of the 64x64 generator case: the oracle's fp32 and fp64 evaluations take different sides of ONE LeakyReLU
    kink and whole gradient tensors differ by 1e-1 of their max; with that one decision handed over they agree to 1e-5."""
    torch.set_num_threads(8)
    case = kink.unet_case(101, 6, 2, 64)
    taps = {}
    _, g32, dx32, _ = kink.run(case, torch.float32, taps)
    dec = {k: v.detach() > 0 for k, v in taps.items() if not k.endswith(".keep")}
    assert sorted(dec) == sorted(kink.tap_shapes(case)) and all(tuple(dec[k].shape) == kink.tap_shapes(case)[k][0] for k in dec)
    _, g64, dx64, _ = kink.run(case, torch.float64)
    assert _worst(g32, dx32, g64, dx64) > 1e-2
    _, g, dx, rep = kink.kink_reference(case, dec)
    assert rep["flipped"] >= 1 and rep["outside"] == 0 and rep["at_risk"] < 1e-3 * rep["units"]
    assert _worst(g32, dx32, g, dx) < 1e-4


def test_a_wrong_decision_outside_the_band_is_reported():
    torch.set_num_threads(8)
    case = kink.patchgan_case(264, 64, 2, True)
    taps = {}
    kink.run(case, torch.float64, taps, backward=False)
    dec = {k: (v.detach() > 0).clone() for k, v in taps.items()}
    z = taps["c2"].detach()
    idx = tuple(int(i) for i in (z.abs() == z.abs().max()).nonzero()[0])
    dec["c2"][idx] = ~dec["c2"][idx]          # a sign error on the largest pre-activation of conv2's BatchNorm output
    _, _, _, rep = kink.kink_reference(case, dec)
    assert rep["outside"] == 1 and rep["outside_worst"] > 1e3 and rep["flipped"] == 0
