"""GPU tier: the device input transform (Resize + ToTensor of train.py:69-72) is bit-exact with Pillow's resampler
(fixture recorded from PIL.Image.resize, tests/golden/make_golden.py case_resize)."""
import numpy as np
import pytest
import torch

from oracle import params as op
from oracle import resize_ref as rr
from util_golden import load

pytestmark = pytest.mark.gpu


def _tf(size):
    import gan_inpainting_amd  # noqa: F401
    from gan_inpainting_amd.lib.data import dataset
    return dataset.DeviceResizeToTensor(size)


def test_resize_matches_pillow_fixture_bit_for_bit():
    fx = load("resize")
    for i, (seed, h, w, size) in enumerate(fx["cases"].tolist()):
        a = op.synth_u8_image(seed, h, w)
        tf = _tf(size)
        x = torch.from_numpy(a)[None].cuda()
        got_u8 = tf(x, return_bytes=True).cpu().numpy()[0, 0]
        ref = fx[f"out_{i}"]
        assert got_u8.shape == ref.shape == rr.resized_output_size(h, w, size)
        assert np.array_equal(got_u8, ref), (i, int((got_u8 != ref).sum()))
        got = tf(x).cpu().numpy()[0, 0]
        assert got.dtype == np.float32 and np.array_equal(got, ref.astype(np.float32) / np.float32(255.0))   # ToTensor


def test_resize_batch_and_large_vs_oracle():
    """A batch of 6 CelebA-sized images (218x178 -> 156x128) and one 2048x1536 photo -> 512, against the oracle."""
    tf = _tf(128)
    batch = np.stack([op.synth_u8_image(300 + i, 218, 178) for i in range(6)])
    got = tf(torch.from_numpy(batch).cuda(), return_bytes=True).cpu().numpy()[:, 0]
    assert np.array_equal(got, rr.resize_bilinear_u8(batch, 156, 128))
    big = op.synth_u8_image(400, 1536, 2048)
    got = _tf(512)(torch.from_numpy(big)[None].cuda()).cpu().numpy()[0, 0]
    assert np.array_equal(got, rr.resize_to_tensor(big, 512))


def test_resize_rejects_host_or_float_input():
    from gan_inpainting_amd import backend as B
    tf = _tf(64)
    with pytest.raises(B.BackendError):
        tf(torch.zeros((1, 8, 8), dtype=torch.uint8))
    with pytest.raises(B.BackendError):
        tf(torch.zeros((1, 8, 8), device="cuda"))
