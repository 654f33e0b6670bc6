"""TEST INFRASTRUCTURE (oracle): CPU restatement of the input transform of the reference's loaders
(train.py:69-72: transforms.Compose([transforms.Resize(image_target_size), transforms.ToTensor()]) applied to
the PIL 'L' images of lib/data/dataset.py:6-12,41-48).

torchvision (absent here; pinned by the reference's environment) implements Resize(int) on a PIL image as
`img.resize((w, h), PIL.Image.BILINEAR)` with the smaller edge matched to `size`, and ToTensor as
uint8 -> float32 / 255. The arithmetic lives in Pillow (third-party dependency, libImaging/Resample.c,
Pillow >= 7: antialiased separable resampling in 22-bit fixed point, horizontal pass first, uint8 intermediate).
This file restates that published algorithm in numpy integers; tests/golden/make_golden.py pins it bit-for-bit
against Pillow itself (Pillow 12.2.0 is importable in the build container)."""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def resized_output_size(h, w, size):
    """torchvision.transforms.functional._compute_resized_output_size for an int `size`: the smaller edge becomes
    `size`, the other int(size * long / short)."""
    short, long_ = (w, h) if w <= h else (h, w)
    new_short, new_long = size, int(size * long_ / short)
    return (new_long, new_short) if w <= h else (new_short, new_long)      # (new_h, new_w)


def precompute_coeffs(in_size, out_size):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for the bilinear (triangle, support 1) filter over
    the full box [0, in_size). Returns (bounds[out_size][2] = (xmin, count), kk[out_size][ksize] int32, ksize)."""
    scale = filterscale = float(in_size) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = np.zeros(xmax, np.float64)
        ww = 0.0
        for x in range(xmax):
            t = (x + xmin - center + 0.5) * ss
            if t < 0.0:
                t = -t
            v = 1.0 - t if t < 1.0 else 0.0
            w[x] = v
            ww += v
        for x in range(xmax):
            if ww != 0.0:
                w[x] /= ww
            kk[xx, x] = int(-0.5 + w[x] * (1 << PRECISION_BITS)) if w[x] < 0 else int(0.5 + w[x] * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk, ksize


def _pass(img, bounds, kk, axis):
    """One resampling pass along `axis` of a uint8 array: clip8((2^21 + sum pixel*k) >> 22)."""
    img = np.moveaxis(img, axis, -1).astype(np.int64)
    out = np.empty(img.shape[:-1] + (bounds.shape[0],), np.uint8)
    for xx in range(bounds.shape[0]):
        xmin, cnt = int(bounds[xx, 0]), int(bounds[xx, 1])
        acc = (1 << (PRECISION_BITS - 1)) + (img[..., xmin:xmin + cnt] * kk[xx, :cnt].astype(np.int64)).sum(-1)
        out[..., xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, -1, axis)


def resize_bilinear_u8(img, out_h, out_w):
    """PIL.Image.resize((out_w, out_h), BILINEAR) of a uint8 (..., H, W) array: horizontal pass, then vertical
    (each only if that dimension changes, as ImagingResample does)."""
    h, w = img.shape[-2:]
    out = img
    if out_w != w:
        b, k, _ = precompute_coeffs(w, out_w)
        out = _pass(out, b, k, -1)
    if out_h != h:
        b, k, _ = precompute_coeffs(h, out_h)
        out = _pass(out, b, k, -2)
    return out


def resize_to_tensor(img_u8, size):
    """Resize(size) + ToTensor() of uint8 (..., H, W): float32 in [0,1], (..., new_h, new_w)."""
    nh, nw = resized_output_size(img_u8.shape[-2], img_u8.shape[-1], size)
    return resize_bilinear_u8(img_u8, nh, nw).astype(np.float32) / np.float32(255.0)
