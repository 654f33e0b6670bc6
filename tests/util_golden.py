"""Helpers shared by the golden-fixture tests (data access only)."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def unpack_masks(fx, prefix=""):
    """{level: uint8 keep-mask tensor (N,C,H,W)} from the bit-packed fixture entries."""
    out = {}
    for k in fx.files:
        if k.startswith(prefix + "mask") and k.endswith("_shape"):
            lvl = int(k[len(prefix) + 4:-6])
            shape = tuple(int(v) for v in fx[k])
            n = int(np.prod(shape))
            bits = np.unpackbits(fx[f"{prefix}mask{lvl}_bits"])[:n].reshape(shape)
            out[lvl] = torch.from_numpy(bits.astype(np.uint8))
    return out


def relerr(a, b, floor=1e-6):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b) / (np.abs(b) + floor))) if a.size else 0.0
