"""MI355X-native (gfx950) compute backend for the GAN-inpainting hot path.

Host side mirrors the reference's Python surface (lib.models.networks / loss / util, the
experiment_list `begin(state, loaders)` plugins); all arithmetic runs in libganinpaint.so
(hand-written HIP, C-ABI in include/ganinpaint.h). There is no CPU fallback: every op raises if
the library or a gfx950 device is missing.
"""
from . import backend  # noqa: F401

__all__ = ["backend"]
