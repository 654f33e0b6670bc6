"""Deterministic build-side weights for the oracle, the golden fixtures and the tests.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Key names and shapes are exactly those of the reference modules' state_dict()
(lib/models/networks.py:216-324 UnetGenerator, :331-363 PatchGANDiscriminator), so the
same dict can be fed to the reference via load_state_dict (done in
tests/golden/make_golden.py) and to the HIP backend.

Values come from numpy's PCG64 stream so they are bit-identical on every machine;
bounds follow PyTorch's default conv init (kaiming_uniform a=sqrt(5) => U(-1/sqrt(fan_in), +)),
BatchNorm affine/running stats are randomised so that every term of the
normalisation is exercised.
"""
from collections import OrderedDict

import numpy as np


def unet_channels(num_downs, ngf, in_c=1, out_c=1):
    """Return per-level (1-based) conv/upconv channel tuples.

    level k conv:   cin_k -> ch_k           (networks.py:285)
    level k upconv: (2*ch_k or ch_k innermost) -> cout_k   (networks.py:293-309)
    """
    ch = [None]
    for k in range(1, num_downs + 1):
        ch.append(ngf * min(2 ** (k - 1), 8))
    levels = []
    for k in range(1, num_downs + 1):
        cin = in_c if k == 1 else ch[k - 1]
        up_in = ch[k] if k == num_downs else 2 * ch[k]
        up_out = out_c if k == 1 else ch[k - 1]
        levels.append(dict(k=k, cin=cin, ch=ch[k], up_in=up_in, up_out=up_out))
    return levels


def unet_key_layout(num_downs):
    """state_dict key prefixes of every level, mirroring the recursive nesting of
    UnetSkipConnectionBlock (networks.py:296-318)."""
    out = []
    prefix = "model.model"
    for k in range(1, num_downs + 1):
        if k == 1:  # outermost: [downconv, sub, uprelu, upconv, tanh]
            d = dict(down=f"{prefix}.0", sub=f"{prefix}.1", up=f"{prefix}.3", dnorm=None, unorm=None)
        elif k == num_downs:  # innermost: [downrelu, downconv, uprelu, upconv, upnorm]
            d = dict(down=f"{prefix}.1", sub=None, up=f"{prefix}.3", dnorm=None, unorm=f"{prefix}.4")
        else:  # middle: [downrelu, downconv, downnorm, sub, uprelu, upconv, upnorm(, dropout)]
            d = dict(down=f"{prefix}.1", sub=f"{prefix}.3", up=f"{prefix}.5", dnorm=f"{prefix}.2",
                     unorm=f"{prefix}.6")
        out.append(d)
        if d["sub"] is not None:
            prefix = d["sub"] + ".model"
    return out


def dropout_levels(num_downs):
    """Levels whose up path ends in Dropout(0.5): the `num_downs-5` ngf*8 blocks
    (networks.py:237-238; use_dropout='False' is a truthy string, networks.py:18-19)."""
    return list(range(5, num_downs))


def synth_dropout_masks(seed, num_downs, N, H, W, ngf=64):
    """Deterministic keep-masks {level: uint8 (N,C,h,w)} with keep probability 0.5 for the Dropout(0.5) levels: level k's
    up path ends with ngf*min(2^(k-2), 8) channels at (H, W) >> (k-1). Imposed on both sides by the parity tests whose
    seeds are chosen from the oracle alone (tools/pick_kink_safe_seeds.py)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = {}
    for k in dropout_levels(num_downs):
        c = ngf * min(2 ** (k - 2), 8)
        out[k] = (rng.random(size=(N, c, H >> (k - 1), W >> (k - 1)), dtype=np.float32) < 0.5).astype(np.uint8)
    return out


def _uniform(rng, shape, bound):
    return ((rng.random(size=shape, dtype=np.float32) * 2.0 - 1.0) * np.float32(bound)).astype(np.float32)


def _bn(rng, P, prefix, c):
    P[f"{prefix}.weight"] = (0.5 + rng.random(size=(c,), dtype=np.float32)).astype(np.float32)
    P[f"{prefix}.bias"] = _uniform(rng, (c,), 0.2)
    P[f"{prefix}.running_mean"] = _uniform(rng, (c,), 0.1)
    P[f"{prefix}.running_var"] = (0.5 + rng.random(size=(c,), dtype=np.float32)).astype(np.float32)
    P[f"{prefix}.num_batches_tracked"] = np.array(0, dtype=np.int64)


def make_unet_params(seed, num_downs=7, ngf=64, in_c=1, out_c=1, norm="batch"):
    """norm: 'batch' (get_network's choice, networks.py:18), 'instance' (get_norm_layer('instance'): InstanceNorm2d without
    affine parameters or running statistics, and every convolution gets a bias, networks.py:279-286,300-309) or 'none'
    (Identity norm layers, no biases but the outermost up-convolution's)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    P = OrderedDict()
    levels = unet_channels(num_downs, ngf, in_c, out_c)
    keys = unet_key_layout(num_downs)
    use_bias = norm == "instance"

    def emit(k):
        lv, ky = levels[k - 1], keys[k - 1]
        P[f"{ky['down']}.weight"] = _uniform(rng, (lv["ch"], lv["cin"], 4, 4), 1.0 / np.sqrt(lv["cin"] * 16))
        if use_bias:
            P[f"{ky['down']}.bias"] = _uniform(rng, (lv["ch"],), 1.0 / np.sqrt(lv["cin"] * 16))
        if ky["dnorm"] and norm == "batch":
            _bn(rng, P, ky["dnorm"], lv["ch"])
        if k < num_downs:
            emit(k + 1)
        # ConvTranspose2d weight is [in, out, kh, kw]; torch takes fan_in from dim 1
        P[f"{ky['up']}.weight"] = _uniform(rng, (lv["up_in"], lv["up_out"], 4, 4), 1.0 / np.sqrt(lv["up_out"] * 16))
        if k == 1 or use_bias:
            P[f"{ky['up']}.bias"] = _uniform(rng, (lv["up_out"],), 1.0 / np.sqrt(lv["up_out"] * 16))
        if ky["unorm"] and norm == "batch":
            _bn(rng, P, ky["unorm"], lv["up_out"])

    emit(1)
    return P


def patchgan_head_features(H, W):
    """Generalised head: reference hard-codes Linear(25,1) (networks.py:354), valid for 128x128."""
    return (H // 16 - 3) * (W // 16 - 3)


def make_patchgan_params(seed, H=128, W=128, in_c=1, nf=64):
    rng = np.random.Generator(np.random.PCG64(seed))
    P = OrderedDict()
    chans = [in_c, nf, nf * 2, nf * 4, nf * 8]
    conv_idx = [0, 2, 5, 8]
    bn_idx = [None, 3, 6, 9]
    for i in range(4):
        P[f"model.{conv_idx[i]}.weight"] = _uniform(rng, (chans[i + 1], chans[i], 4, 4), 1.0 / np.sqrt(chans[i] * 16))
        if bn_idx[i] is not None:
            _bn(rng, P, f"model.{bn_idx[i]}", chans[i + 1])
    P["model.11.weight"] = _uniform(rng, (1, chans[4], 4, 4), 1.0 / np.sqrt(chans[4] * 16))
    F = patchgan_head_features(H, W)
    P["model.13.weight"] = _uniform(rng, (1, F), 1.0 / np.sqrt(F))
    P["model.13.bias"] = _uniform(rng, (1,), 1.0 / np.sqrt(F))
    return P


def synth_batch(seed, N, H, W, fractional_edge=False):
    """Synthetic masked-image batch: ground ~ U[0,1), one axis-aligned
    rectangle of ones per image with h,w in [H/8, H/2]."""
    rng = np.random.Generator(np.random.PCG64(seed))
    ground = rng.random(size=(N, 1, H, W), dtype=np.float32)
    mask = np.zeros((N, 1, H, W), dtype=np.float32)
    for n in range(N):
        h = int(rng.integers(H // 8, H // 2 + 1))
        w = int(rng.integers(W // 8, W // 2 + 1))
        y0 = int(rng.integers(0, H - h + 1))
        x0 = int(rng.integers(0, W - w + 1))
        mask[n, 0, y0:y0 + h, x0:x0 + w] = 1.0
        if fractional_edge:  # values in (0,1) on the border, to exercise ceil()
            mask[n, 0, y0, x0:x0 + w] = 0.25
            mask[n, 0, y0:y0 + h, x0] = 0.5
    return ground, mask


def synth_ssim_pair(seed, n, c, h, w):
    """Inpainting-like pair in [0,1] for the SSIM metric: y = x with one block replaced by noise
    (what experiment1_global_local_D.py:209 feeds it: ground truth vs generator output)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    x = rng.random((n, c, h, w), dtype=np.float32)
    x = (x + np.roll(x, 1, 2) + np.roll(x, 1, 3) + np.roll(x, (1, 1), (2, 3))) / 4   # local variances differ
    y = x.copy()
    y[:, :, h // 4: h // 2, w // 3: 2 * w // 3] = rng.random((n, c, h // 2 - h // 4, 2 * w // 3 - w // 3), dtype=np.float32)
    return x.astype(np.float32), y.astype(np.float32)


def synth_segmentation(seed, n, num_classes, h, w, with_minus_one=False):
    """(labels int64 (n,h,w), logits fp32 (n,num_classes,h,w)) for the face-parsing metrics: blocky label
    maps, logits = noisy one-hot so that predictions agree with the labels on most pixels; one sample
    lacks the last class entirely (empty-union edge case) and ties are present (argmax takes the first)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    coarse = rng.integers(0, num_classes, size=(n, (h + 7) // 8, (w + 7) // 8))
    labels = np.repeat(np.repeat(coarse, 8, 1), 8, 2)[:, :h, :w].astype(np.int64)
    labels[0][labels[0] == num_classes - 1] = 0
    logits = rng.standard_normal((n, num_classes, h, w)).astype(np.float32)
    onehot = (labels[:, None] == np.arange(num_classes)[None, :, None, None]).astype(np.float32)
    logits += 1.5 * onehot
    logits[:, :, 0, :] = 0.25                      # exact ties along the first row
    if with_minus_one:
        labels[:, 1, :] = -1
    return labels, logits


# torchvision vgg19 "features" layout (configuration E), restated from the public architecture:
# conv3x3(+ReLU) channel counts with 'M' = 2x2 max pooling. Module indices 0..36.
VGG19_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512, "M"]


def vgg19_conv_indices():
    """features.<i> index of every convolution, in order (0, 2, 5, 7, 10, ...)."""
    idx, i = [], 0
    for v in VGG19_CFG:
        if v == "M":
            i += 1
        else:
            idx.append(i)
            i += 2
    return idx


def make_vgg19_params(seed, n_convs=16):
    """Deterministic stand-in for the (unavailable) pretrained weights: He-normal fan_out like
    torchvision's initialiser, small non-zero biases so the bias path is exercised. Keys as in
    torchvision's state_dict: features.<i>.weight / .bias."""
    rng = np.random.Generator(np.random.PCG64(seed))
    P, cin = {}, 3
    chans = [v for v in VGG19_CFG if v != "M"]
    for i, cout in zip(vgg19_conv_indices()[:n_convs], chans[:n_convs]):
        std = (2.0 / (cout * 9)) ** 0.5
        P[f"features.{i}.weight"] = (rng.standard_normal((cout, cin, 3, 3)) * std).astype(np.float32)
        P[f"features.{i}.bias"] = (rng.standard_normal(cout) * 0.05).astype(np.float32)
        cin = cout
    return P


def synth_u8_image(seed, h, w):
    """Photo-like uint8 grey image (smooth gradients + texture + saturated patches) for the input-pipeline tests."""
    rng = np.random.Generator(np.random.PCG64(seed))
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    img = 127 + 90 * np.sin(xx / (7.0 + seed % 5)) * np.cos(yy / 11.0) + rng.normal(0, 25, (h, w))
    img[: h // 5, : w // 4] = 255
    img[-(h // 6 + 1):, -(w // 7 + 1):] = 0
    return np.clip(img, 0, 255).astype(np.uint8)
