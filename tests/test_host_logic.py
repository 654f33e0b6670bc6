"""CPU tier: host-side logic that needs no GPU: the WGAN G-update cadence, loss-scale selection,
parameter initialisation order (same torch RNG consumption as the reference constructors)."""
import numpy as np
import torch


def test_wgan_cadence_matches_reference_rule():
    import gan_inpainting_amd  # noqa: F401
    from gan_inpainting_amd import trainer
    from oracle import torch_ref as orc
    for g_iter in (0, 10, 24, 25, 26, 499, 500, 501, 1000):
        for bi in range(0, 300):
            assert trainer.wgan_update_g(bi, g_iter) == orc.wgan_update_g(bi, g_iter)
    assert not trainer.wgan_update_g(0, 100)             # never at batch 0 (wgan_l1.py:163)
    assert trainer.wgan_update_g(5, 100) and trainer.wgan_update_g(140, 3) and not trainer.wgan_update_g(5, 3)


def test_init_statistics_follow_torch_defaults():
    """kaiming_uniform(a=sqrt(5)) => U(-1/sqrt(fan_in), 1/sqrt(fan_in)); BatchNorm weight 1 / bias 0."""
    import gan_inpainting_amd  # noqa: F401
    from gan_inpainting_amd.lib.models import networks
    torch.manual_seed(3)
    g = networks.get_network("generator", "unet")
    sd = g.state_dict()
    w = sd["model.model.1.model.1.weight"]          # Conv2d(64,128): fan_in 64*16
    b = 1.0 / np.sqrt(64 * 16)
    assert float(w.abs().max()) <= b + 1e-7 and abs(float(w.std()) - b / np.sqrt(3)) < 0.05 * b
    wt = sd["model.model.1.model.5.weight"]         # ConvTranspose2d(256,64): torch fan_in = out*16
    bt = 1.0 / np.sqrt(64 * 16)
    assert float(wt.abs().max()) <= bt + 1e-7
    assert float(sd["model.model.1.model.2.weight"].min()) == 1.0 and float(sd["model.model.1.model.2.bias"].abs().max()) == 0.0
    assert float(sd["model.model.1.model.2.running_var"].min()) == 1.0
    # same seed -> same weights (deterministic RNG consumption order)
    torch.manual_seed(3)
    g2 = networks.get_network("generator", "unet")
    assert torch.equal(g2.state_dict()["model.model.3.weight"], sd["model.model.3.weight"])


def test_default_init_equals_reference_under_same_seed():
    """tests/golden/init_parity.npz was recorded from the reference's get_network() under
    torch.manual_seed(7): the backend's modules reproduce every tensor (same RNG consumption order,
    same state_dict keys)."""
    import gan_inpainting_amd  # noqa: F401
    from gan_inpainting_amd.lib.models import networks
    from util_golden import load
    fx = load("init_parity")
    torch.manual_seed(int(fx["seed"]))
    g = networks.get_network("generator", "unet")
    d = networks.get_network("discriminator", "patchgan")
    for tag, net in (("g", g), ("d", d)):
        sd = net.state_dict()
        keys = [k for k in sd if not k.endswith("num_batches_tracked")]
        assert keys == [str(k) for k in fx[f"{tag}_keys"]]
        for i, k in enumerate(keys):
            assert abs(float(sd[k].double().sum()) - fx[f"{tag}_sum"][i]) <= 1e-9 * (1 + abs(fx[f"{tag}_sum"][i])), k
            assert abs(float(sd[k].double().abs().sum()) - fx[f"{tag}_abs"][i]) <= 1e-9 * (1 + fx[f"{tag}_abs"][i]), k
            head = sd[k].reshape(-1)[:4].numpy()
            assert np.array_equal(head, fx[f"{tag}_head"][i][:head.size]), k


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: no module of the shipped package may import it (bench.py's cpu_baseline
    leg, __graft_entry__.smoke and tests/ are the only allowed users)."""
    import os
    import re
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gan-inpainting_amd")
    pat = re.compile(r"^\s*(from|import)\s+oracle\b|^\s*from\s+\.+\s*import\s+oracle\b|importlib\.import_module\(\s*['\"]oracle")
    bad = []
    for d, _, files in os.walk(root):
        for f in files:
            if f.endswith(".py"):
                for i, line in enumerate(open(os.path.join(d, f), encoding="utf-8"), 1):
                    if pat.search(line):
                        bad.append(f"{os.path.join(d, f)}:{i}: {line.strip()}")
    assert not bad, bad


def test_offline_scripts_host_logic(tmp_path):
    """eval.py / local_mse.py mirrors: the reference's CLI names (eval.py:24-29), checkpoint discovery sorted by the
    number after 'epoch' across sub-directories (local_mse.py:43-54); no device needed for either."""
    import gan_inpainting_amd  # noqa: F401
    from gan_inpainting_amd import eval as ev
    from gan_inpainting_amd import local_mse as lm
    a = ev.build_parser().parse_args(["-m", "M", "-d", "D", "-f", "F"])
    assert (a.modelpath, a.datasetpath, a.csvfile, a.generator, a.batchsize, a.eval_mode) == ("M", "D", "F", "unet", 50, False)
    assert ev.build_parser().parse_args(["--modelpath", "M", "--datasetpath", "D", "--csvfile", "F", "-g", "vgg19"]).generator == "vgg19"
    for sub, name in (("a", "epoch100_G.pt"), ("a", "epoch20_G.pt"), ("b/c", "epoch3_G.pt"), ("b", "notes.txt")):
        d = tmp_path / sub
        d.mkdir(parents=True, exist_ok=True)
        (d / name).write_bytes(b"")
    found = lm.checkpoint_paths(str(tmp_path))
    assert [ep for ep, _ in found] == [3, 20, 100]
    assert all(p.endswith(f"epoch{ep}_G.pt") for ep, p in found)
    b = lm.build_parser().parse_args(["--exp-root", "R", "--data", "D"])
    assert (b.exp, b.imagedim, b.batchsize) == ("wgan_rmse", 128, 64)      # local_mse.py:40,56,69


def test_real_data_shards_have_equal_batch_counts():
    """train.py --data <dir> under torch.distributed.run: every rank must see the same number of full batches, or the rank
    with one batch more blocks forever in its gradient all-reduce (127 rows, 2 ranks, bs=32 used to give 2 vs 1)."""
    import pandas as pd
    import gan_inpainting_amd  # noqa: F401
    from gan_inpainting_amd import train
    import pytest
    with pytest.raises(ValueError, match="zero batches"):      # fewer rows than one batch per rank: loud, not an empty epoch
        train.shard_rows(pd.DataFrame({"groundtruth_source": [f"g{i}" for i in range(63)]}), 0, 2, 32)
    for rows, world, bs in ((127, 2, 32), (1000, 8, 32), (64, 2, 32), (5000, 3, 7)):
        df = pd.DataFrame({"groundtruth_source": [f"g{i}" for i in range(rows)]})
        shards = [train.shard_rows(df, r, world, bs) for r in range(world)]
        counts = {len(s) // bs for s in shards}
        assert len(counts) == 1 and all(len(s) % bs == 0 for s in shards), (rows, world, bs, [len(s) for s in shards])
        assert counts.pop() == rows // (world * bs)
        seen = sorted(v for s in shards for v in s["groundtruth_source"])
        assert len(seen) == len(set(seen))               # disjoint


def test_norm_layer_argument_forms():
    import functools
    import pytest
    from torch import nn
    import gan_inpainting_amd  # noqa: F401
    from gan_inpainting_amd.lib.models import networks
    """The constructor's norm_layer argument as the reference reads it (networks.py:270-273): a functools.partial or the bare
    class; use_bias follows `func == nn.InstanceNorm2d`. Anything else get_norm_layer cannot return is refused."""
    g = networks.UnetGenerator(1, 1, 7, ngf=64, norm_layer=nn.InstanceNorm2d)
    assert g.norm_kind == 1 and "model.model.1.model.1.bias" in g.state_dict() and not any("running_mean" in k for k in g.state_dict())
    g = networks.UnetGenerator(1, 1, 7, ngf=64, norm_layer=networks.get_norm_layer("none"))
    assert g.norm_kind == 2 and [k for k in g.state_dict() if k.endswith(".bias")] == ["model.model.3.bias"]
    g = networks.UnetGenerator(1, 1, 7, ngf=64, norm_layer=functools.partial(nn.BatchNorm2d, affine=True, track_running_stats=True))
    assert g.norm_kind == 0
    with pytest.raises(NotImplementedError):
        networks.UnetGenerator(1, 1, 7, ngf=64, norm_layer=functools.partial(nn.InstanceNorm2d, affine=True))
    with pytest.raises(NotImplementedError):
        networks.UnetGenerator(1, 1, 7, ngf=64, norm_layer=nn.GroupNorm)
    # same RNG consumption as the reference constructor: weight then bias per convolution, innermost block first
    torch.manual_seed(5)
    a = networks.UnetGenerator(1, 1, 5, ngf=64, norm_layer=networks.get_norm_layer("instance")).state_dict()
    torch.manual_seed(5)
    inner_down = nn.Conv2d(512, 512, 4, 2, 1, bias=True)
    inner_up = nn.ConvTranspose2d(512, 512, 4, 2, 1, bias=True)
    assert torch.equal(a["model.model.1.model.3.model.3.model.3.model.1.weight"], inner_down.weight.detach())
    assert torch.equal(a["model.model.1.model.3.model.3.model.3.model.1.bias"], inner_down.bias.detach())
    assert torch.equal(a["model.model.1.model.3.model.3.model.3.model.3.bias"], inner_up.bias.detach())


def test_padded_level1_inventory():
    """gi_unet_create_padded (inventory-only handle): ngf = 32 with level 1 computed 64 channels wide - the tensors that touch
    level 1 report the padded shapes, every other tensor the narrow network's (UnetGenerator(1, 4, 7, ngf=32), train.py:171-172)."""
    import ctypes as C
    import gan_inpainting_amd  # noqa: F401
    from gan_inpainting_amd import backend as B
    from gan_inpainting_amd.lib.models import networks
    net = networks.UnetGenerator(1, 4, 7, ngf=32, use_dropout="False")
    assert isinstance(net, networks.EmbeddedUnetGenerator) and isinstance(net.phys, networks._PaddedUnetGenerator)
    phys = {k: tuple(v.shape) for k, v in net.phys.state_dict().items()}
    narrow = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    assert phys["model.model.0.weight"] == (64, 1, 4, 4) and narrow["model.model.0.weight"] == (32, 1, 4, 4)
    assert phys["model.model.1.model.1.weight"] == (64, 64, 4, 4) and narrow["model.model.1.model.1.weight"] == (64, 32, 4, 4)
    assert phys["model.model.3.weight"] == (128, 4, 4, 4) and narrow["model.model.3.weight"] == (64, 4, 4, 4)
    differ = sorted(k for k in narrow if phys[k] != narrow[k])
    assert differ == sorted(["model.model.0.weight", "model.model.1.model.1.weight", "model.model.1.model.5.weight", "model.model.1.model.6.weight",
                             "model.model.1.model.6.bias", "model.model.1.model.6.running_mean", "model.model.1.model.6.running_var",
                             "model.model.3.weight"]), differ
    # a narrow state_dict round-trips through the padded network
    sd = {k: torch.randn(v.shape) if v.dtype.is_floating_point else v for k, v in net.state_dict().items()}
    net.load_state_dict(sd)
    back = net.state_dict()
    for k, v in sd.items():
        assert torch.equal(back[k].cpu(), v), k
    h = C.c_void_p()
    assert B.lib().gi_unet_create_padded(None, 7, 32, 48, 4, 0, 0.0, 128, 128, 1, B.GI_F16, 1, C.byref(h)) != 0   # ch1 must be a multiple of 64
