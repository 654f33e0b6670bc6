"""GPU tier: the host mirror of the reference's eval.py (checkpoint -> generator -> outputs) against the CPU
oracle: `epoch{N}_G.pt` files written with torch.save(state_dict) are read back, the generator runs on the
dataset through the C-ABI, the pickles hold what the reference's script would hold."""
import os
import pickle

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import gan_inpainting_amd  # noqa: F401,E402
from oracle import params as op  # noqa: E402
from oracle import torch_ref as orc  # noqa: E402
from gpu_util import report  # noqa: E402


def _dataset(root, n, hw):
    from PIL import Image
    import pandas as pd
    (root / "img").mkdir(parents=True)
    rng = np.random.Generator(np.random.PCG64(5))
    rows, G, M = [], [], []
    for i in range(n):
        g = rng.integers(0, 256, (hw, hw), dtype=np.uint8)
        m = np.zeros((hw, hw), np.uint8)
        m[16 + 3 * i: 70 + 3 * i, 24: 90] = 255
        m[16 + 3 * i, 24: 90] = 100                                   # a fractional edge: eval.py applies no ceil
        Image.fromarray(g, mode="L").save(root / "img" / f"g{i}.png")
        Image.fromarray(m, mode="L").save(root / "img" / f"m{i}.png")
        rows.append({"groundtruth_source": f"img/g{i}.png", "mask_source": f"img/m{i}.png"})
        G.append(g), M.append(m)
    return pd.DataFrame(rows), np.stack(G), np.stack(M)


@pytest.mark.parametrize("csvname,dtype", [("test_all_masks.csv", "fp32"), ("extra.csv", "fp16")])
def test_eval_script_matches_oracle(tmp_path, csvname, dtype):
    pytest.importorskip("PIL")
    from gan_inpainting_amd import eval as ev
    n, hw, batch = 5, 128, 3                                          # two batches: 3 + 2, the last one is kept
    df, G, M = _dataset(tmp_path / "data", n, hw)
    csv = tmp_path / csvname
    df.to_csv(csv, index=False)
    models = tmp_path / "model"
    models.mkdir()
    P = {3: op.make_unet_params(31), 12: op.make_unet_params(32)}
    for ep, p in P.items():
        sd = {k: torch.from_numpy(np.array(v)) for k, v in p.items()}
        torch.save(sd, models / f"epoch{ep}_G.pt")
    (models / "notes.txt").write_text("not a checkpoint")
    out, (p_in, p_out) = ev.main(["-m", str(models), "-d", str(tmp_path / "data"), "-f", str(csv), "--batchsize", str(batch),
                                  "--dtype", dtype, "--eval-mode"])
    assert sorted(out) == [3, 12]
    ground = (G[3:].astype(np.float32) / 255.0)[:, None]
    mask = (M[3:].astype(np.float32) / 255.0)[:, None]
    masked = ground * mask if "extra" in csvname else ground * (1 - mask)
    with open(p_in, "rb") as f:
        saved = pickle.load(f)
    assert np.array_equal(saved["ground"], ground) and np.array_equal(saved["mask"], mask)
    assert np.array_equal(saved["masked"], masked)                     # pure products of the same floats: bit-exact
    with open(p_out, "rb") as f:
        saved_out = pickle.load(f)
    for ep, p in P.items():
        yo = orc.unet_forward({k: torch.from_numpy(np.array(v)) for k, v in p.items()}, torch.from_numpy(masked), 7, False, None)
        ok, msg = report(f"eval.py epoch {ep} {dtype}", out[ep], yo, 1e-4 if dtype == "fp32" else 2e-2)
        assert ok, msg
        assert torch.equal(saved_out[ep], out[ep]) and tuple(out[ep].shape) == (n - batch, 1, hw, hw)


def test_eval_script_train_mode_default(tmp_path):
    """The reference never calls .eval() (eval.py:77-94): batch statistics and dropout stay on."""
    pytest.importorskip("PIL")
    from gan_inpainting_amd import eval as ev
    df, G, M = _dataset(tmp_path / "data", 4, 128)
    csv = tmp_path / "test_all_masks.csv"
    df.to_csv(csv, index=False)
    models = tmp_path / "model"
    models.mkdir()
    torch.save({k: torch.from_numpy(np.array(v)) for k, v in op.make_unet_params(33).items()}, models / "epoch1_G.pt")
    a, _ = ev.main(["-m", str(models), "-d", str(tmp_path / "data"), "-f", str(csv), "--batchsize", "4"])
    b, _ = ev.main(["-m", str(models), "-d", str(tmp_path / "data"), "-f", str(csv), "--batchsize", "4", "--eval-mode"])
    assert torch.isfinite(a[1]).all() and float(a[1].abs().max()) <= 1.0
    # batch statistics + dropout (seeded per network handle, so repeatable) against running statistics: far apart
    assert float((a[1] - b[1]).abs().mean()) > 1e-2
