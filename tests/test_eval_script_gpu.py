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


def test_local_mse_script_matches_oracle(tmp_path):
    """local_mse.py mirror: masked-region MSE of the raw generator output per checkpoint, train and test sets."""
    pytest.importorskip("PIL")
    from gan_inpainting_amd import local_mse as lm
    root = tmp_path / "data"
    df, G, M = _dataset(root, 6, 128)
    (root / "csv").mkdir()
    df.to_csv(root / "csv" / "train_all_masks.csv", index=False)
    df.iloc[:3].to_csv(root / "csv" / "test_all_masks.csv", index=False)
    exp_root = tmp_path / "model" / "wgan_rmse"
    (exp_root / "run1").mkdir(parents=True)
    P = {20: op.make_unet_params(41), 5: op.make_unet_params(42)}
    for ep, p in P.items():
        torch.save({k: torch.from_numpy(np.array(v)) for k, v in p.items()}, exp_root / "run1" / f"epoch{ep}_G.pt")
    metric, path = lm.main(["--exp-root", str(exp_root), "--data", str(root), "--imagedim", "128", "--batchsize", "3", "--dtype", "fp32",
                            "--eval-mode", "--seed", "1"])
    assert list(metric["test"]) == [5, 20] and list(metric["train"]) == [5, 20]      # sorted by epoch
    with open(path, "rb") as f:
        assert pickle.load(f) == metric

    def oracle(p, idx_batches):
        PT = {k: torch.from_numpy(np.array(v)) for k, v in p.items()}
        tot = 0.0
        for idx in idx_batches:
            g = torch.from_numpy(G[idx].astype(np.float32) / 255.0)[:, None]
            m = torch.ceil(torch.from_numpy(M[idx].astype(np.float32) / 255.0)[:, None])
            y = orc.unet_forward(PT, g * (1 - m), 7, False, None)
            tot += float((((g * m) - (y * m)) ** 2).sum() / (m != 0).float().sum())
        return tot / len(idx_batches)
    for ep, p in P.items():
        want = oracle(p, [np.arange(0, 3)])
        assert abs(metric["test"][ep] - want) <= 1e-4 * max(1.0, abs(want)), (ep, metric["test"][ep], want)
    # the train loader is shuffled: whatever the grouping, a batch's sum/count is a weighted mean of its images'
    # ratios, so the reported mean over batches lies between the smallest and the largest per-image value
    per_image = [oracle(P[5], [np.array([i])]) for i in range(6)]
    assert min(per_image) * (1 - 1e-4) <= metric["train"][5] <= max(per_image) * (1 + 1e-4)
