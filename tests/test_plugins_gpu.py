"""GPU tier: the plugin surface (train.py CLI -> importlib -> begin(state, loaders)) runs end to end
for every accelerated experiment and writes the reference's checkpoint contract (epoch{N}_G.pt with
the reference's state_dict keys)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("exp", ["minimaxgan_l1", "wgan_rmse", "experiment1_global_local_D"])
def test_plugin_runs_and_checkpoints(tmp_path, exp):
    import gan_inpainting_amd  # noqa: F401
    from gan_inpainting_amd import train
    # config 1 of BASELINE.json: 64x64, bs=4 (num_downs=6, generalised critic head)
    train.main(["-exp", exp, "-ep", "1", "-b", "4", "--imagedim", "64", "--saveevery", "1", "--evalevery", "1",
                "--samples", "16", "--outdir", str(tmp_path), "--dtype", "fp32"])
    ck = os.path.join(str(tmp_path), "model", exp, "epoch1_G.pt")
    assert os.path.exists(ck)
    sd = torch.load(ck)
    assert "model.model.0.weight" in sd and tuple(sd["model.model.0.weight"].shape) == (64, 1, 4, 4)
    assert all(torch.isfinite(v.float()).all() for v in sd.values())
    assert os.path.exists(os.path.join(str(tmp_path), "model", exp, "training_epoch_history.obj"))
