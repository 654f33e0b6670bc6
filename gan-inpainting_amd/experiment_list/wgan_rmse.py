"""Plugin `wgan_rmse`: begin(state, loaders), the contract of the reference's train.py:183-186.
Schedule: the reference's experiment_list/wgan_l1.py:110-186 (WGAN critic with weight clipping 0.01,
RMSprop lr=5e-5, RMSE reconstruction; G every 140 batches while G_iter_count < 25 or % 500 == 0, else
every 5, :157-163), with one=+1 / mone=-1 (the reference's torch.FloatTensor(1) is uninitialised memory).
The reference's wgan_rmse.py does not parse (SyntaxError at :104); this is its evident intent: wgan_l1 with RMSELoss."""
from . import _common as C


def begin(state, loaders):
    state, exp_dir, logger, device = C.setup(state, "wgan_rmse")
    net_G, (net_D,) = C.build_networks(state, device, n_disc=1, sigmoid=False)    # wgan_l1.py:58
    opt_G, opt_D = C.make_optimizers("rmsprop", net_G, net_D.parameters())
    step = C.trainer.WGANStep(net_G, net_D, opt_G, opt_D, recon="rmse", clip=0.01, sync=C.make_sync(),
                              gp_lambda=float(state.get("gp_lambda", 0.0)), overlap=bool(state.get("overlap", True)))   # > 0: WGAN-GP extension instead of clipping (fp32)
    counters = {"G_iter_count": 0}

    def batch(bi, ground, mask):
        upd = C.wgan_cadence(state, bi, counters["G_iter_count"])
        L = step(ground, mask, upd)
        if upd:
            counters["G_iter_count"] += 1
        return L, upd

    return C.run_epochs(state, loaders, exp_dir, logger, device, net_G, [net_D], batch, ["avg_d"], step=step)
