"""Plugin `minimaxgan_l1`: begin(state, loaders), the contract of the reference's train.py:183-186.
Schedule: experiment_list/minimaxgan_l1.py:110-173 of the reference (minimax GAN, BCE adversarial loss +
L1 reconstruction, Adam lr=2e-4 betas=(0.5,0.999)), executed by trainer.MinimaxStep on the HIP backend."""
from . import _common as C


def begin(state, loaders):
    state, exp_dir, logger, device = C.setup(state, "minimaxgan_l1")
    net_G, (net_D,) = C.build_networks(state, device, n_disc=1, sigmoid=True)
    opt_G, opt_D = C.make_optimizers("adam", net_G, net_D.parameters())
    step = C.trainer.MinimaxStep(net_G, net_D, opt_G, opt_D, recon="l1", sync=C.make_sync())

    def batch(bi, ground, mask):
        return step(ground, mask), True

    return C.run_epochs(state, loaders, exp_dir, logger, device, net_G, [net_D], batch, ["avg_d"], step=step)
