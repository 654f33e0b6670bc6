"""Plugin `experiment1_global_local_D`: the reference file is a stand-alone script (no `begin`,
its own argparse, experiment1_global_local_D.py:15-29); its loop body (:139-200: G step first,
global + local PatchGAN with LSGAN losses, lambda1 = lambda2 = 300 on global / masked RMSE, one Adam
over both discriminators) is exposed here behind the same begin(state, loaders) contract as the
other plugins. The per-batch SSIM of :209 (ssim(ground, net_G(masked)), detached) is recorded under
'ssim' from the fused HIP metric kernel (lib/pytorch_ssim); state['ssim'] = False skips it."""
from . import _common as C
from .. import optim
from ..lib import pytorch_ssim


def begin(state, loaders):
    state, exp_dir, logger, device = C.setup(state, "experiment1_global_local_D")
    net_G, (net_Dg, net_Dl) = C.build_networks(state, device, n_disc=2, sigmoid=True)
    opt_G, opt_D = C.make_optimizers("adam", net_G, optim.chain(net_Dl.parameters(), net_Dg.parameters()))   # :122-123
    step = C.trainer.DualDStep(net_G, net_Dg, net_Dl, opt_G, opt_D, lam1=state.get("lambda1", 300.0),
                               lam2=state.get("lambda2", 300.0), sync=C.make_sync())

    with_ssim = bool(state.get("ssim", True))

    def batch(bi, ground, mask):
        L = dict(step(ground, mask))
        if with_ssim:
            L["ssim"] = pytorch_ssim.ssim(ground, step.gen)                      # :209
        return L, True

    return C.run_epochs(state, loaders, exp_dir, logger, device, net_G, [net_Dg, net_Dl], batch, ["avg_d_global", "avg_d_local"], step=step)
