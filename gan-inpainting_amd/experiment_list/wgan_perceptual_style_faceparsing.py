"""Plugin `wgan_perceptual_style_faceparsing` (BASELINE config 5): begin(state, loaders), train.py:183-186.
Schedule: experiment_list/wgan_perceptual_style_faceparsing.py:136-232 of the reference - the WGAN loop of
wgan_l1 (critic without sigmoid :57, RMSprop lr=5e-5 :71-72, clipping 0.01 :180-181, G every 140 / 5 batches
:185-188) whose generator loss adds the global + local RMSE, the face-parsing cross entropy through the frozen
segmentation network state['segmentation_model'] (train.py:171-175), VGG-19 perceptual + style terms and total
variation (:206-222). Fixes forced by the file: one=+1 / mone=-1 (uninitialised :78-79), the undefined
`recon_loss` of :222 = recon_global + recon_local, LocalLoss(RMSELoss()) (:62, raises) = masked RMSE.

state['segmentation_model']: a gan_inpainting_amd UnetGenerator(1,4,7,ngf=32) (frozen, eval) or None (term
dropped); state['vgg']: a networks.VGG19Wrapper, None = random-initialised stand-in (the pretrained torchvision
download of networks.py:371 is not available here; load it with VGG19Wrapper.load_state_dict), False = terms dropped."""
from . import _common as C
from ..lib.models import loss, networks


def begin(state, loaders):
    state, exp_dir, logger, device = C.setup(state, "wgan_perceptual_style_faceparsing")
    net_G, (net_D,) = C.build_networks(state, device, n_disc=1, sigmoid=False)          # :56-57
    opt_G, opt_D = C.make_optimizers("rmsprop", net_G, net_D.parameters())              # :71-72
    seg = state.get("segmentation_model")
    if seg is None:
        logger.info("no segmentation_model in state: the face-parsing term is dropped")
    else:
        seg = seg.to(device)
    vgg = state.get("vgg")
    if vgg is None:
        logger.info("no VGG-19 weights in state['vgg']: random-initialised feature network (perceptual/style are logged constants)")
        vgg = networks.VGG19Wrapper(max_pairs=state["batchsize"]).to(device)
    elif vgg is False:
        vgg = None
    step = C.trainer.WGANPerceptualStep(net_G, net_D, opt_G, opt_D, vgg=vgg, segment_model=seg, clip=0.01, sync=C.make_sync(),
                                        overlap=bool(state.get("overlap", True)))
    counters = {"G_iter_count": 0}

    def batch(bi, ground, mask, segment):
        upd = C.wgan_cadence(state, bi, counters["G_iter_count"])     # the reference cadence :185-191 (or --g-every)
        L = step(ground, mask, upd, segment=segment if seg is not None else None)
        if upd:
            counters["G_iter_count"] += 1
        return L, upd

    return C.run_epochs(state, loaders, exp_dir, logger, device, net_G, [net_D], batch, ["avg_d"], pass_extra=True, step=step)
