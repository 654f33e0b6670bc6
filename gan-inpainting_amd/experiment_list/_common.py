"""Shared host logic of the experiment plugins: the epoch loop, logging, bookkeeping and
checkpoint cadence of the reference plugins (experiment_list/minimaxgan_l1.py:83-261 and siblings)
around a per-batch step object from gan_inpainting_amd.trainer.

Differences to the reference, all host-side and listed in DESIGN.md:
  * save directory / log file come from state['outdir'] (the reference hard-codes
    /home/s2125048/thesis/model/<title>/ and log/<title>.log, minimaxgan_l1.py:27,34);
  * losses are read back once per `logevery` batches instead of ~45 blocking .item() calls per batch;
  * the gradient-flow histogram (one .item() per tensor per batch, :180-182,:195-197) is one kernel +
    one copy per batch (util.GradFlow), accumulated on the device;
  * the evaluation pass (minimaxgan_l1.py:234-240) runs lib.models.evaluate.calculate_metric on the
    train and test loaders (reconstruction metrics on fused HIP reductions; FID reported as -1, needs
    Inception weights); `state['eval_fn']`, if given, replaces it and is called with (net_G, loader, epoch).
"""
import contextlib
import logging
import os
import pickle
import time

import torch

from .. import optim, trainer
from ..lib.models import evaluate, networks, util


def _rank():
    import torch.distributed as dist
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def setup(state, title):
    state = state.copy()
    state.update({"title": title})
    outdir = state.get("outdir") or os.path.join(os.getcwd(), "runs")
    exp_dir = os.path.join(outdir, "model", title)
    os.makedirs(exp_dir, exist_ok=True)
    os.makedirs(os.path.join(outdir, "log"), exist_ok=True)
    logger = logging.getLogger(title)
    if not logger.handlers:
        # data-parallel runs: rank 0 owns the log file (and the checkpoints / histories, run_epochs)
        h = logging.FileHandler(os.path.join(outdir, "log", f"{title}.log")) if _rank() == 0 else logging.NullHandler()
        h.setFormatter(logging.Formatter("%(asctime)s %(message)s"))
        logger.addHandler(h)
        logger.propagate = _rank() == 0
    logger.setLevel(logging.INFO)
    if not torch.cuda.is_available():
        raise RuntimeError("the HIP backend needs a gfx950 device (there is no CPU fallback)")
    device = torch.device("cuda", torch.cuda.current_device())
    logger.info("using device %s", device)
    return state, exp_dir, logger, device


def build_networks(state, device, n_disc=1, sigmoid=True):
    size = state["imagedim"]
    dtype = state.get("dtype", "fp16")
    if state.get("generator", "unet") != "unet" or state.get("discriminator", "patchgan") != "patchgan":
        raise NotImplementedError("HIP backend accelerates -g unet -d patchgan (the reference defaults, train.py:27-28)")
    num_downs = state.get("num_downs", 7 if size >= 128 else 6)
    if num_downs == 7:
        G = networks.get_network("generator", "unet", dtype=dtype)
    else:
        G = networks.UnetGenerator(1, 1, num_downs, ngf=64, use_dropout="False", dtype=dtype)
    Ds = [networks.PatchGANDiscriminator(sigmoid=sigmoid, image_size=size, dtype=dtype).to(device) for _ in range(n_disc)]
    return G.to(device), Ds


def to_device_images(t, device, state):
    """Loader item -> (n,1,h,w) float32 on the device. Decoded 8-bit images (lib.data.dataset with transform=None)
    go through the device Resize + ToTensor (train.py:69-72); float tensors are taken as they are."""
    t = t.to(device, non_blocking=True)
    if t.dtype == torch.uint8:
        tf = state.get("_device_transform")
        if tf is None:
            from ..lib.data.dataset import DeviceResizeToTensor
            tf = state["_device_transform"] = DeviceResizeToTensor(state["imagedim"])
        return tf(t if t.dim() == 3 else t.reshape(-1, t.shape[-2], t.shape[-1]))
    return t.float().contiguous()


def run_epochs(state, loaders, exp_dir, logger, device, net_G, nets_D, batch_fn, d_names, pass_extra=False, step=None):
    """batch_fn(batch_index, ground, mask[, extra]) -> (loss dict of device scalars, g_updated: bool); extra = the
    loader's third item (the segmentation labels of dataset.py:35-51) when pass_extra is set."""
    num_epochs, save_every, evaluate_every = state["numepoch"], state["saveevery"], state["evalevery"]
    log_every = state.get("logevery", 50)
    gscale = step.sync.grad_scale() if (step is not None and getattr(step, "sync", None) is not None) else 1.0
    flow_G = util.GradFlow(net_G, gscale)
    flows_D = [util.GradFlow(d, gscale) for d in nets_D]
    history, eval_hist = [], []
    side = step.side_stream() if step is not None else None
    side_keys = set(step.side_keys()) if step is not None else set()
    rank0 = _rank() == 0

    def _on(stream):
        return torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()

    for epoch in range(num_epochs + 1):
        start = time.time()
        sums, g_updates, batches = {}, 0, 0
        acc_g = torch.zeros(len(flow_G.names), device=device)
        with _on(side):
            acc_d = [torch.zeros(len(f.names), device=device) for f in flows_D]
        for bi, (ground, mask, extra) in enumerate(loaders["train"]):
            ground = to_device_images(ground, device, state)
            mask = to_device_images(mask, device, state)
            if pass_extra:
                L, g_updated = batch_fn(bi, ground, mask, extra.to(device, non_blocking=True).contiguous())
            else:
                L, g_updated = batch_fn(bi, ground, mask)
            # values the step produced on its side stream (overlapped critic: its losses, its gradient buffer) are
            # accumulated ON that stream, in stream order behind the kernels that wrote them and ahead of the next
            # batch's zero_grad; everything else on the caller's stream. No cross-stream wait per batch.
            for k, v in L.items():
                with _on(side if k in side_keys else None):
                    if k not in sums:
                        sums[k] = torch.zeros(1, dtype=torch.float32, device=device)
                    sums[k] += v.detach().view(1)
            batches += 1
            with _on(side):
                for f, a in zip(flows_D, acc_d):
                    a += f.measure()
            if g_updated:
                g_updates += 1
                acc_g += flow_G.measure()
            if bi % log_every == 0:
                if step is not None:
                    step.poll_overflow(logger)                 # fp16 guard: back the loss scale off after skipped updates
                    step.sync_for_logging()                    # the values below may come from the side stream
                logger.info("[epoch %d/%d][batch %d/%d] %s", epoch, num_epochs, bi, len(loaders["train"]),
                            " ".join(f"{k}: {float(v):.4f}" for k, v in L.items()))
        if step is not None:
            step.sync_for_logging()
        rec = {k: float(v) / max(batches, 1) for k, v in sums.items()}
        grads = {"avg_g": dict(zip(flow_G.names, (acc_g / g_updates).tolist())) if g_updates
                 else {n: -7777 for n in flow_G.names}}     # -7777 sentinel: minimaxgan_l1.py:209-218
        for name, f, a in zip(d_names, flows_D, acc_d):
            grads[name] = dict(zip(f.names, (a / max(batches, 1)).tolist()))
        history.append({"losses": rec, "g_updates": g_updates, "gradients": grads})
        if epoch % evaluate_every == 0 and epoch > 0:
            if state.get("eval_fn"):
                eval_hist.append(state["eval_fn"](net_G, loaders.get("test"), epoch))
            else:                                                # minimaxgan_l1.py:235-240
                prep = lambda t: to_device_images(t, device, state)   # noqa: E731
                rec_eval = {k: evaluate.calculate_metric(device, loaders[k], net_G, mode=k, epoch=epoch, prepare=prep)
                            for k in ("train", "test") if loaders.get(k) is not None}
                eval_hist.append(rec_eval)
                logger.info("VALIDATION: %s", ", ".join(f"{k} - {v}" for k, v in rec_eval.items()))
            if rank0:        # every rank holds the same replica: one writer
                with open(os.path.join(exp_dir, "training_epoch_history.obj"), "wb") as h:
                    pickle.dump(history, h, protocol=pickle.HIGHEST_PROTOCOL)
                with open(os.path.join(exp_dir, "eval_history.obj"), "wb") as h:
                    pickle.dump(eval_hist, h, protocol=pickle.HIGHEST_PROTOCOL)
        if epoch % save_every == 0 and epoch > 0 and rank0:
            # same contract as the reference: G only, plain state_dict (minimaxgan_l1.py:253-255)
            torch.save({k: v.detach().cpu().contiguous() for k, v in net_G.state_dict().items()},
                       os.path.join(exp_dir, "epoch{}_G.pt".format(epoch)))
        logger.info("epoch: %d, time: %.3fs, %s", epoch, time.time() - start,
                    ", ".join(f"{k}: {v:.6f}" for k, v in rec.items()))
    return history


def make_optimizers(kind, net_G, d_params):
    if kind == "adam":    # minimaxgan_l1.py:64-65
        return (optim.Adam(net_G.parameters(), lr=0.0002, betas=(0.5, 0.999)),
                optim.Adam(d_params, lr=0.0002, betas=(0.5, 0.999)))
    return (optim.RMSprop(net_G.parameters(), lr=0.00005),     # wgan_l1.py:64-65
            optim.RMSprop(d_params, lr=0.00005))


def wgan_cadence(state, batch_index, g_iter_count):
    """Whether this batch updates the generator: the reference rule (wgan_l1.py:157-163: every 140th batch while
    fewer than 25 generator updates were made or their count is a multiple of 500, else every 5th; never batch 0),
    or, with --g-every N (an extension for short runs: the reference rule never fires in epochs shorter than 141
    batches), every Nth batch."""
    if state.get("g_every"):
        return batch_index % int(state["g_every"]) == 0 and batch_index > 0
    return trainer.wgan_update_g(batch_index, g_iter_count, update_g_every=5)


def make_sync():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        from .. import parallel
        return parallel.GradSync()
    return None


__all__ = ["setup", "build_networks", "run_epochs", "make_optimizers", "make_sync", "wgan_cadence", "trainer"]
