"""Drop-in for the numeric part of the reference's lib/models/evaluate.py on fused HIP reductions:

  calculate_metric(device, loader, net, ...)            evaluate.py:85-177 (reconstruction metrics :127-158)
  calculate_segmentation_eval_metric(labels, outputs, unique_labels)     evaluate.py:179-224

The reference file does not import (mixed indentation, undefined `l1_criterion` / `segment`); what is
mirrored is what its plugins rely on (minimaxgan_l1.py:236-237: a dict with recon_rmse_global,
recon_l1_global, recon_rmse_local, recon_l1_local, fid, epoch). Out of this backend's scope and
reported as such: FID (Inception weights; `fid` is -1, the reference's own "not computed" value, :150),
the JPEG dump that feeds it (:145-148), the matplotlib panels (`from_model_object`, `from_saved_obj`).
Like the reference, calculate_metric does NOT switch the network to eval(): BatchNorm and Dropout run
in whatever mode the caller left them (training mode in the plugins)."""
import ctypes as C

import torch

from ... import backend as B


class ReconMeter:
    """Running sums of the four reconstruction metrics on the device (one readback at the end)."""

    def __init__(self, device):
        self.acc = torch.zeros(5, dtype=torch.float32, device=device)
        self._scratch = None

    def update(self, ground, gen, mask_c, want_output=False):
        """ground, gen, mask_c: (n,1,H,W) fp32 on the device; mask_c already ceil-ed / flipped."""
        n = ground.numel()
        lib, ctx = B.lib(), B.get_ctx(ground.device)
        ns = lib.gi_eval_recon_scratch_floats(n)
        if self._scratch is None or self._scratch.numel() * 2 < ns:
            self._scratch = torch.empty((ns + 1) // 2, dtype=torch.float64, device=ground.device)
        out = torch.empty_like(ground) if want_output else None
        B.check(lib.gi_eval_recon(ctx, B.ptr(ground), B.ptr(gen), B.ptr(mask_c), n, 1e-16, B.ptr(out), B.ptr(self.acc), None,
                                  B.ptr(self._scratch)))
        return out

    def result(self):
        a = self.acc.tolist()
        nb = max(a[4], 1.0)
        return {"recon_rmse_global": a[0] / nb, "recon_l1_global": a[1] / nb,
                "recon_rmse_local": a[2] / nb, "recon_l1_local": a[3] / nb}


@torch.no_grad()
def calculate_metric(device, loader, net, fid_stats=(-1, -1), mode="test", inception_model=None, epoch=None,
                     is_flip_mask=False, prepare=None):
    """evaluate.py:85-177. `loader` yields (input, mask, _) like the reference's datasets; `prepare` (optional)
    maps a loader item to a (n,1,h,w) float32 device tensor (the device Resize + ToTensor for decoded bytes)."""
    lib = B.lib()
    meter = ReconMeter(device)
    for inp, mask, *_ in loader:
        if prepare is not None:
            inp, mask = prepare(inp), prepare(mask)
        else:
            inp = inp.to(device, non_blocking=True).float().contiguous()
            mask = mask.to(device, non_blocking=True).float().contiguous()
        m = torch.empty_like(mask)
        masked = torch.empty_like(inp)
        B.check(lib.gi_mask_apply(B.get_ctx(inp.device), B.ptr(inp), B.ptr(mask), B.ptr(m), B.ptr(masked), inp.numel(),
                                  1 | (2 if is_flip_mask else 0)))                  # :128-133
        out = net(masked)                                                            # :134
        meter.update(inp, out.detach().contiguous(), m)                              # :135-143
    metric = meter.result()
    metric.update({"fid": -1, "epoch": epoch})                                       # :150 (FID out of scope)
    return metric


def calculate_segmentation_eval_metric(labels, outputs, unique_labels):
    """evaluate.py:179-224. labels: (n,...) int64, outputs: (n,num_classes,...) fp32 logits, both on the
    device. Returns ({u: {'precision','recall','iou'}}, {'precision','recall','iou'}) of 0-d tensors."""
    if not labels.is_cuda or not outputs.is_cuda:
        raise B.BackendError("segmentation metrics take tensors on the gfx950 device")
    if labels.dtype != torch.int64 or outputs.dtype != torch.float32:
        raise B.BackendError("segmentation metrics take int64 labels and float32 logits")
    n, k = outputs.shape[0], outputs.shape[1]
    labels, outputs = labels.detach().contiguous(), outputs.detach().contiguous()
    hw = outputs.numel() // (n * k)
    if labels.numel() != n * hw:
        raise ValueError("labels %s do not match outputs %s" % (tuple(labels.shape), tuple(outputs.shape)))
    uniq = [int(u) for u in unique_labels]
    lib, ctx = B.lib(), B.get_ctx(outputs.device)
    per = torch.empty(len(uniq) * 3, dtype=torch.float32, device=outputs.device)
    across = torch.empty(3, dtype=torch.float32, device=outputs.device)
    counts = torch.empty(n * 16 * 3, dtype=torch.int32, device=outputs.device)
    hu = (C.c_int * len(uniq))(*uniq)
    B.check(lib.gi_seg_metrics(ctx, B.ptr(labels), B.ptr(outputs), n, k, hw, C.cast(hu, C.c_void_p), len(uniq), B.ptr(per),
                               B.ptr(across), B.ptr(counts)))
    names = ("precision", "recall", "iou")
    metric = {u: {nm: per[i * 3 + j] for j, nm in enumerate(names)} for i, u in enumerate(uniq)}
    return metric, {nm: across[j] for j, nm in enumerate(names)}
