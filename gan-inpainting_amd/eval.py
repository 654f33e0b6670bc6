"""Offline inference over saved generator checkpoints on the HIP engine: the host mirror of the reference's
eval.py (:24-127), i.e. the `epoch{N}_G.pt` round trip of SURVEY 8f rank 3.

What the reference script does, and this one does the same way:
  * -m model directory, -d dataset directory, -f CSV of (groundtruth_source, mask_source) rows, -g generator (:24-29);
  * the CSV paths are joined onto the dataset directory (:39-45), the loader is batch 50, not shuffled (:50-55),
    images go through ToTensor only (no Resize, :47-48): [0,1] floats at their stored size;
  * every file with '.pt' in its name: epoch = int(re 'epoch(\\d+?)_'), `get_network('generator', name)`,
    `load_state_dict` (:73-80);
  * masked = ground * mask when the CSV name contains 'real' or 'extra', else ground * (1 - mask) (:89-92); the mask is
    NOT passed through ceil here (unlike the training plugins);
  * out[epoch] = net_G(masked) and every batch overwrites it: the LAST batch's output is what is kept (:94);
  * the network is never put in eval mode (:77-94): BatchNorm normalises with batch statistics and dropout is
    active. `--eval-mode` switches to running statistics / no dropout, which is what a user most likely wants;
  * `{ts}_input.obj` = pickle of {'ground','mask','masked'} numpy arrays of the last batch, `{ts}_output.obj` = pickle
    of {epoch: output} (:118-127).
The reference asks for torch.device("gpu:0") (:33-36), which torch rejects; the evident intent, cuda:0, is what runs
here. `-g vgg19` is not built (out of scope)."""
import argparse
import os
import pickle
import re
from datetime import datetime as dt

import torch


def build_parser():
    p = argparse.ArgumentParser(description="Inference over generator checkpoints (reference eval.py)")
    p.add_argument("-m", "--modelpath", type=str, required=True, help="Model directory")
    p.add_argument("-d", "--datasetpath", type=str, required=True, help="Dataset directory")
    p.add_argument("-f", "--csvfile", type=str, required=True, help="CSV file for dataset dataframe")
    p.add_argument("-g", "--generator", type=str, choices=["unet", "vgg19"], default="unet", help="Generator network name")
    p.add_argument("--batchsize", type=int, default=50, help="eval.py:53")
    p.add_argument("--dtype", default="fp16", choices=["fp16", "fp32"])
    p.add_argument("--eval-mode", action="store_true", help="running-statistics BatchNorm, no dropout (the reference stays in train mode)")
    return p


def _to_unit_float(u8, device):
    """transforms.ToTensor() for a batch of decoded grey images: (n,H,W) uint8 -> (n,1,H,W) float32 / 255. The
    division runs on the host like ToTensor's (correctly rounded: the device's fast division differs in the last
    bit), then the batch goes to the device."""
    return u8.to(torch.float32).div(255).unsqueeze(1).contiguous().to(device, non_blocking=True)


def main(argv=None):
    import pandas as pd
    from gan_inpainting_amd import backend as B
    from gan_inpainting_amd.lib.data import dataset
    from gan_inpainting_amd.lib.models import networks
    args = build_parser().parse_args(argv)
    if args.generator != "unet":
        raise NotImplementedError("HIP backend accelerates -g unet (the reference default, eval.py:29)")
    B.lib()   # fails loudly when the HIP library or the device is missing
    device = torch.device("cuda:0")
    ts = dt.strftime(dt.now(), "%Y%m%d_%H%M%S")

    eval_df = pd.read_csv(args.csvfile)
    for col in ("mask_source", "groundtruth_source"):            # eval.py:39-45
        eval_df[col] = eval_df[col].map(lambda x: os.path.join(args.datasetpath, x))
    loader = torch.utils.data.DataLoader(dataset.InpaintingDataset("", dataframe=eval_df, transform=None),
                                         batch_size=args.batchsize, num_workers=0, shuffle=False)
    keep_mask_side = ("real" in args.csvfile) or ("extra" in args.csvfile)   # eval.py:89

    out, ground, mask, masked = {}, None, None, None
    for m in sorted(os.listdir(args.modelpath)):
        if ".pt" not in m:
            continue
        epoch = int(re.search(r"epoch(\d+?)\_", m).group(1))
        print("eval", m)
        net_G = networks.get_network("generator", args.generator, dtype=args.dtype).to(device)
        net_G.load_state_dict(torch.load(os.path.join(args.modelpath, m), map_location="cpu"))
        net_G.eval() if args.eval_mode else net_G.train()
        with torch.no_grad():
            for gt, mk, _ in loader:
                ground, mask = _to_unit_float(gt, device), _to_unit_float(mk, device)
                masked = ground * mask if keep_mask_side else ground * (1 - mask)
                out[epoch] = net_G(masked).detach().cpu()

    paths = (os.path.join(args.modelpath, f"{ts}_input.obj"), os.path.join(args.modelpath, f"{ts}_output.obj"))
    if ground is not None:
        with open(paths[0], "wb") as handle:
            pickle.dump({"ground": ground.cpu().numpy(), "mask": mask.cpu().numpy(), "masked": masked.cpu().numpy()}, handle,
                        protocol=pickle.HIGHEST_PROTOCOL)
        with open(paths[1], "wb") as handle:
            pickle.dump(out, handle, protocol=pickle.HIGHEST_PROTOCOL)
    return out, paths


if __name__ == "__main__":
    main()
