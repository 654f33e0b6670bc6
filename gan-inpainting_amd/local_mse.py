"""Masked-region MSE of saved generator checkpoints over the train and test sets, on the HIP engine: the host
mirror of the reference's local_mse.py (:20-138).

What the reference script does (its paths are hard-coded to the thesis machine and it stops on NameErrors at
:27, :63, :99, :102: `eps`, `dataset`, `get_network`, `curr_m_paths`; the intent below is unambiguous):
  * every '.pt' file under the experiment directory, sorted by the number after 'epoch' (:43-54);
  * train / test CSVs through Resize(128) + ToTensor, batch 64, train shuffled, test not (:56-88);
  * per checkpoint and per loader: mask = ceil(mask); masked = ground*(1-mask); out = G(masked) — the RAW generator
    output, not the composite; loss += LocalLoss(MSELoss)(out, ground, mask) = sum((y*m - yhat*m)^2) / count(m != 0)
    (:20-38, :109-127); the mean over batches goes to metric['train'|'test'][epoch] (:129-130);
  * the generator is never put in eval mode (:97-99): batch statistics and dropout are live; `--eval-mode` switches;
  * after every checkpoint the whole metric dict is pickled to `local_mse_{exp}.obj` in the experiment directory (:132-133).
Here Resize + ToTensor run on the device (lib/data/dataset.DeviceResizeToTensor, Pillow-exact) and the loss is the
library's `gi_loss_local` kernel through `loss.LocalLoss(nn.MSELoss)`."""
import argparse
import os
import pickle
import re
from operator import itemgetter

import torch


def build_parser():
    p = argparse.ArgumentParser(description="Local (masked-region) MSE per checkpoint (reference local_mse.py)")
    p.add_argument("--exp", default="wgan_rmse", help="experiment name (local_mse.py:40)")
    p.add_argument("--exp-root", required=True, help="directory searched for epochN_*.pt files (local_mse.py:41)")
    p.add_argument("--data", required=True, help="dataset directory holding csv/train_all_masks.csv and csv/test_all_masks.csv")
    p.add_argument("--imagedim", type=int, default=128, help="local_mse.py:56")
    p.add_argument("--batchsize", type=int, default=64, help="local_mse.py:69,84")
    p.add_argument("--dtype", default="fp16", choices=["fp16", "fp32"])
    p.add_argument("--eval-mode", action="store_true", help="running-statistics BatchNorm, no dropout (the reference stays in train mode)")
    p.add_argument("--seed", type=int, default=None, help="seed of the train loader's shuffle")
    return p


def checkpoint_paths(exp_root):
    found = []
    for root, _dirs, files in os.walk(exp_root):
        for f in files:
            if ".pt" in f:
                found.append({"path": os.path.join(root, f), "ep": int(re.search(r"epoch\d+", f).group(0).split("epoch")[1])})
    return [(d["ep"], d["path"]) for d in sorted(found, key=itemgetter("ep"))]


def main(argv=None):
    import pandas as pd
    from torch import nn
    from gan_inpainting_amd import backend as B
    from gan_inpainting_amd.experiment_list import _common as C
    from gan_inpainting_amd.lib.data import dataset
    from gan_inpainting_amd.lib.models import loss, networks
    args = build_parser().parse_args(argv)
    B.lib()   # fails loudly when the HIP library or the device is missing
    device = torch.device("cuda:0")
    state = {"imagedim": args.imagedim}
    gen = torch.Generator().manual_seed(args.seed) if args.seed is not None else None

    def loader(csv, shuffle):
        df = pd.read_csv(os.path.join(args.data, "csv", csv))
        return torch.utils.data.DataLoader(dataset.InpaintingDataset(args.data, dataframe=df, transform=None), batch_size=args.batchsize,
                                           num_workers=0, shuffle=shuffle, generator=gen if shuffle else None)
    loaders = {"train": loader("train_all_masks.csv", True), "test": loader("test_all_masks.csv", False)}
    criterion = loss.LocalLoss(nn.MSELoss)
    metric = {"train": {}, "test": {}}
    out_path = os.path.join(args.exp_root, f"local_mse_{args.exp}.obj")
    for ep, path in checkpoint_paths(args.exp_root):
        net_G = networks.get_network("generator", "unet", dtype=args.dtype).to(device)
        net_G.load_state_dict(torch.load(path, map_location="cpu"))
        net_G.eval() if args.eval_mode else net_G.train()
        print("epoch", path)
        with torch.no_grad():
            for split in ("train", "test"):
                total, b = torch.zeros((), device=device), 0
                for ground, mask, _ in loaders[split]:
                    b += 1
                    ground, mask = C.to_device_images(ground, device, state), C.to_device_images(mask, device, state)
                    mask = torch.ceil(mask)
                    masked = ground * (1 - mask)
                    total = total + criterion(net_G(masked), ground, mask)
                metric[split][ep] = float(total / max(b, 1))
        with open(out_path, "wb") as handle:
            pickle.dump(metric, handle, protocol=pickle.HIGHEST_PROTOCOL)
    return metric, out_path


if __name__ == "__main__":
    main()
