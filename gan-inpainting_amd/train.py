#!/usr/bin/env python3
"""Launcher with the reference's CLI (train.py:22-36: -exp -ep -b -g -d --imagedim --saveevery
--updatediscevery --evalevery --debug) and plugin loop (:183-186 importlib + begin(state, loaders)),
on the HIP backend. The reference launcher does not run as written (wrong
variable names at :173,:180,:183 and a hard-coded dataset path :64); this one iterates over
--experiments as evidently intended and takes the data as --data synthetic (the benchmark's masked-
image batches) or --data <dir> in the reference's dataset layout (csv/train_all_masks.csv, csv/test_all_masks.csv listing image
files; decoded on the host with PIL, resized + normalised on the device). FID statistics (train.py:157-166) are
outside the backend's scope; the face-segmentation checkpoint (:169-175) is not available (--face-parsing random).

    python gan-inpainting_amd/train.py -exp wgan_l1 -ep 2 -b 32 --imagedim 256 --data synthetic
    python -m torch.distributed.run --nproc-per-node 8 gan-inpainting_amd/train.py -exp wgan_rmse ...
"""
import argparse
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


class SyntheticInpainting(torch.utils.data.Dataset):
    """(groundtruth, mask, segment) triples like lib/data/dataset.py:35-51, synthetic content."""

    def __init__(self, n, size, seed):
        self.n, self.size, self.seed = n, size, seed

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed + i)
        s = self.size
        ground = torch.rand((1, s, s), generator=g)
        h = int(torch.randint(s // 8, s // 2 + 1, (1,), generator=g))
        w = int(torch.randint(s // 8, s // 2 + 1, (1,), generator=g))
        y0 = int(torch.randint(0, s - h + 1, (1,), generator=g))
        x0 = int(torch.randint(0, s - w + 1, (1,), generator=g))
        mask = torch.zeros((1, s, s))
        mask[0, y0:y0 + h, x0:x0 + w] = 1.0
        coarse = torch.randint(0, 4, ((s + 7) // 8, (s + 7) // 8), generator=g)          # blocky 4-class face-parsing labels
        segment = coarse.repeat_interleave(8, 0).repeat_interleave(8, 1)[:s, :s].contiguous()
        return ground, mask, segment


def shard_rows(df, rank, world, batchsize):
    """Rows of `df` for this rank: the first floor(len / (world*batchsize)) * world*batchsize rows, interleaved, so that
    all ranks get the same number of full batches (drop_last loaders)."""
    keep = (len(df) // (world * batchsize)) * world * batchsize
    if keep == 0:
        raise ValueError(f"{len(df)} rows cannot fill one batch of {batchsize} on each of {world} ranks (every rank would run zero batches "
                         f"per epoch): lower --batchsize or the number of ranks")
    return df.iloc[:keep].iloc[rank::world].reset_index(drop=True)


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("-exp", "--experiments", nargs="+", required=True)
    parser.add_argument("-ep", "--numepoch", type=int, default=1500)
    parser.add_argument("-b", "--batchsize", type=int, default=128)
    parser.add_argument("-g", "--generator", choices=["unet", "vgg19"], default="unet")
    parser.add_argument("-d", "--discriminator", choices=["patchgan", "dcgan"], default="patchgan")
    parser.add_argument("--imagedim", type=int, default=128)
    parser.add_argument("--saveevery", type=int, default=50)
    parser.add_argument("--updatediscevery", type=int, default=3)
    parser.add_argument("--evalevery", type=int, default=10)
    parser.add_argument("--debug", default="false")
    # backend additions
    parser.add_argument("--dtype", choices=["fp16", "fp32"], default="fp16")
    parser.add_argument("--gp-lambda", dest="gp_lambda", type=float, default=0.0,
                        help="> 0: WGAN-GP gradient penalty (extension, fp32) instead of the reference's weight clipping")
    parser.add_argument("--face-parsing", dest="face_parsing", choices=["off", "random"], default="off",
                        help="random: a randomly initialised frozen UnetGenerator(1,4,7,ngf=32) stands in for the reference's "
                             "_states/face_segmentation checkpoint (train.py:169-175), which is not available")
    parser.add_argument("--g-every", dest="g_every", type=int, default=0,
                        help="WGAN plugins: update G every N batches from the start (0 = the reference's 140-then-5 cadence, which never fires in epochs shorter than 141 batches)")
    parser.add_argument("--workers", type=int, default=0,
                        help="DataLoader worker processes (0 = decode in the training process like the reference, train.py:148-152; "
                             "the PNG decode, not the GPU, bounds real-data throughput at 0)")
    parser.add_argument("--data", default="synthetic")
    parser.add_argument("--samples", type=int, default=1024)
    parser.add_argument("--outdir", default=os.path.join(os.getcwd(), "runs"))
    args = parser.parse_args(argv)
    state = vars(args)

    import gan_inpainting_amd  # noqa: F401
    from gan_inpainting_amd import parallel
    rank, world = parallel.init_from_env()
    if torch.cuda.is_available():
        torch.cuda.set_device(parallel.local_device())
    lkw = {"num_workers": args.workers}
    if args.workers > 0:
        lkw.update(persistent_workers=True, pin_memory=True, prefetch_factor=4)
    if args.data != "synthetic":
        # the reference's dataset layout (train.py:64-90): <dir>/csv/{train,test}_all_masks.csv with columns
        # groundtruth_source, mask_source[, segment]; decode on the host, Resize + ToTensor on the device
        import pandas as pd
        from gan_inpainting_amd.lib.data import dataset

        def mk_real(csv, shuffle):
            df = pd.read_csv(os.path.join(args.data, "csv", csv))
            if world > 1:
                # every rank must see the same number of batches (each optimizer update holds an all-reduce): cut the
                # index to a multiple of world * batchsize, then one interleaved shard per rank
                df = shard_rows(df, rank, world, args.batchsize)
            return torch.utils.data.DataLoader(dataset.InpaintingDataset(args.data, dataframe=df, transform=None),
                                               batch_size=args.batchsize, shuffle=shuffle, drop_last=True, **lkw)
        loaders = {"train": mk_real("train_all_masks.csv", True), "test": mk_real("test_all_masks.csv", True)}     # :75-90
    else:
        mk = lambda n, seed: torch.utils.data.DataLoader(   # noqa: E731
            SyntheticInpainting(n, args.imagedim, seed + 100000 * rank), batch_size=args.batchsize, shuffle=True, drop_last=True, **lkw)
        loaders = {"train": mk(args.samples, 1), "test": mk(max(args.batchsize, 64), 2), "extra": mk(max(args.batchsize, 64), 3)}
    segmentation_model = None
    if args.face_parsing == "random":
        import functools
        from gan_inpainting_amd.lib.models import networks
        torch.manual_seed(20240)
        segmentation_model = networks.UnetGenerator(1, 4, 7, ngf=32, norm_layer=functools.partial(torch.nn.BatchNorm2d, affine=True,
                                                    track_running_stats=True), use_dropout='False', dtype=args.dtype).eval()   # train.py:171-175
    state.update({"train_fid": None, "test_fid": None, "inception_model": None, "segmentation_model": segmentation_model})
    for name in args.experiments:
        exp = importlib.import_module(f"gan_inpainting_amd.experiment_list.{name}")
        exp.begin(state, loaders)


if __name__ == "__main__":
    main()
