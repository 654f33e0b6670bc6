"""Input pipeline of the reference (lib/data/dataset.py:6-51, train.py:64-90) with the transform on the device.

Reference: PIL decode -> convert('L') -> torchvision Resize(image_target_size) -> ToTensor(), per sample on the
host inside a num_workers=0 DataLoader (train.py:75-79). Here the dataset hands over the decoded 8-bit grey images
(decode stays on the host: PIL) and `DeviceResizeToTensor` runs Resize + ToTensor for a whole batch on the GPU,
bit-exact with Pillow's resampler (csrc/resize.hip; tests/test_resize_gpu.py)."""
import os

import numpy as np
import torch

from ... import backend as B


def pil_loader(path):
    """dataset.py:6-12: open, convert to greyscale."""
    from PIL import Image
    with open(path, "rb") as f:
        img = Image.open(f)
        return img.convert("L")


class InpaintingDataset(torch.utils.data.Dataset):
    """dataset.py:14-51. Rows supply groundtruth_source, mask_source and (optional) segment (.npy label map).
    Returns (groundtruth, mask, segment): with transform=None the two images are uint8 (H, W) tensors for the
    device transform; a host transform (callable on PIL images) is applied like the reference does."""

    def __init__(self, root, dataframe=None, csv_file=None, transform=None):
        if dataframe is not None:
            self.image_df = dataframe
        elif csv_file:
            import pandas as pd
            self.image_df = pd.read_csv(csv_file)
        else:
            raise Exception("Please supply dataframe or file path")
        self.transform = transform
        self.root = root

    def __len__(self):
        return len(self.image_df)

    def __getitem__(self, idx):
        rows = self.image_df.iloc[idx]
        groundtruth = pil_loader(os.path.join(self.root, rows["groundtruth_source"]))
        mask = pil_loader(os.path.join(self.root, rows["mask_source"]))
        if "segment" in rows and isinstance(rows["segment"], str) and rows["segment"]:
            segment = torch.from_numpy(np.load(os.path.join(self.root, rows["segment"])))
        else:
            segment = torch.zeros((1,), dtype=torch.long)
        if self.transform:
            groundtruth, mask = self.transform(groundtruth), self.transform(mask)
        else:
            groundtruth = torch.from_numpy(np.asarray(groundtruth, dtype=np.uint8).copy())
            mask = torch.from_numpy(np.asarray(mask, dtype=np.uint8).copy())
        return groundtruth, mask, segment


class DeviceResizeToTensor:
    """transforms.Compose([transforms.Resize(size), transforms.ToTensor()]) (train.py:69-72) for a batch of decoded
    grey images: (n, H, W) uint8 on the device -> (n, 1, h, w) float32 in [0, 1]."""

    def __init__(self, size):
        self.size = int(size)
        self._tables = {}

    def output_size(self, h, w):
        import ctypes as C
        oh, ow = C.c_int(), C.c_int()
        B.check(B.lib().gi_resize_output_size(h, w, self.size, C.byref(oh), C.byref(ow)))
        return oh.value, ow.value

    def __call__(self, batch_u8, return_bytes=False):
        if batch_u8.dtype != torch.uint8 or not batch_u8.is_cuda or batch_u8.dim() != 3:
            raise B.BackendError("DeviceResizeToTensor takes a (n,H,W) uint8 tensor on the gfx950 device")
        x = batch_u8.contiguous()
        n, h, w = x.shape
        oh, ow = self.output_size(h, w)
        lib, ctx = B.lib(), B.get_ctx(x.device)
        key = (h, w, x.device.index)
        if key not in self._tables:
            t = torch.empty(lib.gi_resize_table_bytes(h, w, oh, ow), dtype=torch.uint8, device=x.device)
            B.check(lib.gi_resize_build_tables(ctx, h, w, oh, ow, B.ptr(t)))
            self._tables[key] = t
        tmp = torch.empty((n, h, ow), dtype=torch.uint8, device=x.device)
        if return_bytes:
            out = torch.empty((n, 1, oh, ow), dtype=torch.uint8, device=x.device)
            B.check(lib.gi_resize_to_tensor(ctx, B.ptr(self._tables[key]), B.ptr(x), n, h, w, oh, ow, None, B.ptr(out), B.ptr(tmp)))
        else:
            out = torch.empty((n, 1, oh, ow), dtype=torch.float32, device=x.device)
            B.check(lib.gi_resize_to_tensor(ctx, B.ptr(self._tables[key]), B.ptr(x), n, h, w, oh, ow, B.ptr(out), None, B.ptr(tmp)))
        return out
