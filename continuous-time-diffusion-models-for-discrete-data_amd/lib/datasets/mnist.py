"""Image datasets resident on the device (reference lib/datasets/mnist.py:15-87): `DiscreteMNIST` and
`DiscreteCIFAR10` yield `(uint8 image (C,H,W), label)` with the whole set moved to `device` up front.

The reference subclasses torchvision and downloads; this build has no network and no torchvision, so the
raw files are read where torchvision would have put them under `root`:
  MNIST    root/MNIST/raw/{train,t10k}-{images-idx3,labels-idx1}-ubyte[.gz]        (IDX format)
  CIFAR-10 root/cifar-10-batches-bin/{data_batch_1..5,test_batch}.bin               (binary version: 1 + 3072 bytes per record;
           the python-pickle version is deliberately not read)
A missing file raises FileNotFoundError naming the expected path (`cfg.data.download` cannot be honoured)."""
import gzip
import os
import struct

import numpy as np
import torch
from torch.utils.data import Dataset

import lib.datasets.dataset_utils as dataset_utils


def _open(path):
    if os.path.exists(path):
        return open(path, "rb")
    if os.path.exists(path + ".gz"):
        return gzip.open(path + ".gz", "rb")
    raise FileNotFoundError(f"{path}[.gz] not found (no network in this build: place the raw file there)")


def read_idx(path):
    """IDX file -> numpy array (magic: 0x0000 | dtype 0x08 = uint8 | ndim, then big-endian dims)."""
    with _open(path) as f:
        zero, dtype, ndim = struct.unpack(">HBB", f.read(4))
        if zero != 0 or dtype != 0x08:
            raise ValueError(f"{path}: not a uint8 IDX file")
        dims = struct.unpack(">" + "I" * ndim, f.read(4 * ndim))
        data = np.frombuffer(f.read(), dtype=np.uint8)
    return data.reshape(dims)


def _rotate_nearest(img, max_deg):
    """Random rotation in [-max_deg, max_deg] with nearest sampling (torchvision RandomRotation's default)."""
    ang = (torch.rand((), device=img.device) * 2 - 1) * (max_deg * np.pi / 180.0)
    c, s = torch.cos(ang), torch.sin(ang)
    theta = torch.stack([torch.stack([c, -s, torch.zeros_like(c)]), torch.stack([s, c, torch.zeros_like(c)])]).unsqueeze(0)
    x = img.unsqueeze(0).float()
    grid = torch.nn.functional.affine_grid(theta, x.shape, align_corners=False)
    return torch.nn.functional.grid_sample(x, grid, mode="nearest", padding_mode="zeros", align_corners=False)[0].to(img.dtype)


@dataset_utils.register_dataset
class DiscreteMNIST(Dataset):
    def __init__(self, cfg, device, root=None):
        split = "train" if cfg.data.train else "t10k"
        raw = os.path.join(root or ".", "MNIST", "raw")
        self.data = torch.from_numpy(read_idx(os.path.join(raw, f"{split}-images-idx3-ubyte")).copy()).to(device).view(-1, 1, 28, 28)
        self.targets = torch.from_numpy(read_idx(os.path.join(raw, f"{split}-labels-idx1-ubyte")).astype(np.int64))
        self.random_flips = cfg.data.use_augm

    def __len__(self):
        return self.data.shape[0]

    def __getitem__(self, index):
        img, target = self.data[index], self.targets[index]
        if self.random_flips:
            img = _rotate_nearest(img, 10.0)
        return img, target


@dataset_utils.register_dataset
class DiscreteCIFAR10(Dataset):
    def __init__(self, cfg, device, root=None):
        base = os.path.join(root or ".", "cifar-10-batches-bin")
        names = [f"data_batch_{i}.bin" for i in range(1, 6)] if cfg.data.train else ["test_batch.bin"]
        recs = []
        for n in names:
            with _open(os.path.join(base, n)) as f:
                recs.append(np.frombuffer(f.read(), dtype=np.uint8).reshape(-1, 3073))
        rec = np.concatenate(recs, 0)
        self.targets = torch.from_numpy(rec[:, 0].astype(np.int64))
        self.data = torch.from_numpy(rec[:, 1:].copy()).to(device).view(-1, 3, 32, 32)
        self.random_flips = cfg.data.use_augm

    def __len__(self):
        return self.data.shape[0]

    def __getitem__(self, index):
        img, target = self.data[index], self.targets[index]
        if self.random_flips and torch.rand(()) < 0.5:
            img = img.flip(-1)
        return img, target
