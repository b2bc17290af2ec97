"""Synthetic 2-D toy data as bit strings (reference lib/datasets/synthetic.py:164-258): `SyntheticData`
(an .npy of shape (N, discrete_dim) in {0,1}, resident on the device) and the sign + magnitude / Gray-code
codec that maps a 2-D point to `discrete_dim` bits (discrete_dim/2 per coordinate: 1 sign bit + b magnitude
bits) and back.  The codec is vectorised here; the reference builds string dictionaries."""
import numpy as np
import torch
from torch.utils.data import Dataset

import lib.datasets.dataset_utils as dataset_utils


def _gray(n):
    return n ^ (n >> 1)


def _ungray(g):
    n = g.copy()
    shift = 1
    while (g >> shift).any():
        n ^= g >> shift
        shift += 1
    return n


def float2bin(samples, discrete_dim, int_scale, binmode="gray"):
    """(N, 2) floats -> (N, discrete_dim) bits: per coordinate [sign | magnitude MSB..LSB] of int(|x * int_scale|),
    the magnitude Gray-coded when binmode == 'gray'."""
    b = discrete_dim // 2 - 1
    v = np.asarray(samples, dtype=np.float64) * int_scale
    mag = np.abs(v).astype(np.int64)                    # int(abs(x)) truncation
    if (mag >= (1 << b)).any():
        raise ValueError(f"value does not fit {b} magnitude bits")
    sign = (v < 0).astype(np.int64)
    code = _gray(mag) if binmode == "gray" else mag
    bits = (code[..., None] >> np.arange(b - 1, -1, -1)) & 1            # (N, 2, b)
    return np.concatenate([sign[..., None], bits], -1).reshape(len(v), discrete_dim).astype(int)


def bin2float(bits, discrete_dim, int_scale, binmode="gray"):
    b = discrete_dim // 2 - 1
    x = np.asarray(bits, dtype=np.int64).reshape(-1, 2, b + 1)
    code = (x[..., 1:] << np.arange(b - 1, -1, -1)).sum(-1)
    mag = _ungray(code) if binmode == "gray" else code
    return np.where(x[..., 0] == 1, -mag, mag) / float(int_scale)


@dataset_utils.register_dataset
class SyntheticData(Dataset):
    def __init__(self, cfg, device, root):
        self.data = torch.from_numpy(np.load(root, allow_pickle=False)).to(device)

    def __len__(self):
        return self.data.shape[0]

    def __getitem__(self, index):
        return self.data[index]
