"""Tiny helpers shared by the score networks (reference lib/networks/network_utils.py:7-25)."""
import math

import torch
import torch.nn.functional as F


def transformer_timestep_embedding(timesteps, embedding_dim, max_positions=10000):
    """Fixed sinusoid [sin | cos] of `timesteps` (B,), zero-padded when embedding_dim is odd."""
    assert timesteps.dim() == 1
    half = embedding_dim // 2
    freq = torch.exp(torch.arange(half, dtype=torch.float32, device=timesteps.device)
                     * -(math.log(max_positions) / (half - 1)))
    arg = timesteps.float()[:, None] * freq[None, :]
    emb = torch.cat([torch.sin(arg), torch.cos(arg)], dim=1)
    if embedding_dim % 2 == 1:
        emb = F.pad(emb, (0, 1), mode="constant")
    return emb


def center_data(x, x_min_max):
    """[min,max] -> [-1,1]."""
    lo, hi = x_min_max
    return 2 * ((x - lo) / (hi - lo)) - 1
