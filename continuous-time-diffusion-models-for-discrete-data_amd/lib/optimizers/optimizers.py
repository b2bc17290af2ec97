"""Optimizer registry entries (reference lib/optimizers/optimizers.py:4-6: plain Adam(params, lr))."""
import torch

import lib.optimizers.optimizers_utils as optimizers_utils


@optimizers_utils.register_optimizer
def Adam(params, cfg):
    return torch.optim.Adam(params, cfg.optimizer.lr)
