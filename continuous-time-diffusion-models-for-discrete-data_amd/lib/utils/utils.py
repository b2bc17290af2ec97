"""Small tensor helpers of the reference's lib/utils/utils.py that sit on the hot path
(expand_dims 59-62, log1mexp 86-91) plus the DDP state-dict key helpers (39-56)."""
import torch


def expand_dims(x, axis):
    for a in axis:
        x = x.unsqueeze(a)
    return x


def log1mexp(x):
    """log(1 - exp(-|x|)), stable on both sides of log(2)."""
    x = -torch.abs(x)
    return torch.where(x > -0.693, torch.log(-torch.expm1(x)), torch.log1p(-torch.exp(x)))


def is_model_state_DDP(state):
    return any(".module." in k for k in state.keys())


def remove_module_from_keys(state):
    return {k.replace(".module.", "."): v for k, v in state.items()}
