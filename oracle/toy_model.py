"""Oracle (test infrastructure): a deterministic stand-in score function.

Used where a test needs `model(x, t) -> (N,D,S) logits` without a network: the golden generator
hands it to the *reference* samplers/losses, the tests hand the same function to the oracle and
to the HIP engine.  Pure function of (x, t); float32; no parameters.
"""
import torch


def toy_logits(x, t, S, scale=1.0):
    """Peaked around a t-dependent shrink of x towards S/2, plus a fixed (d,s) ripple."""
    x = x.to(torch.float32)
    N, D = x.shape
    s = torch.arange(S, dtype=torch.float32, device=x.device).view(1, 1, S)
    d = torch.arange(D, dtype=torch.float32, device=x.device).view(1, D, 1)
    tt = t.to(torch.float32).view(N, 1, 1)
    centre = x.unsqueeze(-1) * (1.0 - 0.5 * tt) + 0.5 * tt * (S / 2.0)
    width = 0.05 * S + 0.25 * S * tt + 0.5
    ripple = 0.3 * torch.sin(0.37 * s + 0.11 * d)
    return scale * (-0.5 * ((s - centre) / width) ** 2 + ripple)


class ToyModel:
    """Duck-types the reference model object: __call__, transition, rate, rate_mat, device."""

    def __init__(self, process, S, device="cpu", scale=1.0):
        self.process, self.S, self.device, self.scale = process, S, device, scale
        self.calls = []

    def __call__(self, x, t, *a):
        self.calls.append((x.clone(), t.clone()))
        return toy_logits(x, t, self.S, self.scale)

    def transition(self, t):
        return self.process.transition(t)

    def rate(self, t):
        return self.process.rate(t)

    def rate_mat(self, y, t):
        return self.process.rate_mat(y, t)

    def transit_between(self, t1, t2):
        return self.process.transit_between(t1, t2)
