"""Oracle (test infrastructure): CTMC forward processes q_{t|0}.

Restates TAUnSDDM/lib/models/forward_model.py:
  BirthDeathForwardBase 9-75, UniformRate 78-129, UniformVariantRate 132-204,
  GaussianTargetRate 207-306.
numpy float64 at construction (as the reference), torch-CPU float32 for the per-t tables.
"""
import math

import numpy as np
import torch


# ----------------------------------------------------------------------------- rate matrices
def gaussian_target_rate_matrix(S, rate_sigma, Q_sigma):
    """forward_model.py:216-236.  Two sequential passes; pass 2 is in place and row-major, so
    a lower-triangle entry sees the already-rescaled mirror entry (order matters)."""
    R = np.zeros((S, S))
    vals = np.exp(-np.arange(0, S) ** 2 / (rate_sigma**2))
    half = S // 2
    for i in range(S):
        if i < half:
            js = range(i + 1, S - i)          # i < j < S-i
            for j in js:
                R[i, j] = vals[j - i - 1]
        elif i > half:
            js = range(S - i, i)              # S-1-i < j < i
            for j in js:
                R[i, j] = vals[i - j - 1]
    two_q2 = 2 * Q_sigma**2
    for i in range(S):
        for j in range(S):
            m = R[j, i]
            if m > 0.0:
                R[i, j] = m * np.exp(
                    -((j + 1) ** 2 - (i + 1) ** 2 + S * (i + 1) - S * (j + 1)) / two_q2
                )
    R = R - np.diag(np.diag(R))
    R = R - np.diag(np.sum(R, axis=1))
    return R


def uniform_rate_matrix(S, rate_const):
    """forward_model.py:84-86."""
    R = rate_const * np.ones((S, S))
    R = R - np.diag(np.diag(R))
    R = R - np.diag(np.sum(R, axis=1))
    return R


def birth_death_rate_matrix(S):
    """forward_model.py:15-17."""
    R = np.diag(np.ones((S - 1,)), 1)
    R += np.diag(np.ones((S - 1,)), -1)
    R -= np.diag(np.sum(R, axis=1))
    return R


# ----------------------------------------------------------------------------- processes
class ForwardProcess:
    """One object for the four reference processes.

    kind: "gaussian" | "uniform" | "univar" | "birthdeath"
    All per-t outputs are float32 torch CPU tensors shaped like the reference's.
    """

    def __init__(self, kind, S, **p):
        self.kind, self.S, self.p = kind, S, dict(p)
        if kind == "gaussian":
            R = gaussian_target_rate_matrix(S, p["rate_sigma"], p["Q_sigma"])
            lam, V = np.linalg.eig(R)                      # forward_model.py:238
            Vinv = np.linalg.inv(V)                        # :239
        elif kind in ("uniform", "univar"):
            R = uniform_rate_matrix(S, p["rate_const"])
            lam, V = np.linalg.eigh(R)                     # :87
            Vinv = V.T
        elif kind == "birthdeath":
            R = birth_death_rate_matrix(S)
            lam, V = np.linalg.eigh(R)                     # :18
            Vinv = V.T
        else:
            raise ValueError(kind)
        f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a)).float()
        self.base_rate = f32(R)
        self.eigvals = f32(lam)
        self.eigvecs = f32(V)
        self.inv_eigvecs = f32(Vinv)

    # -- scalar schedules (t: float32 tensor (B,))
    def integral(self, t):
        k, p = self.kind, self.p
        if k == "gaussian":                                 # :246-247
            return p["time_base"] * (p["time_exp"] ** t) - p["time_base"]
        if k == "uniform":                                  # :110-114 (exp(lambda*t))
            return t
        if k == "univar":                                   # :144-152
            f = p["t_func"]
            if f == "log_sqr":
                return torch.log(t**2 + 1)
            if f == "sqrt_cos":
                return -torch.sqrt(torch.cos(torch.pi / 2 * t))
            if f == "log":
                return p["time_base"] * (p["time_exp"] ** t) - p["time_base"]
            raise ValueError("Unknown t_func %s" % f)
        if k == "birthdeath":                               # :35-41
            smin, smax = p["sigma_min"], p["sigma_max"]
            return 0.5 * smin**2 * (smax / smin) ** (2 * t) - 0.5 * smin**2
        raise ValueError(k)

    def beta(self, t):
        k, p = self.kind, self.p
        if k == "gaussian":                                 # :249-250
            return p["time_base"] * math.log(p["time_exp"]) * (p["time_exp"] ** t)
        if k == "uniform":                                  # :95-102 (rate is t-independent)
            return torch.ones_like(t)
        if k == "univar":                                   # :154-164
            f = p["t_func"]
            if f == "log_sqr":
                return 2 * t / (t**2 + 1)
            if f == "sqrt_cos":
                tt = torch.pi / 2 * t
                return torch.pi / 4.0 * (torch.sin(tt) / torch.sqrt(torch.cos(tt)))
            if f == "log":
                return p["time_base"] * math.log(p["time_exp"]) * p["time_exp"] ** t
            raise ValueError("Unknown t_func %s" % f)
        if k == "birthdeath":                               # :26-33
            smin, smax = p["sigma_min"], p["sigma_max"]
            return smin**2 * (smax / smin) ** (2 * t) * math.log(smax / smin)
        raise ValueError(k)

    # -- tables
    def rate(self, t):
        """(B,) -> (B,S,S): beta(t) * R."""
        return self.base_rate.view(1, self.S, self.S) * self.beta(t).view(-1, 1, 1)

    def rate_mat(self, y, t):
        """Row y of rate(t): (B,...) int -> (B,...,S)  (forward_model.py:104-106,174-178,259-263)."""
        r = self.rate(t)
        b = torch.arange(t.shape[0]).view(-1, *([1] * (y.dim() - 1)))
        return r[b, y.long()]

    def _expm(self, scal, right):
        w = torch.exp(scal.view(-1, 1) * self.eigvals.view(1, self.S))            # (B,S)
        return (self.eigvecs.view(1, self.S, self.S) * w.view(-1, 1, self.S)) @ right.view(
            1, self.S, self.S
        )

    def transition(self, t):
        """q_{t|0}: (B,) -> (B,S,S).  Row-normalised except for `uniform` (A3), then <1e-8 -> 0."""
        if self.kind == "univar":                           # :202-204 = transit_between(0, t)
            return self.transit_between(torch.zeros_like(t), t)
        P = self._expm(self.integral(t), self.inv_eigvecs)
        if self.kind != "uniform":
            P = P / torch.sum(P, dim=-1, keepdim=True)
        P[P < 1e-8] = 0.0
        return P

    def transit_between(self, t1, t2):
        """forward_model.py:128-129,180-200,289-306 (gaussian uses eigvecs.T: reference bug kept)."""
        if self.kind == "uniform":
            return self.transition(t2 - t1)
        if self.kind == "birthdeath":
            raise AttributeError("BirthDeathForwardBase has no transit_between")
        d = self.integral(t2) - self.integral(t1)
        P = self._expm(d, self.eigvecs.T.contiguous())
        P = P / torch.sum(P, dim=-1, keepdim=True)
        P[P < 1e-8] = 0.0
        return P


def from_cfg(cfg):
    """Build the process the reference model class named cfg.model.name would mix in."""
    name, S, m = cfg.model.name, cfg.data.S, cfg.model
    if name.startswith("GaussianTarget") or name.startswith("Gaussian"):
        return ForwardProcess("gaussian", S, rate_sigma=m.rate_sigma, Q_sigma=m.Q_sigma,
                              time_exp=m.time_exp, time_base=m.time_base)
    if name.startswith("UniVar"):
        kw = dict(rate_const=m.rate_const, t_func=m.t_func)
        if m.t_func == "log":
            kw.update(time_base=m.time_base, time_exp=m.time_exp)
        return ForwardProcess("univar", S, **kw)
    if name.startswith("Uniform"):
        return ForwardProcess("uniform", S, rate_const=m.rate_const)
    if name.startswith("BirthDeath"):
        return ForwardProcess("birthdeath", S, sigma_min=m.sigma_min, sigma_max=m.sigma_max)
    raise ValueError(name)
