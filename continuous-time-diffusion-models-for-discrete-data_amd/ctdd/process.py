"""Device-resident CTMC forward process: eigendecomposition on the host at construction (numpy
float64, exactly as lib/models/forward_model.py does), per-t tables by the K1 HIP kernel.

One object serves the four reference processes; `lib.models.forward_model` wraps it with the
reference's class / attribute names."""
import math

import numpy as np
import torch

from . import native


def gaussian_target_rate_matrix(S, rate_sigma, Q_sigma):
    """Rate matrix with a discretised-Gaussian stationary law (forward_model.py:216-236).
    Pass 1 lays down the symmetric jump kernel exp(-k^2/rate_sigma^2) inside the 'hourglass'
    |i-j| < dist-to-border; pass 2 rescales, in place and in row-major order, every entry whose
    mirror is positive by the detailed-balance factor of the target."""
    R = np.zeros((S, S))
    kern = np.exp(-np.arange(0, S) ** 2 / (rate_sigma**2))
    mid = S // 2
    for i in range(S):
        if i < mid:
            j = np.arange(i + 1, S - i)
            R[i, j] = kern[j - i - 1]
        elif i > mid:
            j = np.arange(S - i, i)
            R[i, j] = kern[i - j - 1]
    denom = 2 * Q_sigma**2
    for i in range(S):
        ip = i + 1
        for j in range(S):
            mirror = R[j, i]
            if mirror > 0.0:
                jp = j + 1
                R[i, j] = mirror * np.exp(-(jp**2 - ip**2 + S * ip - S * jp) / denom)
    R = R - np.diag(np.diag(R))
    return R - np.diag(np.sum(R, axis=1))


def uniform_rate_matrix(S, rate_const):
    R = rate_const * (np.ones((S, S)) - np.eye(S))
    return R - np.diag(np.sum(R, axis=1))


def birth_death_rate_matrix(S):
    R = np.diag(np.ones((S - 1,)), 1) + np.diag(np.ones((S - 1,)), -1)
    return R - np.diag(np.sum(R, axis=1))


class DeviceForwardProcess:
    KINDS = ("gaussian", "uniform", "univar", "birthdeath")

    def __init__(self, kind, S, device, **p):
        if kind not in self.KINDS:
            raise ValueError(f"unknown forward process {kind}")
        self.kind, self.S, self.p, self.device = kind, int(S), dict(p), torch.device(device)
        if kind == "gaussian":
            R = gaussian_target_rate_matrix(S, p["rate_sigma"], p["Q_sigma"])
            lam, V = np.linalg.eig(R)
            W = np.linalg.inv(V)
        else:
            R = {"uniform": lambda: uniform_rate_matrix(S, p["rate_const"]),
                 "univar": lambda: uniform_rate_matrix(S, p["rate_const"]),
                 "birthdeath": lambda: birth_death_rate_matrix(S)}[kind]()
            lam, V = np.linalg.eigh(R)
            W = V.T
        f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a)).float().to(self.device).contiguous()
        self.base_rate, self.eigvals, self.eigvecs, self.right = f32(R), f32(lam), f32(V), f32(W)
        self.eigvecsT = f32(np.ascontiguousarray(V.T))
        self.normalise = kind != "uniform"       # UniformRate.transition does not row-normalise (A3)

    # ---- scalar schedules; t is a float32 tensor (any device) -> same device
    def integral(self, t):
        k, p = self.kind, self.p
        if k == "gaussian" or (k == "univar" and p["t_func"] == "log"):
            return p["time_base"] * (p["time_exp"] ** t) - p["time_base"]
        if k == "uniform":
            return t
        if k == "univar":
            if p["t_func"] == "log_sqr":
                return torch.log(t**2 + 1)
            if p["t_func"] == "sqrt_cos":
                return -torch.sqrt(torch.cos(torch.pi / 2 * t))
            raise ValueError("Unknown t_func %s" % p["t_func"])
        smin, smax = p["sigma_min"], p["sigma_max"]
        return 0.5 * smin**2 * (smax / smin) ** (2 * t) - 0.5 * smin**2

    def beta(self, t):
        k, p = self.kind, self.p
        if k == "gaussian" or (k == "univar" and p["t_func"] == "log"):
            return p["time_base"] * math.log(p["time_exp"]) * (p["time_exp"] ** t)
        if k == "uniform":
            return torch.ones_like(t)
        if k == "univar":
            if p["t_func"] == "log_sqr":
                return 2 * t / (t**2 + 1)
            if p["t_func"] == "sqrt_cos":
                a = torch.pi / 2 * t
                return torch.pi / 4.0 * (torch.sin(a) / torch.sqrt(torch.cos(a)))
            raise ValueError("Unknown t_func %s" % p["t_func"])
        smin, smax = p["sigma_min"], p["sigma_max"]
        return smin**2 * (smax / smin) ** (2 * t) * math.log(smax / smin)

    def exponent(self, t, t_from=None):
        """Scalar multiplying the eigenvalues in q_{t|t_from}."""
        c = self.integral(t)
        if t_from is not None:
            c = c - self.integral(t_from)
        elif self.kind == "univar":                  # transition(t) = transit_between(0,t) (:202-204)
            c = c - self.integral(torch.zeros_like(t))
        return c

    # ---- tables through the HIP kernel
    def _dev(self, t):
        return t.to(device=self.device, dtype=torch.float32).contiguous()

    def tables(self, t, *, t_from=None, between=False, want_qt0=True, want_qt0T=False, want_rate=False,
               want_noise_probs=False):
        """(qt0, qt0T, rate, noise_probs), each (nT,S,S) or None.  Scalars are evaluated on the
        device `t` lives on (host float32 for the samplers' grids = the reference's CPU values)."""
        c, b = self._dev(self.exponent(t, t_from)), self._dev(self.beta(t))
        right = self.right
        normalise = self.normalise
        if between:
            if self.kind == "birthdeath":
                raise AttributeError("BirthDeathForwardBase has no transit_between")
            if self.kind == "gaussian":
                right = self.eigvecsT              # reference uses eigvecs.T here (:298), kept
            normalise = self.kind != "uniform"
        return native.rate_table(self.eigvecs, right, self.eigvals, self.base_rate, c, b, self.S, normalise, 1e-8,
                                 want_qt0, want_qt0T, want_rate, want_noise_probs)

    def transition(self, t):
        return self.tables(t)[0]

    def rate(self, t):
        return self.tables(t, want_qt0=False, want_rate=True)[2]

    def transit_between(self, t1, t2):
        if self.kind == "uniform":
            return self.tables(t2 - t1)[0]
        return self.tables(t2, t_from=t1, between=True)[0]

    def rate_mat(self, y, t):
        """rate(t)[b, y, :] without materialising (B,S,S): beta(t) * base_rate[y]."""
        b = self._dev(self.beta(t)).view(-1, *([1] * y.dim()))
        return self.base_rate[y.long()] * b
