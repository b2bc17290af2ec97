"""Oracle (test infrastructure): reverse rates, log-probs, noising and sampler step functions.

torch-CPU float32 restatements; every stochastic op takes its noise explicitly (exponential-race
noise E for categorical draws, integer jump counts for Poisson steps) so the HIP kernels can be
compared on identical inputs.  Reference lines are cited per function.
"""
import numpy as np
import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------- A7 initial state
def gaussian_initial_pmf(S, std):
    """sampling.py:18-21 (float64 numpy): pmf over k=1..S, centred at S//2."""
    target = np.exp(-((np.arange(1, S + 1) - S // 2) ** 2) / (2 * std**2))
    return target / np.sum(target)


# ----------------------------------------------------------------------------- A6 log-probs
def logprob_with_logits(logit_type, logits, xt, qt0=None, xt_target=None):
    """model_utils.py:30-60.  logits (B,D,S), xt (B,D), qt0 (B,S,S) -> ll_all (B,D,S), ll_xt (B,D)."""
    if xt_target is None:
        xt_target = xt
    S = logits.shape[-1]
    if logit_type == "direct":
        ll_all = F.log_softmax(logits, dim=-1)
    elif logit_type == "reverse_prob":
        p0t = F.softmax(logits, dim=-1)
        ll_all = torch.log(p0t @ qt0 + 1e-35)
    elif logit_type == "reverse_logscale":
        log_p0t = F.log_softmax(logits, dim=-1)
        log_qt0 = torch.where(qt0 <= 1e-35, -1e9, torch.log(qt0))
        ll_all = torch.logsumexp(log_p0t.unsqueeze(-1) + log_qt0.unsqueeze(1), dim=-2)
    else:
        raise ValueError("Unknown logit_type: %s" % logit_type)
    ll_xt = torch.gather(ll_all, -1, xt_target.long().unsqueeze(-1)).squeeze(-1)
    return ll_all, ll_xt


# ----------------------------------------------------------------------------- A8 reverse rates
def reverse_rates_ctelbo(logits, x, qt0, rate, eps):
    """sampling.py:32-59.  R^[n,d,s] = rate[n,s,x] * sum_s0 softmax(logits)[s0]/(qt0[n,s0,x]+eps) * qt0[n,s0,s].

    logits (N,D,S) f32, x (N,D) int, qt0/rate (N,S,S).  Returns (reverse_rates, ratio); the
    own-state entry is NOT zeroed here."""
    N, D, S = logits.shape
    xi = x.long()
    p0t = F.softmax(logits, dim=2)
    n = torch.arange(N).view(N, 1)
    den = qt0.transpose(1, 2)[n, xi] + eps            # (N,D,S): qt0[n, s0, x_nd]
    fwd = rate.transpose(1, 2)[n, xi]                 # (N,D,S): rate[n, s, x_nd]
    ratio = (p0t / den) @ qt0
    return fwd * ratio, ratio


def reverse_rates_crm(logit_type, logits, x, qt0, rate):
    """sampling.py:61-73: ratio = exp(ll_all - ll_xt); R^ = ratio * rate[n, x, :]."""
    N = logits.shape[0]
    ll_all, ll_xt = logprob_with_logits(logit_type, logits, x, qt0)
    ratio = torch.exp(ll_all - ll_xt.unsqueeze(-1))
    fwd = rate[torch.arange(N).view(N, 1), x.long()]  # (N,D,S): row x of R_t
    return ratio * fwd, ratio


def transpose_forward_rates(rate, x):
    """rate[n, x_nd, s]  (sampling.py:182-188): the x -> s forward rates used by correctors."""
    N = rate.shape[0]
    return rate[torch.arange(N).view(N, 1), x.long()]


def zero_own_state(r, x):
    """r * (1 - onehot(x))  (sampling.py:127-128) / indexed assignment to 0 (176-180)."""
    out = r.clone()
    out.scatter_(-1, x.long().unsqueeze(-1), 0.0)
    return out


# ----------------------------------------------------------------------------- A9/A11/A12 updates
def tauleap_apply(x, jump_nums, is_ordinal, base=None):
    """sampling.py:135-160 (TauL) / 478-503 (MidPoint: diffs taken w.r.t. `base`=x', added to x).

    x (N,D) int, jump_nums (N,D,S) non-negative counts.  Returns int64 x_new."""
    N, D, S = jump_nums.shape
    jn = jump_nums.to(torch.float32)
    if not is_ordinal:
        keep = (jn.sum(dim=2) <= 1).to(jn.dtype)
        jn = jn * keep.view(N, D, 1)
    ref = (x if base is None else base).to(torch.float32)
    diff = torch.arange(S, dtype=torch.float32).view(1, 1, S) - ref.unsqueeze(-1)
    xp = x.to(torch.float32) + torch.sum(jn * diff, dim=2)
    return torch.clamp(xp, min=0, max=S - 1).long()


def midpoint_predict(x, rates_masked, h, S):
    """sampling.py:437-453: x' = clip(x + round(0.5*h*sum_s R^[s]*(s-x)), 0, S-1).
    rates_masked has the own state already zeroed.  torch.round = half-to-even."""
    diff = torch.arange(S, dtype=torch.float32).view(1, 1, S) - x.to(torch.float32).unsqueeze(-1)
    change = torch.round(0.5 * h * torch.sum(rates_masked * diff, dim=-1)).to(torch.int)
    return torch.clip(x.long() + change, min=0, max=S - 1)


def lbjf_posterior(rates, x, h):
    """sampling.py:278-290: one-step Euler transition row, normalised.  Returns probs (N,D,S)."""
    S = rates.shape[-1]
    onehot = F.one_hot(x.long(), S).to(rates.dtype)
    post0 = rates * (1 - onehot)
    off = torch.sum(post0, dim=-1, keepdim=True)
    diag = torch.clip(1.0 - h * off, min=0, max=float("inf"))
    P = post0 * h + diag * onehot
    return P / torch.sum(P, dim=-1, keepdim=True)


# ----------------------------------------------------------------------------- categorical draws
def categorical_probs_from_logits(logits):
    """What torch.distributions.Categorical(logits=...) hands to multinomial:
    softmax(logits - logsumexp(logits))."""
    lg = logits - logits.logsumexp(dim=-1, keepdim=True)
    return F.softmax(lg, dim=-1)


def exp_race_argmax(probs, E):
    """ATen's one-draw multinomial = argmax_s(probs_s / E_s), E ~ Exp(1).
    First maximal index on ties (torch.argmax)."""
    return torch.argmax(probs / E, dim=-1)


def noise_probs_rows(qt0, x0):
    """losses.py:46-55: rows qt0[b, x0[b,d], :] -> Categorical(logits=where(rows<=0,-1e9,log rows)).
    Returns the (B*D,S) probs the reference's sampler sees."""
    B, D = x0.shape
    rows = qt0[torch.arange(B).view(B, 1), x0.long()].reshape(B * D, -1)
    lg = torch.where(rows <= 0.0, -1e9, torch.log(rows))
    return categorical_probs_from_logits(lg)


def noise_xt(qt0, x0, E):
    """x_t ~ q_{t|0}(.|x0) with explicit exponential noise E (B*D,S).  (losses.py:46-59)"""
    B, D = x0.shape
    return exp_race_argmax(noise_probs_rows(qt0, x0), E).view(B, D)


def xtilde_sample(rate, x_t, E_dim, E_val):
    """losses.py:61-101: one-jump neighbour.  E_dim (B,D), E_val (B,S) exponential noise.
    Returns (square_dims (B,), square_newval (B,), x_tilde (B,D))."""
    B, D = x_t.shape
    rv = zero_own_state(rate[torch.arange(B).view(B, 1), x_t.long()], x_t)      # (B,D,S)
    dimsum = rv.sum(dim=2)                                                       # (B,D)
    pdim = dimsum / dimsum.sum(-1, keepdim=True)       # Categorical(probs=...) normalises
    dims = exp_race_argmax(pdim, E_dim)
    newp = rv[torch.arange(B), dims]                                             # (B,S)
    lg = torch.where(newp <= 0.0, -1e9, torch.log(newp))
    newval = exp_race_argmax(categorical_probs_from_logits(lg), E_val)
    x_tilde = x_t.clone()
    x_tilde[torch.arange(B), dims] = newval
    return dims, newval, x_tilde


# ----------------------------------------------------------------------------- time grids (App. E)
def taul_time_grid(max_t, min_t, num_steps):
    """sampling.py:107-109: ts = concat(linspace(max_t, min_t, num_steps), [0]) in float64."""
    return np.concatenate((np.linspace(max_t, min_t, num_steps), np.array([0])))


def pc_time_grid(min_t, num_steps):
    """sampling.py:553-554."""
    h = 1.0 / num_steps
    return np.linspace(1.0, min_t + h, num_steps)
