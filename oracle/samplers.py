"""Oracle (test infrastructure): full sampler loops, CPU, torch global RNG consumed in the same
order as the reference so a seeded run replays it exactly.

Restates TAUnSDDM/lib/sampling/sampling.py: get_initial_samples 14-28, TauL.sample 98-234,
LBJF.sample 254-356, MidPointTauL.sample 390-526, PCTauL.sample 534-646.
`model` is any object with __call__(x,t)->(N,D,S), transition(t), rate(t), rate_mat(x,t).
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import ctmc_ops as ops


def initial_samples(N, D, S, initial_dist, std=None):
    if initial_dist == "uniform":
        return torch.randint(low=0, high=S, size=(N, D))
    if initial_dist == "gaussian":
        p = torch.from_numpy(ops.gaussian_initial_pmf(S, std))
        p = p / p.sum(-1, keepdim=True)
        return torch.multinomial(p.view(1, -1), N * D, True).T.reshape(N, D)
    raise NotImplementedError("Unrecognized initial dist " + initial_dist)


def _rates(model, branch, logit_type, logits, x, t_ones, eps):
    """get_reverse_rates (sampling.py:31-78)."""
    if branch == "ctelbo":
        return ops.reverse_rates_ctelbo(logits, x, model.transition(t_ones), model.rate(t_ones), eps)
    qt0 = None if logit_type == "direct" else model.transition(t_ones)
    ll_all, ll_xt = ops.logprob_with_logits(logit_type, logits, x, qt0)
    ratio = torch.exp(ll_all - ll_xt.unsqueeze(-1))
    return ratio * model.rate_mat(x.long(), t_ones), ratio


def branch_of(loss_name):
    return "ctelbo" if loss_name in ("CTElbo", "NLL", "CTElboLambda") else "crm"


def _poisson(lam):
    return torch.poisson(lam)


def _categorical_rows(probs_rows):
    """One draw per row of (R,S) probs, the ATen way (exp-race)."""
    E = torch.empty_like(probs_rows).exponential_(1)
    return ops.exp_race_argmax(probs_rows, E)


def taul_sample(model, N, D, S, *, max_t, min_t, num_steps, initial_dist, init_std, eps_ratio,
                is_ordinal, loss_name, logit_type="direct", corrector_entry_time=0.0,
                num_corrector_steps=0, x_init=None):
    branch = branch_of(loss_name)
    x = initial_samples(N, D, S, initial_dist, init_std) if x_init is None else x_init.clone()
    ts = ops.taul_time_grid(max_t, min_t, num_steps)
    change_dim = []
    for idx, t in enumerate(ts[:-1]):
        h = ts[idx] - ts[idx + 1]
        t_ones = t * torch.ones((N,))
        logits = model(x, t_ones)
        rr, _ = _rates(model, branch, logit_type, logits, x, t_ones, eps_ratio)
        rr = ops.zero_own_state(rr, x)
        jumps = _poisson(rr * h)
        x_new = ops.tauleap_apply(x, jumps, is_ordinal)
        change_dim.append(float(torch.sum(x.long() != x_new).item()) / N)
        x = x_new
        if t <= corrector_entry_time:
            for _ in range(num_corrector_steps):
                rate = model.rate(t_ones)
                logits = model(x, t_ones)
                rr, _ = _rates(model, branch, logit_type, logits, x, t_ones, eps_ratio)
                corr = ops.zero_own_state(ops.transpose_forward_rates(rate, x) + ops.zero_own_state(rr, x), x)
                jumps = _poisson(corr * h)
                x = ops.tauleap_apply(x, jumps, is_ordinal)
    if loss_name in ("CTElbo", "NLL"):
        p = F.softmax(model(x, min_t * torch.ones((N,))), dim=2)
        x = torch.max(p, dim=2)[1]
    return x.numpy().astype(int), change_dim


def lbjf_sample(model, N, D, S, *, max_t, min_t, num_steps, initial_dist, init_std, eps_ratio,
                loss_name, logit_type="direct", corrector_entry_time=0.0, num_corrector_steps=0,
                x_init=None):
    branch = branch_of(loss_name)
    x = initial_samples(N, D, S, initial_dist, init_std) if x_init is None else x_init.clone()
    ts = ops.taul_time_grid(max_t, min_t, num_steps)
    change_dim = []
    for idx, t in enumerate(ts[:-1]):
        h = ts[idx] - ts[idx + 1]
        t_ones = t * torch.ones((N,))
        rate = model.rate(t_ones)
        logits = model(x, t_ones)
        rr, _ = _rates(model, branch, logit_type, logits, x, t_ones, eps_ratio)
        P = ops.lbjf_posterior(rr, x, h)
        probs = ops.categorical_probs_from_logits(torch.log(P + 1e-35).view(-1, S))
        x_new = _categorical_rows(probs).view(N, D)
        change_dim.append(float(torch.sum(x != x_new).item()) / N)
        if t <= corrector_entry_time:
            for _ in range(num_corrector_steps):
                logits = model(x_new, t_ones)
                rr, _ = _rates(model, branch, logit_type, logits, x_new, t_ones, eps_ratio)
                corr = ops.zero_own_state(ops.transpose_forward_rates(rate, x_new) + rr, x_new)
                P = ops.lbjf_posterior(corr, x_new, h)
                probs = ops.categorical_probs_from_logits(torch.log(P + 1e-35).view(-1, S))
                x_new = _categorical_rows(probs).view(N, D)
        x = x_new
    if loss_name == "CTElbo":
        p = F.softmax(model(x, min_t * torch.ones((N,))), dim=2)
        x = torch.max(p, dim=2)[1]
    return x.numpy().astype(int), change_dim


def midpoint_sample(model, N, D, S, *, max_t, min_t, num_steps, initial_dist, init_std, eps_ratio,
                    is_ordinal, loss_name, logit_type="direct", x_init=None):
    branch = branch_of(loss_name)
    x = initial_samples(N, D, S, initial_dist, init_std) if x_init is None else x_init.clone()
    t = max_t
    h = (max_t - min_t) / num_steps
    change_jump, change_dim, change_dim_first, change_1to2 = [], [], [], []
    while t - 0.5 * h > min_t:
        t_ones = t * torch.ones((N,))
        t_05 = t_ones - 0.5 * h
        rr, _ = _rates(model, branch, logit_type, model(x, t_ones), x, t_ones, eps_ratio)
        rr = ops.zero_own_state(rr, x)
        x_prime = ops.midpoint_predict(x, rr, h, S)
        change_dim_first.append((torch.sum(x.long() != x_prime) / (N * D)).item())
        rr2, _ = _rates(model, branch, logit_type, model(x_prime, t_05), x_prime, t_05, eps_ratio)
        rr2 = ops.zero_own_state(rr2, x_prime)
        flips = _poisson(rr2 * h)
        if is_ordinal:
            js = torch.sum(flips, dim=-1)
            changes = torch.sum((js > 0).to(dtype=float))
            rej = torch.sum((js > 1).to(dtype=float))
            change_jump.append((rej / changes).item())
        # xp = x + sum_s flips*(s - x')   (sampling.py:499-502); clip afterwards
        diff = torch.arange(S, dtype=torch.float32).view(1, 1, S) - x_prime.to(torch.float32).unsqueeze(-1)
        fl = flips
        if not is_ordinal:
            fl = flips * (torch.sum(flips, dim=-1, keepdim=True) <= 1)
        xp = x.to(torch.float32) + torch.sum(fl * diff, dim=-1)
        x_new = torch.clip(xp, min=0, max=S - 1)
        change_dim.append((torch.sum(xp != x.to(torch.float32)) / (N * D)).item())
        change_1to2.append((torch.sum(x_prime.to(torch.float32) != x_new) / (N * D)).item())
        x = x_new.long()
        t = t - h
    if loss_name == "CTElbo":
        p = F.softmax(model(x, min_t * torch.ones((N,))), dim=2)
        x = torch.max(p, dim=2)[1]
    return x.numpy().astype(int), change_jump, change_dim, change_dim_first, change_1to2


def pctaul_sample(model, N, D, S, *, min_t, num_steps, initial_dist, eps_ratio,
                  corrector_entry_time, num_corrector_steps, corrector_step_size_multiplier,
                  x_init=None):
    """PCTauL hard-codes initial_dist_std=200, t grid from 1.0, ordinal update, CTElbo rates."""
    x = initial_samples(N, D, S, initial_dist, 200) if x_init is None else x_init.clone()
    ts = ops.pc_time_grid(min_t, num_steps)

    def rates_at(xx, tt):
        t_ones = tt * torch.ones((N,))
        qt0, rate = model.transition(t_ones), model.rate(t_ones)
        rr, _ = ops.reverse_rates_ctelbo(model(xx, t_ones), xx, qt0, rate, eps_ratio)
        return ops.transpose_forward_rates(rate, xx), ops.zero_own_state(rr, xx)

    for idx, t in enumerate(ts[:-1]):
        h = ts[idx] - ts[idx + 1]
        _, rr = rates_at(x, t)
        x = ops.tauleap_apply(x, _poisson(rr * h), True)
        if t <= corrector_entry_time:
            for _ in range(num_corrector_steps):
                tf, rr = rates_at(x, t - h)
                corr = ops.zero_own_state(tf + rr, x)
                x = ops.tauleap_apply(x, _poisson(corr * (corrector_step_size_multiplier * h)), True)
    p = F.softmax(model(x, min_t * torch.ones((N,))), dim=2)
    return torch.max(p, dim=2)[1].numpy().astype(int)


def exact_sample(model, N, D, S, *, max_t, min_t, num_steps, initial_dist, init_std, x_init=None):
    """ExactSampling.sample (sampling.py:990-1061, log_prob == 'cat'): per step
    x_new^d ~ Categorical(logits = logsumexp_x0(log p0t[x0] + log(q_{t-h|0}[x0,s] q_{t|t-h}[s,x_t])))."""
    x = initial_samples(N, D, S, initial_dist, init_std) if x_init is None else x_init.clone()
    ts = ops.taul_time_grid(max_t, min_t, num_steps)
    change = []
    for idx, t in enumerate(ts[:-1]):
        h = ts[idx] - ts[idx + 1]
        t_ones = t * torch.ones((N,))
        log_p0t = F.log_softmax(model(x, t_ones), dim=2)
        t_eps = t - h
        q_teps_0 = model.transition(t_eps * torch.ones((N,))).unsqueeze(1)                  # N,1,S,S
        q_t_teps = model.transit_between(t_eps * torch.ones((N,)), t_ones).permute(0, 2, 1)  # [n, x_t, s]
        q_t_teps = q_t_teps[torch.arange(N).view(N, 1), x.long()].unsqueeze(-2)             # N,D,1,S
        log_prob = torch.logsumexp(log_p0t.unsqueeze(-1) + torch.log(q_teps_0 * q_t_teps), dim=-2).view(-1, S)
        x_new = _categorical_rows(ops.categorical_probs_from_logits(log_prob)).view(N, D)
        change.append((torch.sum(x_new != x) / (N * D)).item())
        x = x_new
    return x.numpy().astype(int), change


def lbjf_corrector_posterior(model, logits, xt, t_ones, h, logit_type, xt_target=None):
    """lbjf_corrector_step (sampling.py:1064-1085) as its docstring and the LBJF corrector (296-341) state it:
    posterior = h * (exp(ll_all - ll_xt) + 1) * R_t[x_t, :] off the own state, clip(1 - sum, 0) on it, normalised.
    (The reference body multiplies the (N,D,S) ratio by the (N,S,S) `model.rate(t)`: it only runs for D == S and then
    takes the rate row of the dimension index; nothing in the reference calls it.)  Returns (N, D, S) probabilities.
    PARITY UNPINNED: the reference function cannot run as written, so no fixture of it exists; this is the builder's reading.
    xt_target (1066-1067, 1076): the state whose entry is masked and receives the diagonal; ratios and rate row stay x_t's.
    The rate row's own entry R_t[x_t, x_t] (negative: minus the row sum) is zeroed as well -- with x_target != x_t the
    reference leaves it in and takes the log of a negative number."""
    qt0 = None if logit_type == "direct" else model.transition(t_ones)
    ll_all, ll_xt = ops.logprob_with_logits(logit_type, logits, xt, qt0)
    fwd = model.rate_mat(xt.long(), t_ones)
    own = F.one_hot((xt if xt_target is None else xt_target).long(), logits.shape[-1]).to(fwd.dtype)
    if xt_target is not None:
        fwd = fwd * (1 - F.one_hot(xt.long(), logits.shape[-1]).to(fwd.dtype))
    post = h * (torch.exp(ll_all - ll_xt.unsqueeze(-1)) * fwd + fwd) * (1 - own)
    post = post + torch.clip(1.0 - post.sum(-1, keepdim=True), min=0) * own
    return post / post.sum(-1, keepdim=True)
