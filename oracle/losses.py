"""Oracle (test infrastructure): the training objectives on the CPU with explicit noise.

Restates TAUnSDDM/lib/losses/losses.py: CTElbo 22-286 (and NLL 1514-1778, CTElboLambda 1794-2058,
which share its body), CatRM/CatRMNLL `_comp_loss` 794-836 / 1146-1188 and calc_loss 838-890 /
1190-1242, NLLOriginal 1059-1103, ScoreElbo 1255-1500.  Inputs are the minibatch x0, the drawn times
ts and the noised states (x_t, x~) -- produced by oracle.ctmc_ops.noise_xt / xtilde_sample from
explicit exponential noise -- so the objective is a pure function; `model` is any callable
(x, t) -> (B,D,S) logits with transition(t) / rate(t).
"""
import torch
import torch.nn.functional as F

from . import ctmc_ops as ops


def _log1mexp(x):
    x = -torch.abs(x)
    return torch.where(x > -0.693, torch.log(-torch.expm1(x)), torch.log1p(-torch.exp(x)))


def _rows(tab, x):
    return tab[torch.arange(tab.shape[0]).view(-1, 1), x.long()]


def neg_ct_elbo(logits_reg, logits_sig, x0, reg_x, x_tilde, qt0, rate, eps):
    """losses.py:116-278.  Returns the scalar mean(-sig/norm) + mean(reg)."""
    B, D = x0.shape
    S = qt0.shape[-1]
    p_reg = F.softmax(logits_reg, dim=2)
    # regulariser: sum_{d,s0} p(s0)/(qt0[s0,x]+eps) * sum_s mask(s) rate[s,x] qt0[s0,s]
    den_reg = _rows(qt0.transpose(1, 2), reg_x) + eps
    rv_reg = ops.zero_own_state(_rows(rate.transpose(1, 2), reg_x), reg_x)
    reg_term = torch.sum((p_reg / den_reg) * (rv_reg @ qt0.transpose(1, 2)), dim=(1, 2))
    # signal term
    p_sig = F.softmax(logits_sig, dim=2)
    den_sig = _rows(qt0.transpose(1, 2), x_tilde) + eps
    inner = torch.log((p_sig / den_sig) @ qt0 + eps)
    outer_rate = ops.zero_own_state(_rows(rate.transpose(1, 2), x_tilde), x_tilde)
    q_from_x0 = _rows(qt0, x0)
    q_x0_to_xt = torch.gather(q_from_x0, -1, x_tilde.long().unsqueeze(-1)) + eps
    outer = torch.sum(outer_rate * (q_from_x0 / q_x0_to_xt) * inner, dim=(1, 2))
    row_sums = -torch.diagonal(rate, dim1=1, dim2=2)
    base_tmp = _rows(row_sums.unsqueeze(-1), x_tilde).squeeze(-1) if False else row_sums[torch.arange(B).view(B, 1), x_tilde.long()]
    Z = base_tmp.sum(1).view(B, 1, 1) - base_tmp.view(B, D, 1) + row_sums.view(B, 1, S)
    sig_norm = torch.sum(outer_rate * q_from_x0 / (Z * q_x0_to_xt), dim=(1, 2))
    return torch.mean(-outer / sig_norm) + torch.mean(reg_term)


def ct_elbo_family(kind, model, x0, ts, x_t, x_tilde, *, eps, nll_weight, one_forward_pass, weight=None):
    """kind in {"CTElbo", "NLL", "CTElboLambda"}; weight = n_iter / n_iters for CTElboLambda."""
    qt0, rate = model.transition(ts), model.rate(ts)
    x_logits = model(x_t, ts)
    if one_forward_pass:
        logits_sig, reg_x = x_logits, x_tilde
    else:
        logits_sig, reg_x = model(x_tilde, ts), x_t
    neg_elbo = neg_ct_elbo(x_logits, logits_sig, x0, reg_x, x_tilde, qt0, rate, eps)
    nll = F.cross_entropy(x_logits.permute(0, 2, 1), x0.long())
    if kind == "CTElbo":
        return neg_elbo + nll_weight * nll
    if kind == "NLL":
        return nll
    if kind == "CTElboLambda":
        return weight * neg_elbo + (1 - weight) * nll
    raise ValueError(kind)


def crm_comp_loss(loss_type, S, ll_all, ll_xt, xt, qt0):
    if loss_type == "rm":
        return -ll_xt
    if loss_type == "mle":
        return -((S - 1) * ll_xt + torch.sum(_log1mexp(ll_all), dim=-1) - _log1mexp(ll_xt))
    if loss_type == "elbo":
        own = F.one_hot(xt.long(), S)
        d = ll_all - ll_xt.unsqueeze(-1)
        first = torch.sum(torch.exp(d) * _rows(qt0.transpose(1, 2), xt) * (1 - own), dim=-1)
        second = torch.sum(-d * _rows(qt0, xt) * (1 - own), dim=-1)
        return first - second
    raise ValueError("Unknown loss_type: %s" % loss_type)


def crm_family(kind, model, x0, ts, x_t, *, S, logit_type, loss_type, ce_coeff=0.0, nll_weight=0.0):
    """kind in {"CatRM", "CatRMNLL", "NLLOriginal"}."""
    logits = model(x_t, ts)
    if kind == "NLLOriginal":
        return F.cross_entropy(logits.permute(0, 2, 1), x0.long())
    qt0 = model.transition(ts)
    ll_all, ll_xt = ops.logprob_with_logits(logit_type, logits, x_t, qt0)
    loss = crm_comp_loss(loss_type, S, ll_all, ll_xt, x_t, qt0) * (1 - ce_coeff)
    out = torch.sum(loss) / x0.shape[0]
    if kind == "CatRMNLL":
        out = out + nll_weight * F.cross_entropy(logits.permute(0, 2, 1), x0.long())
    return out


def score_elbo(model, x0, ts, x_t, x_tilde, *, logit_type, eps, nll_weight, one_forward_pass):
    """losses.py:1255-1500."""
    B, D = x0.shape
    qt0, rate = model.transition(ts), model.rate(ts)
    S = qt0.shape[-1]
    reg_x = x_tilde if one_forward_pass else x_t
    logits = model(reg_x, ts)
    ll_all, ll_xt = ops.logprob_with_logits(logit_type, logits, x_tilde, qt0)
    d = ll_all - ll_xt.unsqueeze(-1)
    reg_term = torch.sum(torch.exp(d) * ops.zero_own_state(_rows(rate.transpose(1, 2), reg_x), reg_x), dim=(1, 2))
    outer_rate = ops.zero_own_state(_rows(rate.transpose(1, 2), x_tilde), x_tilde)
    q_from_x0 = _rows(qt0, x0)
    q_x0_to_xt = torch.gather(q_from_x0, -1, x_tilde.long().unsqueeze(-1)) + eps
    outer = torch.sum(outer_rate * (q_from_x0 / q_x0_to_xt) * d, dim=(1, 2))
    row_sums = -torch.diagonal(rate, dim1=1, dim2=2)
    base_tmp = row_sums[torch.arange(B).view(B, 1), x_tilde.long()]
    Z = base_tmp.sum(1).view(B, 1, 1) - base_tmp.view(B, D, 1) + row_sums.view(B, 1, S)
    sig_norm = torch.sum(outer_rate * q_from_x0 / (Z * q_x0_to_xt), dim=(1, 2))
    return torch.mean(-outer / sig_norm) + torch.mean(reg_term) + nll_weight * (torch.sum(-ll_xt) / B)
