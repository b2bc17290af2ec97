"""Training losses behind the reference's registry names (lib/losses/losses.py): CTElbo 12-287,
NLL 1504-1778, CTElboLambda 1783-2058, CatRM 786-890, CatRMNLL 1135-1242, NLLOriginal 1049-1103,
ScoreElbo 1246-1500.

Every loss = forward noising x0 -> x_t (and, for the ELBO family, the one-jump neighbour x~) followed
by an objective on the network's logits.  Noising runs in the HIP kernels K1/K2/K3 (per-sample
q_{t|0} / R_t tables, exponential-race categorical draws, csrc/noising.hip) with a Philox key taken
from torch's generator; the objectives are written as batched gathers and (B,D,S)x(B,S,S) products on
the device so autograd reaches the network (no index-vector construction, SURVEY 6: 63 % of the
reference's CPU time there).  Both argument orders of the reference are accepted: the class-level
`calc_loss(state, minibatch, label=None)` and the stale `calc_loss(minibatch, state)`.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

import lib.losses.losses_utils as losses_utils
import lib.utils.utils as utils
from ctdd import native
from lib.models.model_utils import get_logprob_with_logits


def _seed():
    return int(torch.randint(0, 2**62, (1,), dtype=torch.int64).item())


def _unpack(a, b):
    """(state, minibatch) in either order."""
    return (a, b) if isinstance(a, dict) else (b, a)


def _flatten(minibatch):
    if minibatch.dim() == 4:
        B, C, H, W = minibatch.shape
        minibatch = minibatch.view(B, C * H * W)
    return minibatch


# Parity-test hook: {"ts": (B,), "x_t": (B,D), "x_tilde": (B,D)} replaces the random draws so that the
# objective can be compared with the oracle on identical noise (tests/test_gpu_losses.py).
_FIXED_NOISE = None


def _draw_ts(B, device, lo, hi):
    if _FIXED_NOISE is not None:
        return _FIXED_NOISE["ts"].to(device)
    return torch.rand((B,), device=device) * (hi - lo) + lo


def _noise(model, x0, ts, with_tilde, want_T=False):
    """x_t ~ q_{t|0}(.|x0) per dimension and optionally the one-jump neighbour x~ (losses.py:39-101).
    Returns qt0, rate (B,S,S) and int64 x_t[, x~]; with want_T the transposed table q_{t|0}^T is appended."""
    if _FIXED_NOISE is not None:
        qt0, qT, rate, _ = model.process.tables(ts, want_qt0=True, want_qt0T=want_T, want_rate=True)
        xt = _FIXED_NOISE["x_t"].to(x0.device).long()
        out = (qt0, rate, xt, (_FIXED_NOISE["x_tilde"].to(x0.device).long() if with_tilde else None))
        return out + (qT,) if want_T else out
    qt0, qT, rate, probs = model.process.tables(ts, want_qt0=True, want_qt0T=want_T, want_rate=True, want_noise_probs=True)
    x0i = x0.to(torch.int32).contiguous()
    x_t = native.noise_categorical(probs, x0i, seed=_seed())
    if not with_tilde:
        out = (qt0, rate, x_t.long(), None)
        return out + (qT,) if want_T else out
    _, _, x_tilde = native.xtilde_sample(rate, x_t, seed=_seed())
    out = (qt0, rate, x_t.long(), x_tilde.long())
    return out + (qT,) if want_T else out


class _CtElboFn(torch.autograd.Function):
    """K11 (csrc/losses.hip): CT-ELBO value and logit-gradient in HIP (one forward pass)."""

    @staticmethod
    def forward(ctx, logits, x0, x_tilde, qt0, qt0T, rate, eps, elbo_scale, nll_scale, reg_scale=None):
        val, grad = native.ctelbo_loss(logits.detach().float().contiguous(), x0.to(torch.int32).contiguous(),
                                       x_tilde.to(torch.int32).contiguous(), qt0.contiguous(), qt0T.contiguous(),
                                       rate.contiguous(), eps, elbo_scale, nll_scale, reg_scale)
        ctx.save_for_backward(grad)
        return val

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return (grad * g,) + (None,) * 9


def _masked_rows(tab, x):
    """tab (B,S,S), x (B,D) -> tab[b, x_bd, :] with entry x_bd zeroed: (B,D,S)."""
    rows = tab[torch.arange(tab.shape[0], device=tab.device).view(-1, 1), x]
    return rows.scatter(-1, x.unsqueeze(-1), 0.0)


def _ct_elbo_terms(logits_reg, logits_sig, x0, reg_x, x_tilde, qt0, rate, eps):
    """Negative CT-ELBO of tauLDR (losses.py:106-278): mean(-sig/norm) + mean(reg)."""
    B = x0.shape[0]
    n = torch.arange(B, device=x0.device).view(B, 1)
    qT, rT = qt0.transpose(1, 2), rate.transpose(1, 2)            # qT[b,x,s0] = qt0[b,s0,x]; rT[b,x,s] = rate[b,s,x]
    p_reg = F.softmax(logits_reg, dim=2)
    reg_tmp = _masked_rows(rT, reg_x) @ qT                        # sum_s (mask*rate[s,x]) qt0[s0,s]
    reg_term = torch.sum((p_reg / (qT[n, reg_x] + eps)) * reg_tmp, dim=(1, 2))
    p_sig = p_reg if logits_sig is logits_reg else F.softmax(logits_sig, dim=2)
    inner = torch.log((p_sig / (qT[n, x_tilde] + eps)) @ qt0 + eps)
    outer_rate = _masked_rows(rT, x_tilde)                        # rate[b, s, x~], s != x~
    q_x0 = qt0[n, x0]                                             # qt0[b, x0, s]
    q_x0_xt = torch.gather(q_x0, -1, x_tilde.unsqueeze(-1)) + eps  # (B,D,1)
    outer = torch.sum(outer_rate * (q_x0 / q_x0_xt) * inner, dim=(1, 2))
    row_sums = -torch.diagonal(rate, dim1=1, dim2=2)              # (B,S)
    base_tmp = row_sums[n, x_tilde]                               # (B,D)
    Z = base_tmp.sum(1).view(B, 1, 1) - base_tmp.unsqueeze(-1) + row_sums.unsqueeze(1)
    sig_norm = torch.sum(outer_rate * q_x0 / (Z * q_x0_xt), dim=(1, 2))
    return torch.mean(-outer / sig_norm) + torch.mean(reg_term)


class _CTElboBase:
    def __init__(self, cfg):
        self.cfg = cfg
        self.ratio_eps = cfg.loss.eps_ratio
        self.nll_weight = cfg.loss.nll_weight
        self.min_time = cfg.loss.min_time
        self.one_forward_pass = cfg.loss.one_forward_pass
        self.max_t = cfg.training.max_t
        self.cross_ent = nn.CrossEntropyLoss()

    def _total(self, state, minibatch, elbo_scale, nll_coef):
        """elbo_scale * neg_elbo + nll_coef * CE(logits, x0).  On a GPU the objective and its logit gradient run in K11
        (csrc/losses.hip): one launch chain with one forward pass, one per network output with two."""
        model = state["model"]
        x0 = _flatten(minibatch).long()
        B, D = x0.shape
        ts = _draw_ts(B, model.device, self.min_time, self.max_t)
        qt0, rate, x_t, x_tilde, qT = _noise(model, x0, ts, True, want_T=True)
        x_logits = model(x_t, ts)
        fused = x_logits.is_cuda and x_logits.shape[-1] <= 256 and getattr(self.cfg.loss, "fused", True)
        if self.one_forward_pass and fused:
            return _CtElboFn.apply(x_logits, x0, x_tilde, qt0, qT, rate, float(self.ratio_eps), float(elbo_scale),
                                   float(nll_coef) / (B * D))
        if fused:
            # two forward passes (losses.py:150-158): the regulariser and the cross entropy see model(x_t) at reg_x = x_t,
            # the signal term sees model(x~) at x~  --  K11 once per network output with the other term's weight at zero
            logits_sig = model(x_tilde, ts)
            eps, w = float(self.ratio_eps), float(elbo_scale)
            reg = _CtElboFn.apply(x_logits, x0, x_t, qt0, qT, rate, eps, 0.0, float(nll_coef) / (B * D), w)
            sig = _CtElboFn.apply(logits_sig, x0, x_tilde, qt0, qT, rate, eps, w, 0.0, 0.0)
            return reg + sig
        if self.one_forward_pass:
            logits_sig, reg_x = x_logits, x_tilde
        else:
            logits_sig, reg_x = model(x_tilde, ts), x_t
        neg_elbo = _ct_elbo_terms(x_logits, logits_sig, x0, reg_x, x_tilde, qt0, rate, self.ratio_eps)
        nll = self.cross_ent(x_logits.permute(0, 2, 1), x0)
        return elbo_scale * neg_elbo + nll_coef * nll


@losses_utils.register_loss
class CTElbo(_CTElboBase):
    def calc_loss(self, state, minibatch, label=None):
        state, minibatch = _unpack(state, minibatch)
        return self._total(state, minibatch, 1.0, self.nll_weight)


@losses_utils.register_loss
class NLL(_CTElboBase):
    """Same sampling path as CTElbo, returns only the cross entropy (losses.py:1504-1778)."""

    def calc_loss(self, state, minibatch, label=None):
        state, minibatch = _unpack(state, minibatch)
        return self._total(state, minibatch, 0.0, 1.0)


@losses_utils.register_loss
class CTElboLambda(_CTElboBase):
    """w * neg_elbo + (1 - w) * nll with w = n_iter / n_iters (losses.py:1783-2058)."""

    def __init__(self, cfg):
        super().__init__(cfg)
        self.max_iter = cfg.training.n_iters

    def calc_loss(self, state, minibatch, label=None):
        state, minibatch = _unpack(state, minibatch)
        w = state["n_iter"] / self.max_iter
        return self._total(state, minibatch, w, 1 - w)


def _crm_loss(cfg, model, xt, t, ll_all, ll_xt):
    """Categorical ratio matching objectives (losses.py:794-836): rm / mle / elbo, per (b,d)."""
    S = cfg.data.S
    lt = cfg.loss.loss_type
    if lt == "rm":
        return -ll_xt
    if lt == "mle":
        return -((S - 1) * ll_xt + torch.sum(utils.log1mexp(ll_all), dim=-1) - utils.log1mexp(ll_xt))
    if lt == "elbo":
        qt0 = model.transition(t)
        n = torch.arange(xt.shape[0], device=xt.device).view(-1, 1)
        d = ll_all - ll_xt.unsqueeze(-1)
        own = F.one_hot(xt, S).to(ll_all.dtype)
        first = torch.sum(torch.exp(d) * qt0.transpose(1, 2)[n, xt] * (1 - own), dim=-1)
        second = torch.sum(-d * qt0[n, xt] * (1 - own), dim=-1)
        return first - second
    raise ValueError("Unknown loss_type: %s" % lt)


class _ScoreElboFn(torch.autograd.Function):
    """ScoreElbo value and logit-gradient in HIP (csrc/losses.hip), direct logits."""

    @staticmethod
    def forward(ctx, logits, x0, x_tilde, reg_x, qt0, rate, eps, nll_scale):
        i32 = lambda t: t.to(torch.int32).contiguous()
        val, grad = native.score_elbo_loss(logits.detach().float().contiguous(), i32(x0), i32(x_tilde), i32(reg_x), qt0.contiguous(),
                                           rate.contiguous(), eps, nll_scale)
        ctx.save_for_backward(grad)
        return val

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return (grad * g,) + (None,) * 7


class _CrmLossFn(torch.autograd.Function):
    """K12 (csrc/losses.hip): value and logit-gradient of the CRM objective in one HIP pass."""

    @staticmethod
    def forward(ctx, logits, xt, x0, qt0, loss_type, scale, nll_scale):
        val, grad = native.crm_loss(logits.detach().float().contiguous(), xt.to(torch.int32).contiguous(),
                                    None if x0 is None else x0.to(torch.int32).contiguous(),
                                    None if qt0 is None else qt0.float().contiguous(), loss_type, scale, nll_scale)
        ctx.save_for_backward(grad)
        return val

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None, None, None, None, None, None


def _tables_with_transpose(model, ts):
    """(q_{t|0}, its transpose), each (B,S,S): the transpose straight out of K1 when the model carries a device process."""
    pr = getattr(model, "process", None)
    if pr is not None and hasattr(pr, "tables"):
        qt0, qT, _, _ = pr.tables(ts, want_qt0=True, want_qt0T=True)
        return qt0, qT
    qt0 = model.transition(ts).float().contiguous()
    return qt0, qt0.transpose(1, 2).contiguous()


class _CrmRevFn(torch.autograd.Function):
    """CRM objectives for logit_type reverse_prob / reverse_logscale in HIP: ctdd_logprob -> K12 on ll_all -> ctdd_logprob_bwd
    (+ CatRMNLL's cross-entropy term on the raw logits inside the last launch)."""

    @staticmethod
    def forward(ctx, logits, xt, x0, qt0, qt0T, logit_type, loss_type, scale, nll_scale):
        lg = logits.detach().float().contiguous()
        xti = xt.to(torch.int32).contiguous()
        tidx = torch.arange(lg.shape[0], dtype=torch.int32, device=lg.device)
        ll_all, _ = native.logprob(lg, xti, qt0, logit_type, tidx, qt0T=qt0T)
        val, dll = native.crm_loss_ll(ll_all, xti, qt0 if loss_type == "elbo" else None, loss_type, scale)
        grad, ce = native.logprob_bwd(logit_type, lg, qt0, qt0T, dll, None if x0 is None else x0.to(torch.int32).contiguous(), nll_scale,
                                      ll_all=ll_all)
        ctx.save_for_backward(grad)
        return val if x0 is None else val + ce

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return (grad * g,) + (None,) * 8


class _ScoreElboRevFn(torch.autograd.Function):
    """ScoreElbo for the reverse logit types in HIP (same three-launch chain)."""

    @staticmethod
    def forward(ctx, logits, x0, x_tilde, reg_x, qt0, qt0T, rate, logit_type, eps, nll_scale):
        lg = logits.detach().float().contiguous()
        i32 = lambda t: t.to(torch.int32).contiguous()
        tidx = torch.arange(lg.shape[0], dtype=torch.int32, device=lg.device)
        ll_all, _ = native.logprob(lg, i32(x_tilde), qt0, logit_type, tidx, qt0T=qt0T)
        val, dll = native.score_elbo_loss_ll(ll_all, i32(x0), i32(x_tilde), i32(reg_x), qt0, rate.contiguous(), eps, nll_scale)
        grad, _ = native.logprob_bwd(logit_type, lg, qt0, qt0T, dll, ll_all=ll_all)
        ctx.save_for_backward(grad)
        return val

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return (grad * g,) + (None,) * 9


def _crm_objective(cfg, model, logits, xt, ts, x0, nll_weight):
    """sum(loss_bd) (1 - ce_coeff) / B [+ nll_weight * CE(logits, x0)]; direct logits on a GPU run in K12."""
    B, D = xt.shape
    scale = (1.0 - cfg.loss.ce_coeff) / B
    if cfg.loss.logit_type == "direct" and logits.is_cuda:
        qt0 = model.transition(ts) if cfg.loss.loss_type == "elbo" else None
        return _CrmLossFn.apply(logits, xt, x0 if nll_weight else None, qt0, cfg.loss.loss_type, scale,
                                float(nll_weight) / (B * D) if nll_weight else 0.0)
    if logits.is_cuda and cfg.loss.logit_type in ("reverse_prob", "reverse_logscale") and logits.shape[-1] <= 256 and getattr(cfg.loss, "fused", True):
        qt0, qT = _tables_with_transpose(model, ts)
        return _CrmRevFn.apply(logits, xt, x0 if nll_weight else None, qt0, qT, cfg.loss.logit_type, cfg.loss.loss_type, scale,
                               float(nll_weight) / (B * D) if nll_weight else 0.0)
    ll_all, ll_xt = get_logprob_with_logits(cfg, model, xt, ts, logits)      # (cfg.device == "cpu", S > 256, cfg.loss.fused = False)
    out = torch.sum(_crm_loss(cfg, model, xt, ts, ll_all, ll_xt)) * scale
    if nll_weight:
        out = out + nll_weight * F.cross_entropy(logits.permute(0, 2, 1), x0)
    return out


class _CRMBase:
    clamp_t = True
    t_hi = 1.0

    def __init__(self, cfg):
        self.cfg = cfg
        self.ratio_eps = cfg.loss.eps_ratio
        self.min_time = cfg.loss.min_time
        self.S = cfg.data.S
        self.D = cfg.model.concat_dim

    def _forward(self, state, minibatch):
        model = state["model"]
        x0 = _flatten(minibatch).long()
        B = x0.shape[0]
        ts = _draw_ts(B, model.device, self.min_time, self.t_hi)
        if self.clamp_t:
            ts = torch.clamp(ts, max=0.99999)
        _, _, xt, _ = _noise(model, x0, ts, False)
        return model, x0, ts, xt


@losses_utils.register_loss
class CatRM(_CRMBase):
    def calc_loss(self, state, minibatch, label=None):
        state, minibatch = _unpack(state, minibatch)
        model, x0, ts, xt = self._forward(state, minibatch)
        logits = model(xt, ts)
        return _crm_objective(self.cfg, model, logits, xt, ts, None, 0.0)


@losses_utils.register_loss
class CatRMNLL(_CRMBase):
    """CatRM + nll_weight * CE(logits, x0), t ~ U(min_time, max_t) without clamp (losses.py:1135-1242)."""
    clamp_t = False

    def __init__(self, cfg):
        super().__init__(cfg)
        self.t_hi = cfg.training.max_t
        self.nll_weight = cfg.loss.nll_weight
        self.cross_ent = nn.CrossEntropyLoss()

    def calc_loss(self, minibatch, state=None, label=None):
        state, minibatch = _unpack(minibatch, state)
        model, x0, ts, xt = self._forward(state, minibatch)
        logits = model(xt, ts)
        return _crm_objective(self.cfg, model, logits, xt, ts, x0, self.nll_weight)


@losses_utils.register_loss
class NLLOriginal(_CRMBase):
    """Plain denoising cross entropy, t ~ U(min_time, 1) unclamped (losses.py:1049-1103)."""
    clamp_t = False

    def __init__(self, cfg):
        super().__init__(cfg)
        self.cross_ent = nn.CrossEntropyLoss()

    def calc_loss(self, state, minibatch, label=None):
        state, minibatch = _unpack(state, minibatch)
        model, x0, ts, xt = self._forward(state, minibatch)
        logits = model(xt, ts) if label is None else model(xt, ts, label)
        return self.cross_ent(logits.permute(0, 2, 1), x0)


@losses_utils.register_loss
class ScoreElbo:
    """CT-ELBO evaluated with SDDM ratios exp(ll_all - ll_xt) + nll_weight * mean(-ll_xt)
    (losses.py:1246-1500)."""

    def __init__(self, cfg):
        self.cfg = cfg
        self.ratio_eps = cfg.loss.eps_ratio
        self.nll_weight = cfg.loss.nll_weight
        self.min_time = cfg.loss.min_time
        self.one_forward_pass = cfg.loss.one_forward_pass

    def calc_loss(self, minibatch, state=None, label=None):
        state, minibatch = _unpack(minibatch, state)
        model = state["model"]
        x0 = _flatten(minibatch).long()
        B = x0.shape[0]
        eps = self.ratio_eps
        ts = torch.clamp(_draw_ts(B, model.device, self.min_time, 1.0), max=0.99999)
        qt0, rate, x_t, x_tilde = _noise(model, x0, ts, True)
        reg_x = x_tilde if self.one_forward_pass else x_t
        logits = model(reg_x, ts)
        if self.cfg.loss.logit_type == "direct" and logits.is_cuda and logits.shape[-1] <= 256 and getattr(self.cfg.loss, "fused", True):
            return _ScoreElboFn.apply(logits, x0, x_tilde, reg_x, qt0, rate, float(eps), float(self.nll_weight) / B)
        if (self.cfg.loss.logit_type in ("reverse_prob", "reverse_logscale") and logits.is_cuda and logits.shape[-1] <= 256
                and getattr(self.cfg.loss, "fused", True)):
            _, qT = _tables_with_transpose(model, ts)
            return _ScoreElboRevFn.apply(logits, x0, x_tilde, reg_x, qt0.contiguous(), qT, rate, self.cfg.loss.logit_type, float(eps),
                                         float(self.nll_weight) / B)
        n = torch.arange(B, device=x0.device).view(B, 1)
        rT = rate.transpose(1, 2)
        ll_all, ll_xt = get_logprob_with_logits(self.cfg, model, x_tilde, ts, logits)
        d = ll_all - ll_xt.unsqueeze(-1)
        reg_term = torch.sum(torch.exp(d) * _masked_rows(rT, reg_x), dim=(1, 2))
        outer_rate = _masked_rows(rT, x_tilde)
        q_x0 = qt0[n, x0]
        q_x0_xt = torch.gather(q_x0, -1, x_tilde.unsqueeze(-1)) + eps
        outer = torch.sum(outer_rate * (q_x0 / q_x0_xt) * d, dim=(1, 2))
        row_sums = -torch.diagonal(rate, dim1=1, dim2=2)
        base_tmp = row_sums[n, x_tilde]
        Z = base_tmp.sum(1).view(B, 1, 1) - base_tmp.unsqueeze(-1) + row_sums.unsqueeze(1)
        sig_norm = torch.sum(outer_rate * q_x0 / (Z * q_x0_xt), dim=(1, 2))
        neg_elbo = torch.mean(-outer / sig_norm) + torch.mean(reg_term)
        return neg_elbo + self.nll_weight * (torch.sum(-ll_xt) / B)
