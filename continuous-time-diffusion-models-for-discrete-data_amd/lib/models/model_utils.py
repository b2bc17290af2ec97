"""Model registry + SDDM log-probabilities (reference lib/models/model_utils.py:5-60)."""
import torch
import torch.nn.functional as F

from ctdd import native

_MODELS = {}


def register_model(cls):
    name = cls.__name__
    if name in _MODELS:
        raise ValueError(f"{name} is already registered!")
    _MODELS[name] = cls
    return cls


def get_model(name):
    return _MODELS[name]


def create_model(cfg, device, encoding=None, rank=None):
    ctor = get_model(cfg.model.name)
    model = ctor(cfg, device, rank) if encoding is None else ctor(cfg, device, encoding, rank)
    return model.to(device)


def _as_i32(x):
    return x.to(torch.int32).contiguous()


def get_logprob_with_logits(cfg, model, xt, t, logits, xt_target=None):
    """log p_t(x^d = . | x^{\\d}) for every state and at x_t (model_utils.py:30-60).

    Inference (no autograd) runs the fused HIP kernel; when `logits` carries a graph the same
    three formulas are evaluated with differentiable device ops."""
    S = cfg.data.S
    logit_type = cfg.loss.logit_type
    if logit_type not in native.LOGIT_TYPES:
        raise ValueError("Unknown logit_type: %s" % logit_type)
    if xt_target is None:
        xt_target = xt
    needs_graph = torch.is_grad_enabled() and logits.requires_grad
    if not needs_graph and logits.dim() == 3:
        qt0 = None if logit_type == "direct" else model.transition(t)
        B = logits.shape[0]
        tidx = torch.arange(B, dtype=torch.int32, device=logits.device)
        return native.logprob(logits.float().contiguous(), _as_i32(xt_target), qt0, logit_type, tidx)
    if logit_type == "direct":
        log_prob = F.log_softmax(logits, dim=-1)
    else:
        qt0 = model.transition(t)                                  # (B,S,S)
        mid = [1] * (xt.dim() - 2)                                  # dims between batch and D
        if logit_type == "reverse_prob":
            p0t = F.softmax(logits, dim=-1)
            log_prob = torch.log(p0t @ qt0.view(qt0.shape[0], *mid, S, S) + 1e-35)
        else:
            log_qt0 = torch.where(qt0 <= 1e-35, -1e9, torch.log(qt0)).view(qt0.shape[0], *mid, 1, S, S)
            log_prob = torch.logsumexp(F.log_softmax(logits, dim=-1).unsqueeze(-1) + log_qt0, dim=-2)
    log_xt = torch.gather(log_prob, -1, xt_target.long().unsqueeze(-1)).squeeze(-1)
    return log_prob, log_xt
