"""CPU: the oracle (CPU restatement) against golden vectors frozen from the reference itself
(oracle/gen_golden.py).  Pins P1-P6 of SURVEY.md 8(c)."""
import ast

import numpy as np
import pytest
import torch

from oracle import ctmc_ops as ops
from oracle import samplers as osamp
from oracle.forward_process import ForwardProcess
from oracle.toy_model import ToyModel

torch.set_num_threads(1)
T = torch.from_numpy


def gauss(S):
    return ForwardProcess("gaussian", S, rate_sigma=6.0, Q_sigma=512.0, time_exp=100.0, time_base=3.0)


def univar(S, tf):
    kw = dict(rate_const=1.7, t_func=tf)
    if tf == "log":
        kw.update(time_base=3.0, time_exp=100.0)
    return ForwardProcess("univar", S, **kw)


def make_process(kind, S, t_func="sqrt_cos"):
    if kind == "gaussian":
        return gauss(S)
    if kind == "univar":
        return univar(S, t_func)
    if kind == "uniform":
        return ForwardProcess("uniform", S, rate_const=1.7)
    raise ValueError(kind)


# ------------------------------------------------------------------ P1
def test_forward_process_rate_matrices_bit_exact(golden):
    g = golden("forward_process")
    p8, p256 = gauss(8), gauss(256)
    assert np.array_equal(p8.base_rate.numpy(), g["g8_base_rate"])
    rows = g["g256_rows"]
    assert np.array_equal(p256.base_rate[rows].numpy(), g["g256_base_rate_rows"])
    assert np.array_equal(p256.base_rate.abs().sum(1).numpy(), g["g256_base_rate_rowsum_abs"])
    assert np.array_equal(ForwardProcess("uniform", 3, rate_const=1.7).base_rate.numpy(), g["u3_rate_matrix"])
    assert np.array_equal(ForwardProcess("birthdeath", 8, sigma_min=1.0, sigma_max=100.0).base_rate.numpy(), g["bd8_base_rate"])


def test_forward_process_tables(golden):
    g = golden("forward_process")
    ts = T(g["ts"])
    p8, p256 = gauss(8), gauss(256)
    np.testing.assert_allclose(p8.transition(ts).numpy(), g["g8_qt0"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(p8.rate(ts).numpy(), g["g8_rate"], rtol=1e-6)
    q = p256.transition(ts)
    rows = g["g256_rows"]
    np.testing.assert_allclose(q[:, rows].numpy(), g["g256_qt0_rows"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(q.sum(1).numpy(), g["g256_qt0_colsum"], rtol=1e-5)
    assert np.array_equal((q > 0).sum(-1).numpy(), g["g256_qt0_nnz"])
    np.testing.assert_allclose(p256.rate(ts)[:, rows].numpy(), g["g256_rate_rows"], rtol=1e-6)
    np.testing.assert_allclose(p8.transit_between(ts * 0.5, ts).numpy(), g["g8_between"], rtol=1e-5, atol=1e-7)
    y = torch.tensor([[0, 3, 7], [1, 1, 2], [5, 6, 0], [4, 4, 4], [7, 0, 2]])
    np.testing.assert_allclose(p8.rate_mat(y, ts).numpy(), g["g8_rate_mat"], rtol=1e-6)
    u3 = ForwardProcess("uniform", 3, rate_const=1.7)
    np.testing.assert_allclose(u3.transition(ts).numpy(), g["u3_qt0"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(u3.rate(ts).numpy(), g["u3_rate"], rtol=1e-6)
    np.testing.assert_allclose(u3.transit_between(torch.tensor([0.1, 0.2]), torch.tensor([0.4, 0.9])).numpy(),
                               g["u3_between"], rtol=1e-5, atol=1e-7)
    bd = ForwardProcess("birthdeath", 8, sigma_min=1.0, sigma_max=100.0)
    np.testing.assert_allclose(bd.transition(ts).numpy(), g["bd8_qt0"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(bd.rate(ts).numpy(), g["bd8_rate"], rtol=1e-6)
    tsv, y = T(g["ts_univar"]), T(g["univar_y"])
    for S, tf in ((2, "log_sqr"), (3, "sqrt_cos"), (3, "log"), (2, "sqrt_cos")):
        p, k = univar(S, tf), f"v{S}_{tf}"
        np.testing.assert_allclose(p.transition(tsv).numpy(), g[k + "_qt0"], rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(p.rate(tsv).numpy(), g[k + "_rate"], rtol=1e-6)
        np.testing.assert_allclose(p.rate_mat(y % S, tsv).numpy(), g[k + "_rate_mat"], rtol=1e-6)
        np.testing.assert_allclose(p.transit_between(tsv * 0.5, tsv).numpy(), g[k + "_between"], rtol=1e-5, atol=1e-7)


# ------------------------------------------------------------------ P2
@pytest.mark.parametrize("tag,kind,S,tf", [("g16", "gaussian", 16, None), ("g256", "gaussian", 256, None),
                                           ("v3", "univar", 3, "sqrt_cos"), ("v2", "univar", 2, "log_sqr")])
def test_noising_indices_bit_exact(golden, tag, kind, S, tf):
    g = golden("noising")
    proc = make_process(kind, S, tf)
    x0, ts = T(g[f"{tag}_x0"]), T(g[f"{tag}_ts"])
    qt0, rate = proc.transition(ts), proc.rate(ts)
    x_t = ops.noise_xt(qt0, x0, T(g[f"{tag}_E_xt"]))
    assert torch.equal(x_t, T(g[f"{tag}_x_t"]).long()), "x_t indices must be bit-exact"
    _, _, x_tilde = ops.xtilde_sample(rate, x_t, T(g[f"{tag}_E_dim"]), T(g[f"{tag}_E_val"]))
    assert torch.equal(x_tilde, T(g[f"{tag}_x_tilde"]).long()), "x_tilde must be bit-exact"
    # exactly one entry differs per batch row
    assert ((x_tilde != x_t).sum(1) == 1).all()


# ------------------------------------------------------------------ P3 / P4
@pytest.mark.parametrize("tag,kind,S", [("g256", "gaussian", 256), ("g16", "gaussian", 16),
                                        ("v3", "univar", 3), ("u3", "uniform", 3)])
def test_logprob_and_reverse_rates(golden, tag, kind, S):
    g = golden("rates")
    proc = make_process(kind, S)
    logits, x, t = T(g[f"{tag}_logits"]), T(g[f"{tag}_x"]), T(g[f"{tag}_t"])
    qt0, rate = proc.transition(t), proc.rate(t)
    for lt in ("direct", "reverse_prob", "reverse_logscale"):
        ll_all, ll_xt = ops.logprob_with_logits(lt, logits, x, qt0)
        np.testing.assert_allclose(ll_all.numpy(), g[f"{tag}_{lt}_ll_all"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(ll_xt.numpy(), g[f"{tag}_{lt}_ll_xt"], rtol=1e-5, atol=1e-6)
        rr, ratio = ops.reverse_rates_crm(lt, logits, x, qt0, rate)
        np.testing.assert_allclose(rr.numpy(), g[f"{tag}_{lt}_crm_rates"], rtol=1e-4, atol=1e-30)
        np.testing.assert_allclose(ratio.numpy(), g[f"{tag}_{lt}_crm_ratio"], rtol=1e-4, atol=1e-30)
    N = x.shape[0]
    for nm, tt in (("shared", torch.full((N,), 0.37)), ("perrow", t)):
        rr, ratio = ops.reverse_rates_ctelbo(logits, x, proc.transition(tt), proc.rate(tt), 1e-9)
        np.testing.assert_allclose(rr.numpy(), g[f"{tag}_ctelbo_{nm}_rates"], rtol=1e-4, atol=1e-30)
        np.testing.assert_allclose(ratio.numpy(), g[f"{tag}_ctelbo_{nm}_ratio"], rtol=1e-4, atol=1e-30)


# ------------------------------------------------------------------ P5 / P6
def _sampler_cases():
    import os
    path = os.path.join(os.path.dirname(__file__), "golden", "samplers.npz")
    return sorted(k[: -len("__meta")] for k in np.load(path).files if k.endswith("__meta"))


@pytest.mark.parametrize("tag", _sampler_cases())
def test_sampler_replay_exact(golden, tag):
    """Seeded CPU replay of the reference's sampler loops: identical integer outputs."""
    g = golden("samplers")
    m = ast.literal_eval(str(g[f"{tag}__meta"]))
    S, D, N = m["S"], m["D"], m["N"]
    model = ToyModel(make_process(m["kind"], S, m["t_func"]), S, scale=m["scale"])
    torch.manual_seed(m["seed"])
    common = dict(min_t=m["min_t"], num_steps=m["num_steps"], initial_dist=m["initial_dist"], eps_ratio=m["eps_ratio"])
    if m["sampler"] == "TauL":
        out = osamp.taul_sample(model, N, D, S, max_t=m["max_t"], init_std=512.0, is_ordinal=m["is_ordinal"],
                                loss_name=m["loss"], logit_type=m["logit_type"],
                                corrector_entry_time=m["corrector_entry_time"],
                                num_corrector_steps=m["num_corrector_steps"], **common)
    elif m["sampler"] == "LBJF":
        out = osamp.lbjf_sample(model, N, D, S, max_t=m["max_t"], init_std=512.0, loss_name=m["loss"],
                                logit_type=m["logit_type"], corrector_entry_time=m["corrector_entry_time"],
                                num_corrector_steps=m["num_corrector_steps"], **common)
    elif m["sampler"] == "MidPointTauL":
        out = osamp.midpoint_sample(model, N, D, S, max_t=m["max_t"], init_std=512.0, is_ordinal=m["is_ordinal"],
                                    loss_name=m["loss"], logit_type=m["logit_type"], **common)
    elif m["sampler"] == "ExactSampling":
        out = osamp.exact_sample(model, N, D, S, max_t=m["max_t"], min_t=m["min_t"], num_steps=m["num_steps"],
                                 initial_dist=m["initial_dist"], init_std=512.0)
    else:
        out = (osamp.pctaul_sample(model, N, D, S, corrector_entry_time=m["corrector_entry_time"],
                                   num_corrector_steps=m["num_corrector_steps"],
                                   corrector_step_size_multiplier=m["corrector_step_size_multiplier"], **common),)
    assert torch.equal(model.calls[0][0].long(), T(g[f"{tag}__x_init"]).long())
    assert np.array_equal(out[0], g[f"{tag}__samples"]), "sampler output must replay exactly"
    for i, extra in enumerate(out[1:]):
        np.testing.assert_allclose(np.asarray(extra, dtype=np.float64), g[f"{tag}__aux{i}"], rtol=0, atol=1e-12, equal_nan=True)


# ------------------------------------------------------------------ P8 (U-Net)
def load_unet_case(g, tag):
    cfg = ast.literal_eval(str(g[f"{tag}__cfg"]))
    pre = f"{tag}__sd__"
    sd = {k[len(pre):]: T(v) for k, v in g.items() if k.startswith(pre)}
    return cfg, sd, T(g[f"{tag}__x"]), T(g[f"{tag}__t"]), g[f"{tag}__out"]


@pytest.mark.parametrize("tag", ["logits", "logistic"])
def test_unet_oracle_matches_reference(golden, tag):
    from oracle import nets
    cfg, sd, x, t, out = load_unet_case(golden("unet"), tag)
    got = nets.image_model_forward(sd, x, t, **cfg)
    np.testing.assert_allclose(got.numpy(), out, rtol=0, atol=1e-4)       # BASELINE bar: logits within 1e-4 fp32
    assert np.abs(out).max() > 1.0                                          # non-degenerate fixture


# ------------------------------------------------------------------ P7 (losses)
def _loss_cases():
    import os
    path = os.path.join(os.path.dirname(__file__), "golden", "losses.npz")
    return sorted(k[: -len("__meta")] for k in np.load(path).files if k.endswith("__meta"))


def oracle_loss_from_golden(g, tag, model_factory):
    """Rebuild the reference's noise draw from the stored E noise and evaluate the oracle loss.
    model_factory(kind, S, t_func) -> (model object with transition/rate, call(x,t)->logits)."""
    from oracle import losses as ol
    m = ast.literal_eval(str(g[f"{tag}__meta"]))
    S, B, D = m["S"], m["B"], m["D"]
    x0, u = T(g[f"{tag}__x0"]), T(g[f"{tag}__u"])
    name = m["loss"]
    hi = m["max_t"] if name in ("CTElbo", "NLL", "CTElboLambda", "CatRMNLL") else 1.0
    ts = u * (hi - m["min_time"]) + m["min_time"]
    if name in ("CatRM", "ScoreElbo"):
        ts = torch.clamp(ts, max=0.99999)
    model = model_factory(m["kind"], S, m["t_func"], m["theta"])
    qt0, rate = model.transition(ts), model.rate(ts)
    if name == "ScoreElbo":             # Categorical(probs=rows): rows / rowsum
        rows = qt0[torch.arange(B).view(B, 1), x0.long()].reshape(B * D, S)
        x_t = ops.exp_race_argmax(rows / rows.sum(-1, keepdim=True), T(g[f"{tag}__E_xt"])).view(B, D)
    else:
        x_t = ops.noise_xt(qt0, x0, T(g[f"{tag}__E_xt"]))
    x_tilde = None
    if f"{tag}__E_dim" in g:
        if name == "ScoreElbo":         # probs given directly (no log / -1e9 floor)
            rv = ops.zero_own_state(rate[torch.arange(B).view(B, 1), x_t.long()], x_t)
            ds = rv.sum(2)
            dims = ops.exp_race_argmax(ds / ds.sum(-1, keepdim=True), T(g[f"{tag}__E_dim"]))
            newp = rv[torch.arange(B), dims]
            newval = ops.exp_race_argmax(newp / newp.sum(-1, keepdim=True), T(g[f"{tag}__E_val"]))
            x_tilde = x_t.clone()
            x_tilde[torch.arange(B), dims] = newval
        else:
            _, _, x_tilde = ops.xtilde_sample(rate, x_t, T(g[f"{tag}__E_dim"]), T(g[f"{tag}__E_val"]))
    if name in ("CTElbo", "NLL", "CTElboLambda"):
        val = ol.ct_elbo_family(name, model, x0, ts, x_t, x_tilde, eps=m["eps_ratio"], nll_weight=m["nll_weight"],
                                one_forward_pass=m["one_forward_pass"], weight=m["n_iter"] / m["n_iters"])
    elif name in ("CatRM", "CatRMNLL", "NLLOriginal"):
        val = ol.crm_family(name, model, x0, ts, x_t, S=S, logit_type=m["logit_type"], loss_type=m["loss_type"],
                            ce_coeff=m["ce_coeff"], nll_weight=m["nll_weight"])
    else:
        val = ol.score_elbo(model, x0, ts, x_t, x_tilde, logit_type=m["logit_type"], eps=m["eps_ratio"],
                            nll_weight=m["nll_weight"], one_forward_pass=m["one_forward_pass"])
    return m, ts, x_t, x_tilde, model, val


class ThetaToy:
    """toy score function scaled by a differentiable theta (the golden generator's model)."""

    def __init__(self, proc, S, theta):
        from oracle.toy_model import toy_logits
        self.proc, self.S, self.f = proc, S, toy_logits
        self.theta = torch.tensor(float(theta), requires_grad=True)
        self.first = None

    def __call__(self, x, t, *a):
        if self.first is None:
            self.first = x.clone()
        return self.f(x, t, self.S, 1.0) * self.theta

    def transition(self, t):
        return self.proc.transition(t)

    def rate(self, t):
        return self.proc.rate(t)


@pytest.mark.parametrize("tag", _loss_cases())
def test_losses_match_reference(golden, tag):
    g = golden("losses")
    fac = lambda kind, S, tf, th: ThetaToy(make_process(kind, S, tf), S, th)
    m, ts, x_t, x_tilde, model, val = oracle_loss_from_golden(g, tag, fac)
    np.testing.assert_allclose(ts.numpy(), g[f"{tag}__ts"], rtol=0, atol=0)
    assert torch.equal(model.first.long(), T(g[f"{tag}__x_first"]).long()), "noised state fed to the network must replay exactly"
    np.testing.assert_allclose(val.item(), float(g[f"{tag}__loss"]), rtol=2e-5)
    grad, = torch.autograd.grad(val, model.theta)
    # gradient through (B,D,S)x(S,S) fp32 contractions with ~1e9 dynamic range at S=256: 1e-3
    np.testing.assert_allclose(grad.item(), float(g[f"{tag}__grad"]), rtol=1e-3, atol=1e-6)


# ------------------------------------------------------------------ P8 (hollow transformer)
def load_hollow_case(g, tag):
    cfg = ast.literal_eval(str(g[f"{tag}__cfg"]))
    pre = f"{tag}__sd__"
    sd = {k[len(pre):]: T(v) for k, v in g.items() if k.startswith(pre)}
    return cfg, sd, T(g[f"{tag}__x"]), T(g[f"{tag}__t"]), g[f"{tag}__out"]


@pytest.mark.parametrize("tag", ["s3", "s2"])
def test_hollow_oracle_matches_reference(golden, tag):
    from oracle import nets
    cfg, sd, x, t, out = load_hollow_case(golden("hollow"), tag)
    got = nets.hollow_forward(sd, x, t, S=cfg["S"], embed_dim=cfg["embed_dim"], num_layers=cfg["num_layers"],
                              num_heads=cfg["num_heads"], time_scale_factor=cfg["time_scale_factor"])
    np.testing.assert_allclose(got.numpy(), out, rtol=0, atol=1e-4)
    assert np.abs(out).max() > 0.5
