"""GPU: the product losses (lib.losses.losses: HIP noising + device objectives) against the oracle.
(a) objective parity on identical noise (test hook _FIXED_NOISE) incl. the gradient w.r.t. a scalar
    model parameter; (b) the HIP noising path inside calc_loss: finite, autograd-connected, and the
    mean loss over seeds agrees with the oracle's mean over its own draws."""
import ast

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from test_oracle_golden import ThetaToy, _loss_cases, make_process, oracle_loss_from_golden

T = torch.from_numpy
GAUSS = dict(rate_sigma=6.0, Q_sigma=512.0, time_exp=100.0, time_base=3.0)


class DeviceThetaToy:
    def __init__(self, kind, S, t_func, theta):
        from ctdd.process import DeviceForwardProcess
        from oracle.toy_model import toy_logits
        p = GAUSS if kind == "gaussian" else dict(rate_const=1.7, t_func=t_func)
        self.process = DeviceForwardProcess(kind, S, "cuda", **p)
        self.S, self.device, self.f = S, torch.device("cuda"), toy_logits
        self.theta = torch.tensor(float(theta), device="cuda", requires_grad=True)

    def __call__(self, x, t, *a):
        return self.f(x, t, self.S, 1.0) * self.theta

    def transition(self, t):
        return self.process.transition(t)

    def rate(self, t):
        return self.process.rate(t)


def _cfg(m):
    from config.mnist_config.config_tauUnet_mnist import get_config
    c = get_config()
    c.data.S, c.model.concat_dim = m["S"], m["D"]
    c.loss.update(name=m["loss"], eps_ratio=m["eps_ratio"], nll_weight=m["nll_weight"], min_time=m["min_time"],
                  one_forward_pass=m["one_forward_pass"], logit_type=m["logit_type"], loss_type=m["loss_type"],
                  ce_coeff=m["ce_coeff"])
    c.training.max_t, c.training.n_iters = m["max_t"], m["n_iters"]
    return c


@pytest.mark.parametrize("tag", _loss_cases())
def test_objective_matches_oracle_on_identical_noise(golden, tag):
    import lib.losses.losses as L
    import lib.losses.losses_utils as lu
    g = golden("losses")
    fac = lambda kind, S, tf, th: ThetaToy(make_process(kind, S, tf), S, th)
    m, ts, x_t, x_tilde, omodel, oval = oracle_loss_from_golden(g, tag, fac)
    ograd, = torch.autograd.grad(oval, omodel.theta)
    model = DeviceThetaToy(m["kind"], m["S"], m["t_func"], m["theta"])
    loss = lu.get_loss(_cfg(m))
    L._FIXED_NOISE = {"ts": ts, "x_t": x_t, "x_tilde": x_tilde if x_tilde is not None else x_t}
    from ctdd import native
    before = dict(native.LAUNCH_COUNTS)
    try:
        state = {"model": model, "n_iter": m["n_iter"]}
        x0 = T(g[f"{tag}__x0"]).cuda()
        val = loss.calc_loss(x0, state) if m["loss"] in ("CatRMNLL", "ScoreElbo") else loss.calc_loss(state, x0)
        grad, = torch.autograd.grad(val, model.theta)
    finally:
        L._FIXED_NOISE = None
    if m["loss"] in ("CatRM", "CatRMNLL", "ScoreElbo") and m["logit_type"] != "direct":
        # the reverse logit types (what every shipped hollow config uses) run the HIP chain, not torch device ops
        ran = {k: native.LAUNCH_COUNTS.get(k, 0) - before.get(k, 0) for k in ("ctdd_logprob_bwd", "ctdd_crm_loss_ll", "ctdd_score_elbo_loss_ll")}
        assert ran["ctdd_logprob_bwd"] == 1 and ran["ctdd_crm_loss_ll"] + ran["ctdd_score_elbo_loss_ll"] == 1, ran
    if m["loss"] in ("CTElbo", "NLL", "CTElboLambda"):
        # K11 for both settings of one_forward_pass: one launch chain, or one per network output (reg at x_t + signal at x~)
        ran = {k: native.LAUNCH_COUNTS.get(k, 0) - before.get(k, 0) for k in ("ctdd_ctelbo_loss", "ctdd_ctelbo_loss_terms")}
        assert ran == ({"ctdd_ctelbo_loss": 1, "ctdd_ctelbo_loss_terms": 0} if m["one_forward_pass"] else
                       {"ctdd_ctelbo_loss": 0, "ctdd_ctelbo_loss_terms": 2}), ran
    # same tables on both sides would make this ~1e-6; the GPU builds q_{t|0} itself (K1), rtol 2e-4
    np.testing.assert_allclose(val.item(), oval.item(), rtol=2e-4, atol=1e-6)
    # two forward passes: d/dtheta is the difference of two halves of ~0.05 each (regulariser at model(x_t), signal at model(x~)),
    # each carrying the fp32 softmax-backward cancellation of ~1e-3 of its scale at q_{t|0} entries of 1e-9 (measured against fp64
    # autograd on the same tables: HIP 5e-4 / 1.2e-3, torch fp32 3e-4 / 5e-4 per element): absolute bound 2e-3 of the halves
    np.testing.assert_allclose(grad.item(), ograd.item(), rtol=2e-3, atol=1e-5 if m["one_forward_pass"] else 1e-4)
    np.testing.assert_allclose(val.item(), float(g[f"{tag}__loss"]), rtol=3e-4, atol=1e-6)   # = the reference's value


@pytest.mark.parametrize("tag", ["ctelbo_g16", "catrm_v3_rm", "score_v3", "catrmnll_v2"])
def test_hip_noising_inside_loss(golden, tag):
    import lib.losses.losses_utils as lu
    import lib.losses.losses  # noqa: F401
    g = golden("losses")
    m = ast.literal_eval(str(g[f"{tag}__meta"]))
    model = DeviceThetaToy(m["kind"], m["S"], m["t_func"], m["theta"])
    loss = lu.get_loss(_cfg(m))
    x0 = T(g[f"{tag}__x0"]).cuda().repeat(64, 1)            # bigger batch: tighter mean
    vals = []
    for seed in range(6):
        torch.manual_seed(seed)
        state = {"model": model, "n_iter": m["n_iter"]}
        v = loss.calc_loss(x0, state) if m["loss"] in ("CatRMNLL", "ScoreElbo") else loss.calc_loss(state, x0)
        assert torch.isfinite(v) and v.requires_grad
        vals.append(v.item())
    # oracle mean over its own draws (torch CPU RNG), same batch
    from oracle import ctmc_ops as ops, losses as ol
    proc = make_process(m["kind"], m["S"], m["t_func"])
    om = ThetaToy(proc, m["S"], m["theta"])
    ovals = []
    x0c = x0.cpu()
    B, D = x0c.shape
    for seed in range(6):
        torch.manual_seed(100 + seed)
        hi = m["max_t"] if m["loss"] in ("CTElbo", "CatRMNLL") else 1.0
        ts = torch.rand(B) * (hi - m["min_time"]) + m["min_time"]
        if m["loss"] in ("CatRM", "ScoreElbo"):
            ts = ts.clamp(max=0.99999)
        qt0, rate = proc.transition(ts), proc.rate(ts)
        x_t = ops.noise_xt(qt0, x0c, torch.empty(B * D, m["S"]).exponential_(1))
        if m["loss"] in ("CTElbo", "ScoreElbo"):
            _, _, xtl = ops.xtilde_sample(rate, x_t, torch.empty(B, D).exponential_(1), torch.empty(B, m["S"]).exponential_(1))
        if m["loss"] == "CTElbo":
            ov = ol.ct_elbo_family("CTElbo", om, x0c, ts, x_t, xtl, eps=m["eps_ratio"], nll_weight=m["nll_weight"], one_forward_pass=m["one_forward_pass"])
        elif m["loss"] == "ScoreElbo":
            ov = ol.score_elbo(om, x0c, ts, x_t, xtl, logit_type=m["logit_type"], eps=m["eps_ratio"], nll_weight=m["nll_weight"], one_forward_pass=m["one_forward_pass"])
        else:
            ov = ol.crm_family(m["loss"], om, x0c, ts, x_t, S=m["S"], logit_type=m["logit_type"], loss_type=m["loss_type"], ce_coeff=m["ce_coeff"], nll_weight=m["nll_weight"])
        ovals.append(ov.item())
    a, b = np.array(vals), np.array(ovals)
    se = np.sqrt(a.var(ddof=1) / len(a) + b.var(ddof=1) / len(b)) + 1e-9
    assert abs(a.mean() - b.mean()) < 6 * se + 0.02 * abs(b.mean()), (a, b)


@pytest.mark.parametrize("S,D", [(3, 15), (256, 37), (100, 8)])
@pytest.mark.parametrize("loss_type", ["rm", "mle", "elbo"])
def test_crm_kernel_matches_oracle_formulas(S, D, loss_type):
    """K12 (ctdd_crm_loss) against the oracle's CRM objective + CE on random logits: value and d/dlogits."""
    from ctdd import native
    from oracle import losses as ol, ctmc_ops as ops
    import torch.nn.functional as F
    B = 3
    g = torch.Generator().manual_seed(S + D)
    logits = torch.randn(B, D, S, generator=g) * 2.0
    xt = torch.randint(0, S, (B, D), generator=g)
    x0 = torch.randint(0, S, (B, D), generator=g)
    qt0 = torch.softmax(torch.randn(B, S, S, generator=g), -1)
    scale, nllw = 0.7 / B, 0.05
    lo = logits.clone().requires_grad_(True)
    ll_all, ll_xt = ops.logprob_with_logits("direct", lo, xt, qt0)
    want = torch.sum(ol.crm_comp_loss(loss_type, S, ll_all, ll_xt, xt, qt0)) * scale + nllw * F.cross_entropy(lo.permute(0, 2, 1), x0)
    wgrad, = torch.autograd.grad(want, lo)
    val, grad = native.crm_loss(logits.cuda(), xt.to(torch.int32).cuda(), x0.to(torch.int32).cuda(), qt0.cuda(), loss_type, scale,
                                nllw / (B * D))
    np.testing.assert_allclose(val.item(), want.item(), rtol=2e-5)
    np.testing.assert_allclose(grad.cpu().numpy(), wgrad.numpy(), rtol=2e-4, atol=2e-6)


@pytest.mark.parametrize("S,D", [(3, 15), (256, 37), (100, 9)])
@pytest.mark.parametrize("logit_type", ["reverse_prob", "reverse_logscale"])
@pytest.mark.parametrize("loss_type", ["rm", "mle", "elbo"])
def test_crm_reverse_logit_types_match_oracle_formulas(S, D, logit_type, loss_type):
    """ctdd_logprob -> ctdd_crm_loss_ll -> ctdd_logprob_bwd against autograd through the oracle's restatement of
    get_logprob_with_logits (model_utils.py:42-56) + the CRM objective + CE on random logits: value and d/dlogits."""
    from ctdd import native
    from oracle import losses as ol, ctmc_ops as ops
    import lib.losses.losses as L
    import torch.nn.functional as F
    B = 3
    g = torch.Generator().manual_seed(S * 7 + D)
    logits = torch.randn(B, D, S, generator=g) * 2.0
    xt = torch.randint(0, S, (B, D), generator=g)
    x0 = torch.randint(0, S, (B, D), generator=g)
    qt0 = torch.softmax(torch.randn(B, S, S, generator=g) * 2, -1)
    qt0[qt0 < 1e-4] = 0.0                                        # exact zeros as the clamped transition tables have
    scale, nllw = 0.7 / B, 0.05
    lo = logits.clone().requires_grad_(True)
    ll_all, ll_xt = ops.logprob_with_logits(logit_type, lo, xt, qt0)
    want = torch.sum(ol.crm_comp_loss(loss_type, S, ll_all, ll_xt, xt, qt0)) * scale + nllw * F.cross_entropy(lo.permute(0, 2, 1), x0)
    wgrad, = torch.autograd.grad(want, lo)
    lg = logits.cuda().requires_grad_(True)
    q = qt0.cuda()
    val = L._CrmRevFn.apply(lg, xt.cuda(), x0.cuda(), q, q.transpose(1, 2).contiguous(), logit_type, loss_type, scale, nllw / (B * D))
    grad, = torch.autograd.grad(val, lg)
    np.testing.assert_allclose(val.item(), want.item(), rtol=3e-5)
    np.testing.assert_allclose(grad.cpu().numpy(), wgrad.numpy(), rtol=5e-4, atol=1e-6 * wgrad.abs().max().item() + 1e-9)


@pytest.mark.parametrize("S,D,B", [(256, 50, 3), (32, 37, 4), (96, 130, 2)])
def test_logprob_reverse_prob_matrix_core_path_matches_generic(S, D, B):
    """ctdd_logprob_rp_mfma / ctdd_logprob_rp_bwd_mfma (S x S contractions on v_mfma_f32_32x32x2_f32) against the ORACLE
    (oracle.ctmc_ops.logprob_with_logits, model_utils.py:30-60; d/dlogits by autograd through it in float64) and against the
    generic row kernels (fp32 FMA chains) on the same inputs: forward log-probabilities, backward d/dlogits with the
    cross-entropy term."""
    from ctdd import native
    from oracle import ctmc_ops as ops
    g = torch.Generator().manual_seed(S + D)
    logits = (torch.randn(B, D, S, generator=g) * 2).cuda()
    x = torch.randint(0, S, (B, D), generator=g).to(torch.int32).cuda()
    x0 = torch.randint(0, S, (B, D), generator=g).to(torch.int32).cuda()
    q = torch.softmax(torch.randn(B, S, S, generator=g) * 3, dim=-1).cuda().contiguous()        # row-stochastic tables
    qT = q.transpose(1, 2).contiguous()
    tidx = torch.arange(B, dtype=torch.int32, device="cuda")
    ll_ref, llx_ref = native.logprob(logits, x, q, "reverse_prob", tidx)
    ll, llx = native.logprob(logits, x, q, "reverse_prob", tidx, qt0T=qT)
    np.testing.assert_allclose(ll.cpu().numpy(), ll_ref.cpu().numpy(), rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(llx.cpu().numpy(), llx_ref.cpu().numpy(), rtol=2e-5, atol=2e-6)
    # the independent reference: the oracle's restatement on the CPU (fp32 for the values, float64 autograd for the gradient)
    ll_o, llx_o = ops.logprob_with_logits("reverse_prob", logits.cpu(), x.cpu(), q.cpu())
    np.testing.assert_allclose(ll.cpu().numpy(), ll_o.numpy(), rtol=2e-5, atol=5e-6)
    np.testing.assert_allclose(llx.cpu().numpy(), llx_o.numpy(), rtol=2e-5, atol=5e-6)
    dll = torch.randn(B, D, S, generator=g).cuda()
    for x0_, w in ((None, 0.0), (x0, 0.37)):
        g_ref, ce_ref = native.logprob_bwd("reverse_prob", logits, q, qT, dll, x0_, w)
        g_new, ce_new = native.logprob_bwd("reverse_prob", logits, q, qT, dll, x0_, w, ll_all=ll_ref)
        scale = float(g_ref.abs().max())
        assert float((g_new - g_ref).abs().max()) < 2e-5 * scale
        np.testing.assert_allclose(float(ce_new), float(ce_ref), rtol=1e-5, atol=1e-6)
        l64 = logits.cpu().double().requires_grad_(True)
        ll64, _ = ops.logprob_with_logits("reverse_prob", l64, x.cpu(), q.cpu().double())
        obj = (ll64 * dll.cpu().double()).sum()
        ce64 = torch.zeros((), dtype=torch.float64)
        if x0_ is not None:                                   # + w * sum_{b,d} -log_softmax(logits)[x0]
            ce64 = -torch.log_softmax(l64, -1).gather(-1, x0_.cpu().long().unsqueeze(-1)).sum()
            obj = obj + w * ce64
        g64, = torch.autograd.grad(obj, l64)
        assert float((g_new.cpu().double() - g64).abs().max()) < 5e-5 * float(g64.abs().max())
        if x0_ is not None:
            np.testing.assert_allclose(float(ce_new), w * float(ce64.detach()), rtol=2e-5)     # (the kernels return the weighted term)


@pytest.mark.parametrize("S,D,B", [(16, 12, 5), (256, 9, 3), (37, 20, 4)])
def test_ctelbo_term_weights_compose_the_two_pass_objective(S, D, B):
    """ctdd_ctelbo_loss_terms: value and d/dlogits of (reg + CE at model(x_t), reg_x = x_t) + (signal at model(x~)) against the
    differentiable restatement of losses.py:150-278 in fp64 (the oracle's two-pass golden case covers S = 16 only), and the
    one-pass entry = the terms entry with both weights equal."""
    import lib.losses.losses as L
    from ctdd import native
    from ctdd.process import DeviceForwardProcess
    gen = torch.Generator().manual_seed(S * 100 + D)
    # (the S = 256 Gaussian tables on 16 or 37 states put q(x0 -> x_t) below fp32 resolution for random pairs: uniform rates there)
    proc = DeviceForwardProcess("gaussian", S, "cuda", **GAUSS) if S == 256 else DeviceForwardProcess("uniform", S, "cuda", rate_const=1.7, t_func="sqrt_cos")
    ts = (torch.rand(B, generator=gen) * 0.9 + 0.05).cuda()
    qt0, qT, rate, _ = proc.tables(ts, want_qt0=True, want_qt0T=True, want_rate=True)
    x0 = torch.randint(0, S, (B, D), generator=gen).cuda()
    x_t = torch.randint(0, S, (B, D), generator=gen).cuda()
    x_tilde = x_t.clone()
    x_tilde[:, 1] = (x_tilde[:, 1] + 1) % S
    la = torch.randn(B, D, S, generator=gen).cuda().requires_grad_()
    lb = torch.randn(B, D, S, generator=gen).cuda().requires_grad_()
    w, nllw, eps = 0.7, 0.3, 1e-9
    ref = w * L._ct_elbo_terms(la.double(), lb.double(), x0, x_t, x_tilde, qt0.double(), rate.double(), eps) + \
        nllw * torch.nn.functional.cross_entropy(la.double().permute(0, 2, 1), x0)
    ga, gb = torch.autograd.grad(ref, (la, lb))
    va, da = native.ctelbo_loss(la.detach(), x0.int(), x_t.int(), qt0, qT, rate, eps, 0.0, nllw / (B * D), reg_scale=w)
    vb, db = native.ctelbo_loss(lb.detach(), x0.int(), x_tilde.int(), qt0, qT, rate, eps, w, 0.0, reg_scale=0.0)
    np.testing.assert_allclose((va + vb).item(), ref.item(), rtol=2e-5)
    scale = max(ga.abs().max().item(), gb.abs().max().item())       # (with uniform rates the signal term's gradient vanishes identically)
    for got, want in ((da, ga), (db, gb)):
        # fp32 softmax backward p (dp - sum p dp): dp is constant (and ~1e2 x the result) where the gradient vanishes -> 1e-3 of scale
        assert (got - want.float()).abs().max().item() <= 1e-3 * scale, ((got - want.float()).abs().max().item(), scale)
    v1, d1 = native.ctelbo_loss(lb.detach(), x0.int(), x_tilde.int(), qt0, qT, rate, eps, w, nllw / (B * D))
    v2, d2 = native.ctelbo_loss(lb.detach(), x0.int(), x_tilde.int(), qt0, qT, rate, eps, w, nllw / (B * D), reg_scale=w)
    np.testing.assert_allclose(v1.item(), v2.item(), rtol=1e-6)
    assert (d1 - d2).abs().max().item() <= 1e-6 * d1.abs().max().item()
