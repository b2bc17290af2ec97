"""GPU parity tests proper: every C-ABI entry point of libctdd.so against the CPU oracle, on the
golden fixtures frozen from the reference and on seeded random inputs.  Integer outputs are
compared exactly (bit-exact given the same tables and noise); floating-point outputs within the
tolerance written at each assert."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ctmc_ops as ops
from oracle import philox as oph
from oracle.forward_process import ForwardProcess

T = torch.from_numpy


@pytest.fixture(scope="module")
def nat():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from ctdd import native
    native.load()
    return native


def dev(t, dtype=None):
    t = t if isinstance(t, torch.Tensor) else torch.from_numpy(np.asarray(t))
    if dtype is not None:
        t = t.to(dtype)
    return t.contiguous().cuda()


def gauss(S):
    return ForwardProcess("gaussian", S, rate_sigma=6.0, Q_sigma=512.0, time_exp=100.0, time_base=3.0)


def univar(S, tf="sqrt_cos"):
    return ForwardProcess("univar", S, rate_const=1.7, t_func=tf)


def make_process(kind, S, tf="sqrt_cos"):
    return {"gaussian": lambda: gauss(S), "univar": lambda: univar(S, tf),
            "uniform": lambda: ForwardProcess("uniform", S, rate_const=1.7),
            "birthdeath": lambda: ForwardProcess("birthdeath", S, sigma_min=1.0, sigma_max=100.0)}[kind]()


def tables(nat, proc, t, **want):
    right = proc.inv_eigvecs
    integral = proc.integral(t)
    if proc.kind == "univar":          # transition(t) = transit_between(0, t)  (forward_model.py:202-204)
        integral = integral - proc.integral(torch.zeros_like(t))
    return nat.rate_table(dev(proc.eigvecs), dev(right), dev(proc.eigvals), dev(proc.base_rate),
                          dev(integral), dev(proc.beta(t)), proc.S, proc.kind != "uniform", **want)


# ------------------------------------------------------------------ RNG
def test_philox_bit_exact(nat):
    u = nat.philox_uniform(0x123456789ABCDEF, 77, 1000, 3, "cuda").cpu().numpy()
    ref = oph.row_uniforms(np.arange(1000), 77, 0x123456789ABCDEF, 3)
    assert np.array_equal(u, ref)
    assert u.min() > 0 and u.max() < 1


# ------------------------------------------------------------------ K1
@pytest.mark.parametrize("kind,S", [("gaussian", 256), ("gaussian", 8), ("gaussian", 100), ("univar", 3),
                                    ("univar", 2), ("uniform", 3), ("birthdeath", 8), ("birthdeath", 32)])
def test_rate_table(nat, kind, S):
    proc = make_process(kind, S)
    t = torch.tensor([0.01, 0.25, 0.5, 0.75, 0.99999])
    q, qT, r, pn = tables(nat, proc, t, want_qt0=True, want_qt0T=True, want_rate=True, want_noise_probs=True)
    q_ref, r_ref = proc.transition(t), proc.rate(t)
    # fp32 V diag W products in a different summation order: the reference's own fp32 result is
    # ~3e-7 absolute from the exact expm (measured), so compare at atol 2e-6 + rtol 1e-5.
    np.testing.assert_allclose(q.cpu().numpy(), q_ref.numpy(), rtol=1e-5, atol=2e-6)
    assert torch.equal(qT.cpu(), q.cpu().transpose(1, 2))
    np.testing.assert_allclose(r.cpu().numpy(), r_ref.numpy(), rtol=1e-6)
    x0 = torch.arange(S).view(1, S).repeat(t.numel(), 1)
    probs_ref = ops.noise_probs_rows(q.cpu(), x0).view(t.numel(), S, S)
    np.testing.assert_allclose(pn.cpu().numpy(), probs_ref.numpy(), rtol=2e-5, atol=1e-9)
    np.testing.assert_allclose(pn.sum(-1).cpu().numpy(), 1.0, rtol=1e-5)


# ------------------------------------------------------------------ K2 / K3
@pytest.mark.parametrize("tag,kind,S,tf", [("g16", "gaussian", 16, None), ("g256", "gaussian", 256, None),
                                           ("v3", "univar", 3, "sqrt_cos"), ("v2", "univar", 2, "log_sqr")])
def test_noising_golden_bit_exact(nat, golden, tag, kind, S, tf):
    g = golden("noising")
    proc = make_process(kind, S, tf)
    x0, ts = T(g[f"{tag}_x0"]), T(g[f"{tag}_ts"])
    B, D = x0.shape
    qt0 = proc.transition(ts)
    allx = torch.arange(S).view(1, S).repeat(B, 1)
    probs = ops.noise_probs_rows(qt0, allx).view(B, S, S)      # oracle table -> kernel: exact given (probs,E)
    xt = nat.noise_categorical(dev(probs), dev(x0, torch.int32), E=dev(g[f"{tag}_E_xt"]))
    assert torch.equal(xt.cpu().long(), T(g[f"{tag}_x_t"]).long())
    rate = proc.rate(ts)
    dims, newval, xtil = nat.xtilde_sample(dev(rate), xt, E_dim=dev(g[f"{tag}_E_dim"]), E_val=dev(g[f"{tag}_E_val"]))
    assert torch.equal(xtil.cpu().long(), T(g[f"{tag}_x_tilde"]).long())


def test_noising_random_large_bit_exact(nat):
    S, B, D = 256, 8, 784
    proc = gauss(S)
    g = torch.Generator().manual_seed(5)
    ts = torch.rand(B, generator=g) * 0.99 + 0.01
    x0 = torch.randint(0, S, (B, D), generator=g)
    E = torch.empty(B * D, S).exponential_(1, generator=g)
    qt0 = proc.transition(ts)
    ref = ops.noise_xt(qt0, x0, E)
    allx = torch.arange(S).view(1, S).repeat(B, 1)
    probs = ops.noise_probs_rows(qt0, allx).view(B, S, S)
    xt = nat.noise_categorical(dev(probs), dev(x0, torch.int32), E=dev(E))
    assert torch.equal(xt.cpu().long(), ref)
    # device-built probability table (K1) + same E: indices may differ only at fp near-ties
    _, _, _, pn = tables(nat, proc, ts, want_qt0=False, want_noise_probs=True)
    xt2 = nat.noise_categorical(pn, dev(x0, torch.int32), E=dev(E))
    assert (xt2.cpu().long() != ref).float().mean().item() < 2e-4


def test_noising_philox_replay_and_distribution(nat):
    S, B, D = 16, 64, 512
    proc = gauss(S)
    ts = torch.full((B,), 0.3)
    x0 = torch.full((B, D), 5, dtype=torch.int64)
    qt0 = proc.transition(ts)
    probs = ops.noise_probs_rows(qt0, torch.arange(S).view(1, S).repeat(B, 1)).view(B, S, S)
    xt = nat.noise_categorical(dev(probs), dev(x0, torch.int32), seed=11, offset=3).cpu().long()
    rows = np.arange(B * D, dtype=np.uint64)
    lo, hi = (rows & 0xFFFFFFFF).astype(np.uint32), (rows >> np.uint64(32)).astype(np.uint32)
    E = np.stack([oph.exp1(oph.philox4x32_10(lo, hi, np.uint32(3), np.uint32(s), 11, 0)[0]) for s in range(S)], -1)
    ref = ops.exp_race_argmax(probs[0, 5].view(1, S), T(E)).view(B, D)
    assert (xt != ref).float().mean().item() < 1e-3       # logf ulp differences only
    counts = np.bincount(xt.numpy().ravel(), minlength=S).astype(np.float64)
    p = probs[0, 5].double().numpy()
    n = counts.sum()
    m = p > 0
    chi2 = (((counts - n * p) ** 2)[m] / (n * p)[m]).sum()
    assert chi2 < 60.0 and counts[~m].sum() == 0     # dof <= 15: P(chi2>60) ~ 1e-7


# ------------------------------------------------------------------ A6 / K4 / K5
@pytest.mark.parametrize("tag,kind,S", [("g256", "gaussian", 256), ("g16", "gaussian", 16), ("v3", "univar", 3),
                                        ("u3", "uniform", 3)])
def test_logprob_and_rates_golden(nat, golden, tag, kind, S):
    g = golden("rates")
    proc = make_process(kind, S)
    logits, x, t = T(g[f"{tag}_logits"]), T(g[f"{tag}_x"]), T(g[f"{tag}_t"])
    N = x.shape[0]
    dl, dx = dev(logits), dev(x, torch.int32)
    tidx = torch.arange(N, dtype=torch.int32).cuda()
    qt0, rate = dev(proc.transition(t)), dev(proc.rate(t))
    for lt in ("direct", "reverse_prob", "reverse_logscale"):
        ll_all, ll_xt = nat.logprob(dl, dx, qt0, lt, tidx)
        np.testing.assert_allclose(ll_all.cpu().numpy(), g[f"{tag}_{lt}_ll_all"], rtol=1e-5, atol=2e-6)
        np.testing.assert_allclose(ll_xt.cpu().numpy(), g[f"{tag}_{lt}_ll_xt"], rtol=1e-5, atol=2e-6)
        rr, ratio = nat.reverse_rates(nat.BRANCH_CRM, lt, dl, dx, qt0, rate, 1e-9, tidx)
        np.testing.assert_allclose(rr.cpu().numpy(), g[f"{tag}_{lt}_crm_rates"], rtol=1e-4, atol=1e-30)
        np.testing.assert_allclose(ratio.cpu().numpy(), g[f"{tag}_{lt}_crm_ratio"], rtol=1e-4, atol=1e-30)
    for nm, tt, ti in (("shared", torch.full((1,), 0.37), None), ("perrow", t, tidx)):
        rr, ratio = nat.reverse_rates(nat.BRANCH_CTELBO, "direct", dl, dx, dev(proc.transition(tt)),
                                      dev(proc.rate(tt)), 1e-9, ti)
        np.testing.assert_allclose(rr.cpu().numpy(), g[f"{tag}_ctelbo_{nm}_rates"], rtol=1e-4, atol=1e-30)
        np.testing.assert_allclose(ratio.cpu().numpy(), g[f"{tag}_ctelbo_{nm}_ratio"], rtol=1e-4, atol=1e-30)


@pytest.mark.parametrize("S,N,D", [(2, 7, 33), (3, 5, 225), (7, 3, 10), (64, 3, 17), (100, 2, 9), (256, 2, 50),
                                   (300, 2, 5), (1000, 1, 3)])
def test_rates_random_shapes(nat, S, N, D):
    proc = gauss(S) if S >= 8 else univar(S)
    g = torch.Generator().manual_seed(S)
    logits = torch.randn(N, D, S, generator=g) * 3
    x = torch.randint(0, S, (N, D), generator=g)
    t = torch.rand(N, generator=g) * 0.9 + 0.05
    qt0, rate = proc.transition(t), proc.rate(t)
    tidx = torch.arange(N, dtype=torch.int32).cuda()
    rr_ref, ratio_ref = ops.reverse_rates_ctelbo(logits, x, qt0, rate, 1e-9)
    rr, ratio = nat.reverse_rates(nat.BRANCH_CTELBO, "direct", dev(logits), dev(x, torch.int32), dev(qt0), dev(rate), 1e-9, tidx)
    np.testing.assert_allclose(rr.cpu().numpy(), rr_ref.numpy(), rtol=1e-4, atol=1e-30)
    np.testing.assert_allclose(ratio.cpu().numpy(), ratio_ref.numpy(), rtol=1e-4, atol=1e-30)
    for lt in ("direct", "reverse_prob", "reverse_logscale"):
        rr_ref, _ = ops.reverse_rates_crm(lt, logits, x, qt0, rate)
        rr, _ = nat.reverse_rates(nat.BRANCH_CRM, lt, dev(logits), dev(x, torch.int32), dev(qt0), dev(rate), 0.0, tidx)
        np.testing.assert_allclose(rr.cpu().numpy(), rr_ref.numpy(), rtol=2e-4, atol=1e-30)


# ------------------------------------------------------------------ K6
@pytest.mark.parametrize("ordinal", [True, False])
def test_tauleap_apply_exact(nat, ordinal):
    S, N, D = 16, 9, 40
    g = torch.Generator().manual_seed(3)
    x = torch.randint(0, S, (N, D), generator=g)
    xb = torch.randint(0, S, (N, D), generator=g)
    jn = torch.poisson(torch.rand(N, D, S, generator=g) * 0.08, generator=g)
    ref = ops.tauleap_apply(x, jn, ordinal)
    out = nat.tauleap_apply(dev(x, torch.int32), dev(jn), ordinal)
    assert torch.equal(out.cpu().long(), ref)
    ref = ops.tauleap_apply(x, jn, ordinal, base=xb)
    out = nat.tauleap_apply(dev(x, torch.int32), dev(jn), ordinal, x_base=dev(xb, torch.int32))
    assert torch.equal(out.cpu().long(), ref)


@pytest.mark.parametrize("S,ordinal", [(256, True), (16, False), (3, False), (2, True), (100, True)])
def test_tauleap_draw_replay(nat, S, ordinal):
    N, D = 6, 300
    g = torch.Generator().manual_seed(S + 1)
    rates = torch.rand(N, D, S, generator=g) ** 4 * (40.0 / S)
    rates[0, :10] *= 200.0 / rates[0, :10].sum(-1, keepdim=True)      # a few dense rows (lambda > 12)
    x = torch.randint(0, S, (N, D), generator=g)
    h = 0.05
    changed = torch.zeros(1, dtype=torch.int32).cuda()
    out = nat.tauleap_draw(dev(rates), dev(x, torch.int32), h, ordinal, seed=99, offset=5, changed=changed).cpu().long()
    ref, decided = oph.tauleap_draw_replay(rates.numpy(), x.numpy(), h, ordinal, 99, 5)
    dec = T(decided)
    assert dec.float().mean() > 0.9
    assert torch.equal(out[dec], T(ref)[dec])
    assert int(changed.item()) == int((out != x).sum())
    assert out.min() >= 0 and out.max() < S


@pytest.mark.parametrize("S", [256, 16])
def test_tauleap_draw_heavy_rates(nat, S):
    """Rates as a random-init logistic head produces them (h r up to ~1e6 per destination): the heavy sub-block rule
    (independent Poisson per destination, PTRS above 12) replays on the CPU, saturates the state clamp, and its
    per-destination counts have Poisson mean and variance."""
    N, D = 3, 200
    g = torch.Generator().manual_seed(S)
    rates = torch.rand(N, D, S, generator=g) ** 6 * 3e4
    rates[1] *= 1e3                                             # up to 3e7 -> h r ~ 3e5
    rates[2, :, : S // 2] = 0.0
    x = torch.randint(0, S, (N, D), generator=g)
    h = 0.01
    out = nat.tauleap_draw(dev(rates), dev(x, torch.int32), h, True, seed=5, offset=9).cpu().long()
    ref, decided = oph.tauleap_draw_replay(rates.numpy(), x.numpy(), h, True, 5, 9)
    dec = T(decided)
    assert dec.float().mean() > 0.95
    assert torch.equal(out[dec], T(ref)[dec])
    assert out.min() >= 0 and out.max() <= S - 1
    below = x[2] < S // 2                                       # all rate mass above the state: saturates at S - 1
    assert below.any() and (out[2][below] == S - 1).float().mean() > 0.95
    # one heavy destination next to the state: x_new - x = k ~ Poisson(h r) while the clamp is inactive
    Sb, n = 1024, 20000
    lam = 300.0
    r = torch.zeros(1, n, Sb)
    r[..., 1] = lam / h                                         # (row total 300 > 64: dense regime, heavy sub-block 0)
    x0 = torch.zeros(1, n, dtype=torch.int64)
    k = nat.tauleap_draw(dev(r), dev(x0, torch.int32), h, True, seed=3, offset=1).cpu().double().view(-1)
    assert abs(k.mean().item() - lam) < 6 * np.sqrt(lam / n) and abs(k.var().item() - lam) < 0.06 * lam


def test_tauleap_draw_poisson_marginals(nat):
    """Superposition rule == independent Poisson per destination: check the jump-count marginals."""
    S, N, D = 64, 64, 4096                   # S large enough that the clamp at S-1 never binds
    r8 = torch.tensor([0.0, 3.0, 0.5, 0.0, 1.0, 0.0, 0.25, 0.0])
    rates = torch.cat([r8, torch.zeros(S - 8)]).view(1, 1, S).repeat(N, D, 1)
    x = torch.zeros(N, D, dtype=torch.int64)
    h = 0.1
    out = nat.tauleap_draw(dev(rates), dev(x, torch.int32), h, False, seed=1, offset=0).cpu().numpy().ravel()
    n = out.size
    lam = rates[0, 0].double().numpy() * h
    L = lam.sum()
    # non-ordinal: exactly one jump in total -> moved to s with prob lam_s * exp(-L); else stays at 0
    p = lam * np.exp(-L)
    p[0] = 1.0 - p[1:].sum()
    counts = np.bincount(out, minlength=S).astype(np.float64)
    m = p > 0
    chi2 = (((counts - n * p) ** 2)[m] / (n * p)[m]).sum()
    assert chi2 < 50.0 and counts[~m].sum() == 0
    # ordinal: E[x_new] = sum_s lam_s * s (clamp inactive here)
    out = nat.tauleap_draw(dev(rates), dev(x, torch.int32), h, True, seed=2, offset=0).cpu().double()
    mean_ref = (lam * np.arange(S)).sum()
    var_ref = (lam * np.arange(S) ** 2).sum()
    assert abs(out.mean().item() - mean_ref) < 6 * np.sqrt(var_ref / n)
    assert abs(out.var().item() - var_ref) < 0.05 * var_ref


@pytest.mark.parametrize("kind,S,branch,lt,flags", [("gaussian", 256, 0, "direct", 1), ("gaussian", 16, 0, "direct", 0),
                                                    ("univar", 3, 1, "reverse_prob", 0), ("univar", 2, 1, "direct", 0),
                                                    ("gaussian", 16, 0, "direct", 3), ("univar", 3, 1, "reverse_logscale", 2)])
def test_tauleap_step_fused_matches_unfused(nat, kind, S, branch, lt, flags):
    """Fused step == reverse_rates (checked against the oracle above) followed by the draw kernel."""
    N, D = 4, 200
    proc = make_process(kind, S)
    g = torch.Generator().manual_seed(17)
    logits = torch.randn(N, D, S, generator=g) * 2
    x = torch.randint(0, S, (N, D), generator=g)
    t = torch.full((1,), 0.4)
    qt0, rate = proc.transition(t), proc.rate(t)
    beta = float(proc.beta(t)[0])
    if branch == 0:
        rr0, _ = ops.reverse_rates_ctelbo(logits, x, qt0.repeat(N, 1, 1), rate.repeat(N, 1, 1), 1e-9)
    else:
        rr0, _ = ops.reverse_rates_crm(lt, logits, x, qt0.repeat(N, 1, 1), rate.repeat(N, 1, 1))
    # step size such that the median row expects ~1 jump (most rows on the K~Poisson(total) path)
    h = float(1.0 / ops.zero_own_state(rr0, x).sum(-1).median().clamp_min(1e-6))
    dl, dx = dev(logits), dev(x, torch.int32)
    out = nat.tauleap_step(branch, lt, dl, dx, dev(qt0), dev(proc.base_rate), beta, 1e-9, h, flags, 7, 21)
    rr, _ = nat.reverse_rates(branch, lt, dl, dx, dev(qt0), dev(rate), 1e-9)
    if flags & 2:
        rr = rr + dev(ops.transpose_forward_rates(rate.repeat(N, 1, 1), x))
    out2 = nat.tauleap_draw(rr.contiguous(), dx, h, bool(flags & 1), 7, 21)
    assert (out != out2).float().mean().item() < 2e-3
    # and against the oracle's rates + CPU replay of the draw rule
    if branch == 0:
        rr_ref, _ = ops.reverse_rates_ctelbo(logits, x, qt0.repeat(N, 1, 1), rate.repeat(N, 1, 1), 1e-9)
    else:
        rr_ref, _ = ops.reverse_rates_crm(lt, logits, x, qt0.repeat(N, 1, 1), rate.repeat(N, 1, 1))
    if flags & 2:
        rr_ref = rr_ref + ops.transpose_forward_rates(rate.repeat(N, 1, 1), x)
    ref, decided = oph.tauleap_draw_replay(rr_ref.numpy(), x.numpy(), h, bool(flags & 1), 7, 21)
    dec = T(decided)
    assert dec.float().mean() > 0.6
    assert (out.cpu().long()[dec] != T(ref)[dec]).float().mean().item() < 2e-3
    assert (out.cpu().long() != x).float().mean() > 0.05


# ------------------------------------------------------------------ K7 / K8 / K10 / A7
@pytest.mark.parametrize("kind,S,branch,lt", [("gaussian", 256, 0, "direct"), ("gaussian", 16, 0, "direct"),
                                              ("univar", 3, 1, "reverse_prob"), ("univar", 2, 1, "reverse_logscale")])
def test_lbjf_step(nat, kind, S, branch, lt):
    N, D = 3, 150
    proc = make_process(kind, S)
    g = torch.Generator().manual_seed(23)
    logits = torch.randn(N, D, S, generator=g) * 2
    x = torch.randint(0, S, (N, D), generator=g)
    t = torch.full((1,), 0.6)
    qt0, rate = proc.transition(t).repeat(N, 1, 1), proc.rate(t).repeat(N, 1, 1)
    h = 0.01
    E = torch.empty(N * D, S).exponential_(1, generator=g)
    if branch == 0:
        rr, _ = ops.reverse_rates_ctelbo(logits, x, qt0, rate, 1e-9)
    else:
        rr, _ = ops.reverse_rates_crm(lt, logits, x, qt0, rate)
    P = ops.lbjf_posterior(rr, x, h)
    probs_ref = ops.categorical_probs_from_logits(torch.log(P + 1e-35).view(-1, S))
    vals = probs_ref / E
    ref = torch.argmax(vals, -1).view(N, D)
    top2 = torch.topk(vals, 2, dim=-1).values
    decided = ((top2[:, 0] - top2[:, 1]) > 1e-4 * top2[:, 0]).view(N, D)
    out, probs = nat.lbjf_step(branch, lt, dev(logits), dev(x, torch.int32), dev(qt0[:1]), dev(proc.base_rate),
                               float(proc.beta(t)[0]), 1e-9, h, E=dev(E), want_probs=True)
    np.testing.assert_allclose(probs.cpu().numpy().reshape(-1, S), probs_ref.numpy(), rtol=2e-4, atol=1e-30)
    assert decided.float().mean() > 0.99
    assert torch.equal(out.cpu().long()[decided], ref[decided])


@pytest.mark.parametrize("kind,S", [("univar", 3), ("uniform", 3), ("gaussian", 16), ("gaussian", 256)])
def test_exact_step(nat, kind, S):
    """ctdd_exact_step (ExactSampling, sampling.py:990-1061) against the oracle's log-space formula
    logsumexp_x0(log p0t[x0] + log(q_{t-h|0}[x0, s] q_{t|t-h}[s, x_t])) and the same exponential noise."""
    import torch.nn.functional as F
    N, D = 3, 70
    proc = make_process(kind, S)
    g = torch.Generator().manual_seed(31)
    logits = torch.randn(N, D, S, generator=g) * 2
    x = torch.randint(0, S, (N, D), generator=g)
    t_hi, t_lo = torch.full((1,), 0.6), torch.full((1,), 0.55)
    q_lo = proc.transition(t_lo)[0]
    q_step = proc.transit_between(t_lo, t_hi)[0]
    E = torch.empty(N * D, S).exponential_(1, generator=g)
    log_p0t = F.log_softmax(logits.double(), dim=2)
    qq = q_lo.double().view(1, 1, S, S) * q_step.double().t()[x].unsqueeze(-2)                   # [n, d, x0, s]
    log_prob = torch.logsumexp(log_p0t.unsqueeze(-1) + torch.log(qq), dim=-2).view(-1, S)
    probs_ref = ops.categorical_probs_from_logits(log_prob.float())
    vals = probs_ref / E
    ref = torch.argmax(vals, -1).view(N, D)
    top2 = torch.topk(vals, 2, dim=-1).values
    decided = ((top2[:, 0] - top2[:, 1]) > 1e-4 * top2[:, 0]).view(N, D)
    cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    out, probs = nat.exact_step(dev(logits), dev(x, torch.int32), dev(q_lo), dev(q_step), E=dev(E), want_probs=True, changed=cnt)
    np.testing.assert_allclose(probs.cpu().numpy().reshape(-1, S), probs_ref.numpy(), rtol=2e-4, atol=1e-30)
    assert decided.float().mean() > 0.99
    assert torch.equal(out.cpu().long()[decided], ref[decided])
    assert int(cnt.item()) == int((out.cpu().long() != x).sum())
    # Philox noise: replay of the (seed, offset, row, s) stream
    out2 = nat.exact_step(dev(logits), dev(x, torch.int32), dev(q_lo), dev(q_step), seed=7, offset=2)
    rows = np.arange(N * D, dtype=np.uint64)
    lo, hi = (rows & 0xFFFFFFFF).astype(np.uint32), (rows >> np.uint64(32)).astype(np.uint32)
    Eph = np.stack([oph.exp1(oph.philox4x32_10(lo, hi, np.uint32(2), np.uint32(s_), 7, 0)[0]) for s_ in range(S)], -1)
    v2 = probs_ref / T(Eph)
    t2 = torch.topk(v2, 2, dim=-1).values
    dec2 = ((t2[:, 0] - t2[:, 1]) > 1e-4 * t2[:, 0]).view(N, D)
    assert torch.equal(out2.cpu().long()[dec2], torch.argmax(v2, -1).view(N, D)[dec2])


@pytest.mark.parametrize("kind,S,branch,lt", [("univar", 3, 1, "reverse_prob"), ("univar", 2, 1, "direct"),
                                              ("gaussian", 16, 0, "direct")])
def test_midpoint_predict(nat, kind, S, branch, lt):
    N, D = 5, 120
    proc = make_process(kind, S)
    g = torch.Generator().manual_seed(29)
    logits = torch.randn(N, D, S, generator=g) * 3
    x = torch.randint(0, S, (N, D), generator=g)
    t = torch.full((1,), 0.8)
    qt0, rate = proc.transition(t).repeat(N, 1, 1), proc.rate(t).repeat(N, 1, 1)
    if branch == 0:
        rr, _ = ops.reverse_rates_ctelbo(logits, x, qt0, rate, 1e-9)
    else:
        rr, _ = ops.reverse_rates_crm(lt, logits, x, qt0, rate)
    rr = ops.zero_own_state(rr, x)
    diff = torch.arange(S, dtype=torch.float32).view(1, 1, S) - x.float().unsqueeze(-1)
    h = float(2.4 / torch.sum(rr * diff, -1).abs().median().clamp_min(1e-6))     # median drift 1.2 states
    drift = 0.5 * h * torch.sum(rr * diff, -1)
    decided = (drift - torch.floor(drift) - 0.5).abs() > 1e-4 * drift.abs() + 1e-5
    ref = ops.midpoint_predict(x, rr, h, S)
    out = nat.midpoint_predict(branch, lt, dev(logits), dev(x, torch.int32), dev(qt0[:1]), dev(proc.base_rate),
                               float(proc.beta(t)[0]), 1e-9, h).cpu().long()
    assert decided.float().mean() > 0.9
    assert torch.equal(out[decided], ref[decided])
    assert (out != x).float().mean() > 0.2


def test_argmax_and_initial(nat):
    g = torch.Generator().manual_seed(31)
    logits = torch.randn(4, 33, 256, generator=g)
    logits[0, 0, 7] = logits[0, 0, 200] = 9.0      # tie -> first index
    out = nat.argmax(dev(logits)).cpu().long()
    assert torch.equal(out, torch.argmax(logits, -1)) and out[0, 0] == 7
    # every kernel variant (S = 256 vector path with a ragged row count, lane-per-row for S <= 8, generic), ties included
    for S_, N_, D_ in ((256, 3, 171), (2, 5, 33), (3, 7, 225), (8, 2, 19), (40, 3, 17), (300, 2, 9)):
        lg = torch.randn(N_, D_, S_, generator=g).round(decimals=1)       # one decimal: plenty of ties
        assert torch.equal(nat.argmax(dev(lg)).cpu().long(), torch.argmax(lg, -1)), S_
    S, N, D = 256, 64, 784
    pmf = ops.gaussian_initial_pmf(S, 512.0)
    cdf = torch.from_numpy(np.cumsum(pmf)).float()
    x = nat.initial_samples(N, D, S, "cuda", 5, 0, cdf=dev(cdf)).cpu().numpy().ravel()
    counts = np.bincount(x, minlength=S).astype(np.float64)
    n = counts.sum()
    chi2 = ((counts - n * pmf) ** 2 / (n * pmf)).sum()
    assert chi2 < 400.0           # dof 255: mean 255, sd 22.6
    xu = nat.initial_samples(N, D, 3, "cuda", 6, 0).cpu().numpy().ravel()
    cu = np.bincount(xu, minlength=3) / xu.size
    assert np.abs(cu - 1 / 3).max() < 0.01


def test_errors_are_loud(nat):
    with pytest.raises(nat.CtddError):
        nat.argmax(torch.zeros(1, 1, 4))           # CPU tensor: no fallback
    with pytest.raises(nat.CtddError):
        nat.reverse_rates(5, "direct", dev(torch.zeros(1, 1, 4)), dev(torch.zeros(1, 1), torch.int32),
                          dev(torch.zeros(1, 4, 4)), dev(torch.zeros(1, 4, 4)), 0.0)
