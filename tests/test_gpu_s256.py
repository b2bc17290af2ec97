"""GPU: the S=256 MFMA fast path (csrc/steps_s256.hip) against the oracle and against the generic
exact-fp32 kernel."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ctmc_ops as ops
from oracle import philox as oph
from oracle.forward_process import ForwardProcess

S = 256


@pytest.fixture(scope="module")
def env():
    from ctdd import native
    from ctdd.process import DeviceForwardProcess
    p = dict(rate_sigma=6.0, Q_sigma=512.0, time_exp=100.0, time_base=3.0)
    return native, DeviceForwardProcess("gaussian", S, "cuda", **p), ForwardProcess("gaussian", S, **p)


def _case(N, D, seed, scale):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(N, D, S, generator=g) * scale, torch.randint(0, S, (N, D), generator=g)


@pytest.mark.parametrize("t,scale,N,D", [(0.5, 1.0, 3, 100), (0.05, 4.0, 2, 131), (0.97, 2.0, 1, 784), (0.3, 8.0, 5, 64)])
def test_rates_match_oracle(env, t, scale, N, D):
    """Masked reverse rates from the split-bf16 MFMA path: rtol 1e-4 vs the oracle (the stated
    parity bar for rates), and the measured error vs float64 stays below 3e-5."""
    native, pr, op = env
    logits, x = _case(N, D, 1, scale)
    tt = torch.tensor([t])
    # same table on both sides: the reference zeroes q_{t|0} entries below 1e-8, so an entry that
    # lands on the other side of that threshold in a different summation order changes
    # 1/(q+eps) by an order of magnitude -- a property of the algorithm, tested in K1's own test
    qt0 = op.transition(tt).cuda()
    tabs = native.S256Tables(qt0, pr.base_rate, 1e-9)
    beta = float(pr.beta(tt)[0])
    _, rates = native.tauleap_step_s256(logits.cuda().contiguous(), x.to(torch.int32).cuda(), tabs, 0, beta, 1e-3, 1, 1, 0,
                                        want_rates=True)
    q, r = op.transition(tt).repeat(N, 1, 1), op.rate(tt).repeat(N, 1, 1)
    ref = ops.zero_own_state(ops.reverse_rates_ctelbo(logits, x, q, r, 1e-9)[0], x)
    got = rates.cpu()
    # rates below 1e-25 sit in / next to the fp32 denormal range (forward rates exp(-k^2/36) far
    # from x): the reference's own fp32 value there has no relative accuracy, and such a rate
    # never produces a jump, so they are compared absolutely.
    np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=1e-4, atol=1e-25)
    # the contraction itself (ratio = rates / forward rate) against float64 from the same fp32 table
    p = torch.softmax(logits.double(), -1)
    q64 = qt0[0].cpu().double()
    truth = (p / (q64.t()[x] + 1e-9)) @ q64
    fwd = ops.zero_own_state(r[0].t()[x], x).double()
    ok = fwd > 1e-20
    rel = ((got.double() / fwd.clamp_min(1e-300) - truth).abs() / truth)[ok]
    assert rel.max().item() < 3e-5, rel.max().item()


@pytest.mark.parametrize("flags", [1, 0, 3])
def test_fused_step_matches_generic_and_replay(env, flags):
    native, pr, op = env
    N, D = 4, 300
    logits, x = _case(N, D, 5, 2.0)
    tt = torch.tensor([0.4])
    qt0 = pr.tables(tt, want_qt0=True)[0]
    beta = float(pr.beta(tt)[0])
    q, r = op.transition(tt).repeat(N, 1, 1), op.rate(tt).repeat(N, 1, 1)
    rr = ops.reverse_rates_ctelbo(logits, x, q, r, 1e-9)[0]
    if flags & 2:
        rr = rr + ops.transpose_forward_rates(r, x)
    h = float(1.0 / ops.zero_own_state(rr, x).sum(-1).median())
    tabs = native.S256Tables(qt0, pr.base_rate, 1e-9)
    dl, dx = logits.cuda().contiguous(), x.to(torch.int32).cuda()
    changed = torch.zeros(1, dtype=torch.int32, device="cuda")
    out = native.tauleap_step_s256(dl, dx, tabs, 0, beta, h, flags, 77, 3, changed=changed).cpu().long()
    gen = native.tauleap_step(native.BRANCH_CTELBO, "direct", dl, dx, qt0[0], pr.base_rate, beta, 1e-9, h, flags, 77, 3).cpu().long()
    assert (out != gen).float().mean().item() < 3e-3          # same Philox stream; only fp near-ties differ
    ref, decided = oph.tauleap_draw_replay(rr.numpy(), x.numpy(), h, bool(flags & 1), 77, 3)
    dec = torch.from_numpy(decided)
    assert dec.float().mean() > 0.6
    assert (out[dec] != torch.from_numpy(ref)[dec]).float().mean().item() < 3e-3
    assert int(changed.item()) == int((out != x).sum())
    assert (out != x).float().mean() > 0.05


def test_midpoint_base_and_ragged_tail(env):
    """x_base semantics (rates at x', move added to x) and a row count that is not a tile multiple."""
    native, pr, op = env
    N, D = 1, 131
    logits, x = _case(N, D, 9, 2.0)
    xb = (x + torch.randint(-3, 4, x.shape)).clamp(0, S - 1)
    tt = torch.tensor([0.6])
    qt0 = pr.tables(tt, want_qt0=True)[0]
    beta = float(pr.beta(tt)[0])
    tabs = native.S256Tables(qt0, pr.base_rate, 1e-9)
    q, r = op.transition(tt).repeat(N, 1, 1), op.rate(tt).repeat(N, 1, 1)
    rr = ops.reverse_rates_ctelbo(logits, xb, q, r, 1e-9)[0]
    h = float(1.0 / ops.zero_own_state(rr, xb).sum(-1).median())
    out = native.tauleap_step_s256(logits.cuda().contiguous(), x.to(torch.int32).cuda(), tabs, 0, beta, h, 0, 5, 1,
                                   x_base=xb.to(torch.int32).cuda()).cpu().long()
    ref, decided = oph.tauleap_draw_replay(rr.numpy(), x.numpy(), h, False, 5, 1, x_base=xb.numpy())
    dec = torch.from_numpy(decided)
    assert (out[dec] != torch.from_numpy(ref)[dec]).float().mean().item() < 1e-2 and dec.float().mean() > 0.5


def test_full_size_properties(env):
    """BASELINE size (N=256, D=784, S=256): size-independent properties of the fused step -- determinism under a fixed
    Philox key/offset, a different stream for a different offset, no move at h = 0, states stay in range, and the
    change counter equals the number of moved dimensions."""
    native, pr, _ = env
    N, D = 256, 784
    g = torch.Generator(device="cuda").manual_seed(3)
    logits = torch.randn((N, D, S), generator=g, device="cuda") * 2.0
    x = torch.randint(0, S, (N, D), generator=g, device="cuda", dtype=torch.int32)
    tt = torch.tensor([0.4])
    tabs = native.S256Tables(pr.transition(tt), pr.base_rate, 1e-9)
    beta = float(pr.beta(tt)[0])
    cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    a = native.tauleap_step_s256(logits, x, tabs, 0, beta, 2e-3, native.STEP_ORDINAL, 7, 11, changed=cnt)
    b = native.tauleap_step_s256(logits, x, tabs, 0, beta, 2e-3, native.STEP_ORDINAL, 7, 11)
    c = native.tauleap_step_s256(logits, x, tabs, 0, beta, 2e-3, native.STEP_ORDINAL, 7, 12)
    z = native.tauleap_step_s256(logits, x, tabs, 0, beta, 0.0, native.STEP_ORDINAL, 7, 11)
    assert torch.equal(a, b) and not torch.equal(a, c) and torch.equal(z, x)
    assert int(a.min()) >= 0 and int(a.max()) <= S - 1
    moved = int((a != x).sum())
    assert int(cnt.item()) == moved and 0 < moved < N * D
    # non-ordinal: a dimension with two or more jumps stays, so no more dimensions move than in the ordinal run with the same stream
    d = native.tauleap_step_s256(logits, x, tabs, 0, beta, 2e-3, 0, 7, 11)
    assert int((d != x).sum()) <= moved


@pytest.mark.parametrize("t,scale,N,D", [(0.5, 1.0, 3, 100), (0.05, 4.0, 2, 131), (0.97, 2.0, 1, 784)])
def test_crm_reverse_prob_rates_match_oracle(env, t, scale, N, D):
    """CRM branch with logit_type reverse_prob on the same MFMA kernel (STEP_CRM, tables with a unit left scaling):
    rates = exp(ll_all - ll_xt) * beta R[x][s] (sampling.py:61-73), rtol 1e-4 against the oracle."""
    native, pr, op = env
    logits, x = _case(N, D, 11, scale)
    tt = torch.tensor([t])
    qt0 = op.transition(tt).cuda()
    tabs = native.S256Tables(qt0, pr.base_rate, 0.0, crm=True)
    beta = float(pr.beta(tt)[0])
    _, rates = native.tauleap_step_s256(logits.cuda().contiguous(), x.to(torch.int32).cuda(), tabs, 0, beta, 1e-3, 1, 1, 0,
                                        want_rates=True)
    q, r = op.transition(tt).repeat(N, 1, 1), op.rate(tt).repeat(N, 1, 1)
    ref = ops.zero_own_state(ops.reverse_rates_crm("reverse_prob", logits, x, q, r)[0], x)
    np.testing.assert_allclose(rates.cpu().numpy(), ref.numpy(), rtol=1e-4, atol=1e-25)


@pytest.mark.parametrize("flags", [1, 3])
def test_crm_fused_step_matches_generic_and_replay(env, flags):
    native, pr, op = env
    N, D = 4, 300
    logits, x = _case(N, D, 15, 2.0)
    tt = torch.tensor([0.4])
    qt0 = pr.tables(tt, want_qt0=True)[0]
    beta = float(pr.beta(tt)[0])
    q, r = op.transition(tt).repeat(N, 1, 1), op.rate(tt).repeat(N, 1, 1)
    rr = ops.reverse_rates_crm("reverse_prob", logits, x, q, r)[0]
    if flags & 2:
        rr = rr + ops.transpose_forward_rates(r, x)
    h = float(1.0 / ops.zero_own_state(rr, x).sum(-1).median())
    tabs = native.S256Tables(qt0, pr.base_rate, 0.0, crm=True)
    dl, dx = logits.cuda().contiguous(), x.to(torch.int32).cuda()
    changed = torch.zeros(1, dtype=torch.int32, device="cuda")
    out = native.tauleap_step_s256(dl, dx, tabs, 0, beta, h, flags, 77, 3, changed=changed).cpu().long()
    gen = native.tauleap_step(native.BRANCH_CRM, "reverse_prob", dl, dx, qt0[0], pr.base_rate, beta, 1e-9, h, flags, 77, 3).cpu().long()
    assert (out != gen).float().mean().item() < 3e-3          # same Philox stream; only fp near-ties differ
    ref, decided = oph.tauleap_draw_replay(rr.numpy(), x.numpy(), h, bool(flags & 1), 77, 3)
    dec = torch.from_numpy(decided)
    assert dec.float().mean() > 0.6
    assert (out[dec] != torch.from_numpy(ref)[dec]).float().mean().item() < 3e-3
    assert int(changed.item()) == int((out != x).sum())
    assert (out != x).float().mean() > 0.05


@pytest.mark.parametrize("branch", ["ctelbo", "crm"])
def test_lbjf_and_midpoint_on_matrix_core_rates(env, branch):
    """S = 256 Euler (LBJF) step and midpoint predictor with the reverse rates taken from the matrix-core kernel
    (ctdd_lbjf_from_rates / ctdd_midpoint_from_rates) against the generic fp32 kernels on the same Philox stream."""
    native, pr, op = env
    N, D = 3, 211
    logits, x = _case(N, D, 21, 2.0)
    tt = torch.tensor([0.35])
    qt0 = pr.tables(tt, want_qt0=True)[0]
    beta = float(pr.beta(tt)[0])
    crm = branch == "crm"
    tabs = native.S256Tables(qt0, pr.base_rate, 0.0 if crm else 1e-9, crm=crm)
    br, lt = (native.BRANCH_CRM, "reverse_prob") if crm else (native.BRANCH_CTELBO, "direct")
    dl, dx = logits.cuda().contiguous(), x.to(torch.int32).cuda()
    q, r = op.transition(tt).repeat(N, 1, 1), op.rate(tt).repeat(N, 1, 1)
    rr = (ops.reverse_rates_crm("reverse_prob", logits, x, q, r) if crm else ops.reverse_rates_ctelbo(logits, x, q, r, 1e-9))[0]
    h = float(0.5 / ops.zero_own_state(rr, x).sum(-1).median())
    for flags in (0, native.STEP_CORRECTOR):
        _, rates = native.tauleap_step_s256(dl, dx, tabs, 0, beta, h, flags, 9, 4, want_rates=True, want_x=False)
        changed = torch.zeros(1, dtype=torch.int32, device="cuda")
        got, probs = native.lbjf_from_rates(rates, dx, h, None, 9, 4, want_probs=True, changed=changed)
        ref, rprobs = native.lbjf_step(br, lt, dl, dx, qt0[0], pr.base_rate, beta, 1e-9, h, flags, None, 9, 4, want_probs=True)
        # (atol: the own-state entry 1 - h * sum(rates) cancels where h * sum ~ 1; the sums agree to 3e-5 relative)
        np.testing.assert_allclose(probs.cpu().numpy(), rprobs.cpu().numpy(), rtol=2e-4, atol=5e-6)
        assert (got != ref).float().mean().item() < 3e-3            # exponential-race near-ties only
        assert int(changed.item()) == int((got != dx).sum())
    _, rates = native.tauleap_step_s256(dl, dx, tabs, 0, beta, h, 0, 9, 4, want_rates=True, want_x=False)
    xp = native.midpoint_from_rates(rates, dx, 4 * h)
    xr = native.midpoint_predict(br, lt, dl, dx, qt0[0], pr.base_rate, beta, 1e-9, 4 * h)
    diff = (xp != xr)
    assert diff.float().mean().item() < 5e-3 and (xp - xr).abs().max().item() <= 1    # rounding of a drift that sits on .5
    assert (xp != dx).float().mean().item() > 0.05


# ---------------------------------------------------------------- single-product bf16 mode (csrc/steps_s256_b16.hip)
# Tolerance of the mode: 1/(q+eps), w = e^{l-max}/(q+eps) and q_{t|0} are each rounded to bf16 once (unit roundoff 2^-8) and
# every term of the contraction is >= 0, so a rate is within 3 * 2^-8 = 1.2e-2 of the exact one (observed maximum ~8e-3, rms
# ~2e-3); the bound asserted is 3 * 2^-8.
B16_RTOL = 3 * 2.0 ** -8


@pytest.mark.parametrize("t,scale,N,D", [(0.5, 1.0, 3, 100), (0.05, 4.0, 2, 131), (0.97, 2.0, 1, 784), (0.3, 8.0, 5, 64)])
def test_b16_rates_match_oracle(env, t, scale, N, D):
    native, pr, op = env
    logits, x = _case(N, D, 1, scale)
    tt = torch.tensor([t])
    qt0 = op.transition(tt).cuda()
    tabs = native.S256Tables(qt0, pr.base_rate, 1e-9, bf16=True)
    beta = float(pr.beta(tt)[0])
    _, rates = native.tauleap_step_s256(logits.cuda().contiguous(), x.to(torch.int32).cuda(), tabs, 0, beta, 1e-3, 1, 1, 0,
                                        want_rates=True)
    q, r = op.transition(tt).repeat(N, 1, 1), op.rate(tt).repeat(N, 1, 1)
    ref = ops.zero_own_state(ops.reverse_rates_ctelbo(logits, x, q, r, 1e-9)[0], x)
    np.testing.assert_allclose(rates.cpu().numpy(), ref.numpy(), rtol=B16_RTOL, atol=1e-25)
    # and the three-product kernel on the same inputs: the two modes differ by the bf16 rounding only
    tabs3 = native.S256Tables(qt0, pr.base_rate, 1e-9)
    _, rates3 = native.tauleap_step_s256(logits.cuda().contiguous(), x.to(torch.int32).cuda(), tabs3, 0, beta, 1e-3, 1, 1, 0,
                                         want_rates=True)
    np.testing.assert_allclose(rates.cpu().numpy(), rates3.cpu().numpy(), rtol=B16_RTOL, atol=1e-25)


@pytest.mark.parametrize("flags,crm", [(1, False), (0, False), (3, False), (1, True), (3, True)])
def test_b16_fused_step_replays_from_its_own_rates(env, flags, crm):
    """The draw of the bf16 kernel replayed on the CPU (oracle/philox.py) from the rates the same kernel reports: same Philox
    stream, same rule.  The in-group cumulative offsets are parked in LDS as bf16, so a pick within 2^-9 of a boundary may land
    on the neighbouring destination: the mismatch bar is 2e-2 of the decided rows (the parity kernel: 3e-3)."""
    native, pr, op = env
    N, D = 4, 300
    logits, x = _case(N, D, 5 + int(crm), 2.0)
    tt = torch.tensor([0.4])
    qt0 = pr.tables(tt, want_qt0=True)[0]
    beta = float(pr.beta(tt)[0])
    q, r = op.transition(tt).repeat(N, 1, 1), op.rate(tt).repeat(N, 1, 1)
    rr = (ops.reverse_rates_crm("reverse_prob", logits, x, q, r) if crm else ops.reverse_rates_ctelbo(logits, x, q, r, 1e-9))[0]
    if flags & 2:
        rr = rr + ops.transpose_forward_rates(r, x)
    h = float(1.0 / ops.zero_own_state(rr, x).sum(-1).median())
    tabs = native.S256Tables(qt0, pr.base_rate, 0.0 if crm else 1e-9, crm=crm, bf16=True)
    dl, dx = logits.cuda().contiguous(), x.to(torch.int32).cuda()
    changed = torch.zeros(1, dtype=torch.int32, device="cuda")
    out = native.tauleap_step_s256(dl, dx, tabs, 0, beta, h, flags, 77, 3, changed=changed).cpu().long()
    _, rates = native.tauleap_step_s256(dl, dx, tabs, 0, beta, h, flags, 77, 3, want_rates=True, want_x=False)
    np.testing.assert_allclose(rates.cpu().numpy(), ops.zero_own_state(rr, x).numpy(), rtol=B16_RTOL, atol=1e-25)
    ref, decided = oph.tauleap_draw_replay(rates.cpu().numpy(), x.numpy(), h, bool(flags & 1), 77, 3)
    dec = torch.from_numpy(decided)
    assert dec.float().mean() > 0.6
    assert (out[dec] != torch.from_numpy(ref)[dec]).float().mean().item() < 2e-2
    assert int(changed.item()) == int((out != x).sum())
    assert (out != x).float().mean() > 0.05
    # same stream as the parity kernel: the two modes decide alike except where the 2^-8 rate difference flips a comparison
    tabs3 = native.S256Tables(qt0, pr.base_rate, 0.0 if crm else 1e-9, crm=crm)
    out3 = native.tauleap_step_s256(dl, dx, tabs3, 0, beta, h, flags, 77, 3).cpu().long()
    assert (out != out3).float().mean().item() < 3e-2


def test_b16_dense_rows_and_mixed_regimes(env):
    """A step size that puts rows on both sides of Lambda = 64 (superposition / dense sub-block draw) in the same waves.
    Dense rows replay exactly up to float near-ties.  Rows just below 64 draw 30-60 destinations each; every one of them
    is resolved inside its group of 8 from bf16 offsets (2^-8), so the SUM of the moves may differ from the fp32 replay by a
    destination or two: there the test bounds the size of the difference, not its frequency."""
    native, pr, op = env
    N, D = 2, 200
    logits, x = _case(N, D, 31, 2.0)
    tt = torch.tensor([0.8])
    qt0 = pr.tables(tt, want_qt0=True)[0]
    beta = float(pr.beta(tt)[0])
    q, r = op.transition(tt).repeat(N, 1, 1), op.rate(tt).repeat(N, 1, 1)
    rr = ops.reverse_rates_ctelbo(logits, x, q, r, 1e-9)[0]
    tot = ops.zero_own_state(rr, x).sum(-1)
    h = float(64.0 / tot.median())                       # half the rows above 64, half below
    tabs = native.S256Tables(qt0, pr.base_rate, 1e-9, bf16=True)
    dl, dx = logits.cuda().contiguous(), x.to(torch.int32).cuda()
    for flags in (1, 0):
        out = native.tauleap_step_s256(dl, dx, tabs, 0, beta, h, flags, 5, 9).cpu().long()
        _, rates = native.tauleap_step_s256(dl, dx, tabs, 0, beta, h, flags, 5, 9, want_rates=True, want_x=False)
        lam = rates.cpu().sum(-1) * h
        is_dense = lam > 64
        assert 0.2 < is_dense.float().mean() < 0.8
        ref, decided = oph.tauleap_draw_replay(rates.cpu().numpy(), x.numpy(), h, bool(flags & 1), 5, 9)
        ref, dec = torch.from_numpy(ref), torch.from_numpy(decided)
        assert dec.float().mean() > 0.5
        assert (out[dec & is_dense] != ref[dec & is_dense]).float().mean().item() < 2e-2
        sp = dec & ~is_dense
        assert (out[sp] - ref[sp]).abs().max().item() <= 3 and (out[sp] - ref[sp]).abs().float().mean().item() < 0.5
        assert int(out.min()) >= 0 and int(out.max()) <= S - 1


def test_b16_midpoint_base_and_ragged_tail(env):
    native, pr, op = env
    N, D = 1, 131
    logits, x = _case(N, D, 9, 2.0)
    xb = (x + torch.randint(-3, 4, x.shape)).clamp(0, S - 1)
    tt = torch.tensor([0.6])
    qt0 = pr.tables(tt, want_qt0=True)[0]
    beta = float(pr.beta(tt)[0])
    tabs = native.S256Tables(qt0, pr.base_rate, 1e-9, bf16=True)
    q, r = op.transition(tt).repeat(N, 1, 1), op.rate(tt).repeat(N, 1, 1)
    rr = ops.reverse_rates_ctelbo(logits, xb, q, r, 1e-9)[0]
    h = float(1.0 / ops.zero_own_state(rr, xb).sum(-1).median())
    dl, dx, dxb = logits.cuda().contiguous(), x.to(torch.int32).cuda(), xb.to(torch.int32).cuda()
    out = native.tauleap_step_s256(dl, dx, tabs, 0, beta, h, 0, 5, 1, x_base=dxb).cpu().long()
    _, rates = native.tauleap_step_s256(dl, dx, tabs, 0, beta, h, 0, 5, 1, x_base=dxb, want_rates=True, want_x=False)
    np.testing.assert_allclose(rates.cpu().numpy(), ops.zero_own_state(rr, xb).numpy(), rtol=B16_RTOL, atol=1e-25)
    ref, decided = oph.tauleap_draw_replay(rates.cpu().numpy(), x.numpy(), h, False, 5, 1, x_base=xb.numpy())
    dec = torch.from_numpy(decided)
    assert (out[dec] != torch.from_numpy(ref)[dec]).float().mean().item() < 2e-2 and dec.float().mean() > 0.5


def test_b16_full_size_properties_and_jump_law(env):
    """BASELINE size (N=256, D=784): determinism, stream separation, h = 0, range, change counter -- and the law of the move:
    the mean and the variance of the ordinal jump per dimension against the rates' own first two moments
    (E J = h sum_s r_s (s - x), Var J = h sum_s r_s (s - x)^2 for independent Poisson counts), over 200 704 dimensions."""
    native, pr, _ = env
    N, D = 256, 784
    g = torch.Generator(device="cuda").manual_seed(3)
    logits = torch.randn((N, D, S), generator=g, device="cuda") * 2.0
    x = torch.randint(40, S - 40, (N, D), generator=g, device="cuda", dtype=torch.int32)
    tt = torch.tensor([0.4])
    tabs = native.S256Tables(pr.transition(tt), pr.base_rate, 1e-9, bf16=True)
    beta = float(pr.beta(tt)[0])
    cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    h = 2e-3
    a = native.tauleap_step_s256(logits, x, tabs, 0, beta, h, native.STEP_ORDINAL, 7, 11, changed=cnt)
    b = native.tauleap_step_s256(logits, x, tabs, 0, beta, h, native.STEP_ORDINAL, 7, 11)
    c = native.tauleap_step_s256(logits, x, tabs, 0, beta, h, native.STEP_ORDINAL, 7, 12)
    z = native.tauleap_step_s256(logits, x, tabs, 0, beta, 0.0, native.STEP_ORDINAL, 7, 11)
    assert torch.equal(a, b) and not torch.equal(a, c) and torch.equal(z, x)
    assert int(a.min()) >= 0 and int(a.max()) <= S - 1
    moved = int((a != x).sum())
    assert int(cnt.item()) == moved and 0 < moved < N * D
    d = native.tauleap_step_s256(logits, x, tabs, 0, beta, h, 0, 7, 11)
    assert int((d != x).sum()) <= moved
    _, rates = native.tauleap_step_s256(logits, x, tabs, 0, beta, h, native.STEP_ORDINAL, 7, 11, want_rates=True, want_x=False)
    sx = (torch.arange(S, device="cuda")[None, None, :] - x[..., None]).double()
    mean = (rates.double() * sx).sum(-1) * h
    var = (rates.double() * sx * sx).sum(-1) * h
    J = (a - x).double()
    inner = (a > 0) & (a < S - 1)                      # (the clamp at the borders censors the move)
    assert inner.float().mean() > 0.99
    n = float(inner.sum())
    zscore = float(((J - mean)[inner]).sum() / var[inner].sum().sqrt())
    assert abs(zscore) < 5.0, zscore
    ratio = float(((J - mean)[inner] ** 2).sum() / var[inner].sum())
    assert abs(ratio - 1.0) < 0.05, ratio


def test_b16_bf16_logits_input(env):
    """CTDD_STEP_LOGITS_BF16: the bf16 kernel reading bf16 logits (what the bf16 U-Net engine's output convolution writes on
    request) against the same kernel reading the same values widened to fp32 -- same rates up to the order of the row sums, same
    draws up to float near-ties; and the ragged tail / x_base / rate output of the general variant."""
    native, pr, op = env
    N, D = 3, 211
    logits, x = _case(N, D, 41, 3.0)
    lb = logits.to(torch.bfloat16)
    tt = torch.tensor([0.45])
    qt0 = pr.tables(tt, want_qt0=True)[0]
    beta = float(pr.beta(tt)[0])
    tabs = native.S256Tables(qt0, pr.base_rate, 1e-9, bf16=True)
    dx = x.to(torch.int32).cuda()
    d16, d32 = lb.cuda().contiguous(), lb.float().cuda().contiguous()
    _, r16 = native.tauleap_step_s256(d16, dx, tabs, 0, beta, 1e-3, 1, 3, 0, want_rates=True, want_x=False)
    _, r32 = native.tauleap_step_s256(d32, dx, tabs, 0, beta, 1e-3, 1, 3, 0, want_rates=True, want_x=False)
    np.testing.assert_allclose(r16.cpu().numpy(), r32.cpu().numpy(), rtol=1e-5, atol=1e-30)
    q, r = op.transition(tt).repeat(N, 1, 1), op.rate(tt).repeat(N, 1, 1)
    ref = ops.zero_own_state(ops.reverse_rates_ctelbo(lb.float(), x, q, r, 1e-9)[0], x)
    np.testing.assert_allclose(r16.cpu().numpy(), ref.numpy(), rtol=B16_RTOL, atol=1e-25)
    h = float(1.0 / ref.sum(-1).median())
    for flags in (1, 0, 3):
        a16 = native.tauleap_step_s256(d16, dx, tabs, 0, beta, h, flags, 7, 2).cpu()
        a32 = native.tauleap_step_s256(d32, dx, tabs, 0, beta, h, flags, 7, 2).cpu()
        assert (a16 != a32).float().mean().item() < 3e-3
        assert (a16 != x).float().mean() > 0.05
    with pytest.raises(native.CtddError):                  # bf16 logits are for the bf16 step only
        native.tauleap_step_s256(d16, dx, native.S256Tables(qt0, pr.base_rate, 1e-9), 0, beta, h, 1, 7, 2)
