"""GPU: the hollow-transformer TRAINING path on HIP kernels (ctdd/hollow_train.py over csrc/hollow_train_kernels.hip, the
implicit-GEMM kernels and ctdd_unet_wgrad) against autograd through the torch module (the reference's
`l.backward()`, TAUnSDDM/lib/training/training.py:27 through lib/networks/hollow_networks.py:668-755)."""
import ast

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

T = torch.from_numpy


def _tiny(golden, tag, p_drop=0.0, p_att=0.0):
    import lib.models.models  # noqa: F401
    import lib.models.model_utils as mu
    from config.maze_config.config_hollow_maze import get_config
    g = golden("hollow")
    meta = ast.literal_eval(str(g[f"{tag}__cfg"]))
    cfg = get_config()
    cfg.device = "cuda"
    cfg.data.S = meta["S"]
    cfg.model.update(concat_dim=meta["D"], embed_dim=meta["embed_dim"], num_layers=meta["num_layers"], num_heads=meta["num_heads"],
                     mlp_dim=meta["mlp_dim"], qkv_dim=meta["embed_dim"], readout_dim=meta["S"], t_func=meta["t_func"],
                     dropout_rate=p_drop, attention_dropout_rate=p_att)
    model = mu.create_model(cfg, torch.device("cuda"))
    pre = f"{tag}__sd__"
    sd = {k[len(pre):]: T(v).cuda() for k, v in g.items() if k.startswith(pre)}
    missing, unexpected = torch.nn.Module.load_state_dict(model, sd, strict=False)
    assert not missing and not unexpected
    model.init_ema()
    return cfg, model, T(g[f"{tag}__x"]).cuda().long(), T(g[f"{tag}__t"]).cuda(), g[f"{tag}__out"]


def _grads(model, fn):
    for p in model.parameters():
        p.grad = None
    out = fn()
    return out, {n: (None if p.grad is None else p.grad.detach().clone()) for n, p in model.named_parameters()}


def _compare(g_hip, g_ref, tol, l2=False):
    """max-abs difference relative to the tensor's range; l2: relative L2 error (the bf16 modes: a ReLU whose bf16
    pre-activation changes sign moves single entries by O(1) of the range)."""
    worst = 0.0
    # floor of the per-tensor scale: gradients that are zero in exact arithmetic (the readout's key bias: a softmax does
    # not see a constant added to every key) are rounding noise on both sides
    floor = 1e-3 * max(float(r.abs().max()) for r in g_ref.values() if r is not None)
    for n, r in g_ref.items():
        h = g_hip[n]
        if r is None:
            assert h is None or float(h.abs().max()) == 0.0, n
            continue
        assert h is not None, f"no gradient for {n}"
        scale = max(float(r.abs().max()), floor)
        err = float((h - r).norm() / max(float(r.norm()), floor)) if l2 else float((h - r).abs().max()) / scale
        worst = max(worst, err)
        assert err < tol, f"{n}: gradient differs by {err:.3e} of its range (bar {tol})"
    return worst


@pytest.mark.parametrize("tag", ["s3", "s2"])
def test_hollow_train_matches_autograd_on_golden_nets(golden, tag):
    """Forward logits against the reference's golden output (1e-4) and every parameter gradient against autograd
    through the torch module, on the nets of tests/golden/hollow.npz (dropout 0)."""
    from ctdd.hollow_train import HollowTrainer, training_supported
    cfg, model, x, t, ref = _tiny(golden, tag)
    assert training_supported(model)
    torch.manual_seed(3)
    wgt = torch.randn(ref.shape, device="cuda")
    cfg.model.engine = "torch"
    out_ref, g_ref = _grads(model, lambda: (lambda o: ((o * wgt).sum().backward(), o.detach())[1])(model(x, t)))
    cfg.model.engine = "hip"
    tr = HollowTrainer(model, precision="fp32")
    out, g_hip = _grads(model, lambda: (lambda o: ((o * wgt).sum().backward(), o.detach())[1])(tr(x, t)))
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=0, atol=1e-4)            # BASELINE bar, training-mode forward
    np.testing.assert_allclose(out.cpu().numpy(), out_ref.cpu().numpy(), rtol=0, atol=1e-4)
    _compare(g_hip, g_ref, 1e-4)
    # the wrapper routes grad-enabled calls to the same path (default precision: bf16 GEMM operands)
    out_w, g_w = _grads(model, lambda: (lambda o: ((o * wgt).sum().backward(), o.detach())[1])(model(x, t)))
    assert model._trainer is not None
    assert float((out_w - out_ref).abs().max()) < 5e-2 * max(1.0, float(out_ref.abs().max()))
    _compare(g_w, g_ref, 0.25, l2=True)      # few tokens: single bf16 ReLU flips weigh percents (maze size below: 0.1)


def test_hollow_train_matches_autograd_maze_size():
    """config_hollow_maze (D=225, S=3, E=128, 8 layers per direction), batch 4, dropout 0."""
    import lib.models.models  # noqa: F401
    import lib.models.model_utils as mu
    from config.maze_config.config_hollow_maze import get_config
    from ctdd.hollow_train import HollowTrainer
    cfg = get_config()
    cfg.device = "cuda"
    cfg.model.update(dropout_rate=0.0, attention_dropout_rate=0.0)
    torch.manual_seed(0)
    model = mu.create_model(cfg, torch.device("cuda"))
    x = torch.randint(0, 3, (4, 225), device="cuda")
    t = torch.tensor([0.02, 0.3, 0.5, 0.99], device="cuda")
    wgt = torch.randn((4, 225, 3), device="cuda")
    run = lambda f: (lambda o: ((o * wgt).sum().backward(), o.detach())[1])(f(x, t))
    cfg.model.engine = "torch"
    out_ref, g_ref = _grads(model, lambda: run(model))
    cfg.model.engine = "hip"
    out, g_hip = _grads(model, lambda: run(HollowTrainer(model, precision="fp32")))
    assert float((out - out_ref).abs().max()) < 2e-4 * max(1.0, float(out_ref.abs().max()))
    # 16 blocks deep, ~1e6 ReLU pre-activations per layer: a few lie within rounding of zero and flip between two fp32
    # evaluations, moving single gradient entries by percents of the range.  Against a float64 evaluation of the module
    # (tools/hollow_grad_f64.py) torch's own fp32 gradients are off by up to 3.1e-3 in relative L2 and 3.0e-2 in max-abs,
    # this path by 4.9e-3 / 3.2e-2; the tiny nets above (no such flips) hold 1e-4 max-abs.
    _compare(g_hip, g_ref, 1e-2, l2=True)
    out_b, g_b = _grads(model, lambda: run(HollowTrainer(model, precision="bf16")))
    assert float((out_b - out_ref).abs().max()) < 5e-2 * max(1.0, float(out_ref.abs().max()))
    _compare(g_b, g_ref, 0.1, l2=True)


def test_hollow_train_dropout_masks_are_consistent(golden):
    """Dropout on (residual, MLP and attention-probability sites): the backward regenerates the forward's Philox masks --
    the analytic directional derivative equals a central finite difference taken with the SAME masks; another step
    draws different masks; eval mode is deterministic and equals the dropout-free network."""
    from ctdd.hollow_train import HollowTrainer
    cfg, model, x, t, ref = _tiny(golden, "s3", p_drop=0.2, p_att=0.2)
    tr = HollowTrainer(model, precision="fp32")
    torch.manual_seed(5)
    wgt = torch.randn(ref.shape, device="cuda", dtype=torch.float64)
    params = [p for n, p in model.named_parameters() if not n.startswith(("net.embedding", "net.temb_net"))]
    dirs = [torch.randn_like(p) for p in params]

    def loss_at(step):
        tr.rng[1] = step                                  # the forward bumps it: masks of step + 1
        return (tr(x, t).double() * wgt).sum()

    for p in model.parameters():
        p.grad = None
    l0 = loss_at(10)
    l0.backward()
    analytic = sum(float((p.grad.double() * d.double()).sum()) for p, d in zip(params, dirs))
    eps = 2e-3
    with torch.no_grad():
        for p, d in zip(params, dirs):
            p.add_(eps * d)
        lp = float(loss_at(10))
        for p, d in zip(params, dirs):
            p.sub_(2 * eps * d)
        lm = float(loss_at(10))
        for p, d in zip(params, dirs):
            p.add_(eps * d)
        again = float(loss_at(10))
        other = float(loss_at(11))
    fd = (lp - lm) / (2 * eps)
    l0v = float(l0.detach())
    assert abs(again - l0v) < 1e-6 * max(1.0, abs(l0v))                      # same step -> same masks
    assert abs(other - l0v) > 1e-4 * max(1.0, abs(l0v))                      # next step -> different masks
    assert abs(fd - analytic) < 2e-2 * max(abs(analytic), 1.0), (fd, analytic)
    model.eval()
    with torch.enable_grad():
        out_eval = tr(x, t)
    np.testing.assert_allclose(out_eval.detach().cpu().numpy(), ref, rtol=0, atol=1e-4)
    model.train()


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_hollow_train_two_forwards_before_backward_keep_their_masks(golden, precision):
    """A second training forward before the first one's backward (the two-forward-pass CT-ELBO, `model(x_t)` then
    `model(x_tilde)`; gradient accumulation) must not move the first forward's dropout masks: every forward carries its
    own {seed, step} snapshot.  Gradients of the interleaved run equal those of each forward run and backpropagated alone."""
    from ctdd.hollow_train import HollowTrainer
    cfg, model, x, t, ref = _tiny(golden, "s3", p_drop=0.2, p_att=0.2)
    tr = HollowTrainer(model, precision=precision)
    torch.manual_seed(7)
    w1 = torch.randn(ref.shape, device="cuda")
    w2 = torch.randn(ref.shape, device="cuda")
    x2 = (x + 1) % cfg.data.S

    def alone(step, xx, ww):
        tr.rng[1] = step
        return _grads(model, lambda: (tr(xx, t) * ww).sum().backward())[1]

    g1 = alone(20, x, w1)                                  # masks of step 21
    g2 = alone(21, x2, w2)                                 # masks of step 22

    def both():
        tr.rng[1] = 20
        o1 = tr(x, t)                                      # step 21
        o2 = tr(x2, t)                                     # step 22, before the first backward
        ((o1 * w1).sum() + (o2 * w2).sum()).backward()

    gb = _grads(model, both)[1]
    tol = 1e-4 if precision == "fp32" else 2e-2
    for n, a in g1.items():
        if a is None:
            continue
        want = a + g2[n]
        scale = max(float(want.abs().max()), 1e-6)
        err = float((gb[n] - want).abs().max()) / scale
        # (bf16: the sum of two backward passes accumulates in another order; fp32 atomics of the weight-gradient kernel likewise)
        assert err < tol, f"{n}: interleaved forwards changed the gradient by {err:.3e} of its range"


def test_hollow_train_dropout_rate():
    """The attention-probability dropout keeps 1 - p of the entries and rescales by 1 / (1 - p)."""
    from ctdd.hollow_train import AttentionFn, DropoutFn, ActFn
    torch.manual_seed(0)
    B, D, H, hd, p = 3, 40, 4, 8, 0.25
    E = H * hd
    rng = torch.tensor([1234, 7], dtype=torch.int64, device="cuda")
    qkv = torch.zeros((B * D, 3 * E), device="cuda")
    qkv[:, 2 * E:] = 1.0                                         # uniform probabilities, V = 1: out = (#kept / n) / (1 - p)
    out = AttentionFn.apply(qkv, None, None, B, D, D, H, hd, 0, p, rng, 3).view(B, D, H, hd)
    n = torch.arange(1, D + 1, device="cuda").view(1, D, 1, 1).float()
    kept = out * n * (1 - p)
    assert float((kept - kept.round()).abs().max()) < 1e-3
    frac = float(kept[..., 0].sum() / (B * H * n.sum()))
    assert abs(frac - (1 - p)) < 0.02
    z = torch.ones((64, 1024), device="cuda")
    y = DropoutFn.apply(z, p, rng, 5)
    assert abs(float((y != 0).float().mean()) - (1 - p)) < 0.01 and abs(float(y.max()) - 1 / (1 - p)) < 1e-6
    y2 = ActFn.apply(z, 1, p, rng, 6)
    assert abs(float((y2 != 0).float().mean()) - (1 - p)) < 0.01
    assert not torch.equal(y != 0, y2 != 0)                      # another layer id, another mask


@pytest.mark.parametrize("hd", [16, 32])
@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("p", [0.0, 0.2])
def test_hollow_attention_matrix_core_kernels_match_fp32_kernels(mode, hd, p):
    """ctdd_hollow_attention_train_bf16 / _bwd_bf16 (v_mfma_f32_32x32x16_bf16) against the fp32 FMA kernels on the same
    inputs and the SAME Philox dropout masks: maze-sized sequences (225 queries; 451 keys in the readout mode)."""
    from ctdd.hollow_train import AttentionFn
    torch.manual_seed(mode * 7 + hd)
    B, D, H = 3, 225, 4
    E = H * hd
    Tk = 2 * D + 1 if mode == 2 else D
    rng = torch.tensor([99, 5], dtype=torch.int64, device="cuda")
    res = {}
    if mode == 2:
        base = [torch.randn((B * D, E), device="cuda"), torch.randn((B * Tk, E), device="cuda"), torch.randn((B * Tk, E), device="cuda")]
    else:
        base = [torch.randn((B * D, 3 * E), device="cuda")]
    wgt = torch.randn((B * D, E), device="cuda")
    for bf in (False, True):
        ins = [t.clone().requires_grad_(True) for t in base]
        q, k, v = (ins + [None, None])[:3]
        out = AttentionFn.apply(q, k, v, B, D, Tk, H, hd, mode, p, rng if p > 0 else None, 4, bf)
        (out * wgt).sum().backward()
        res[bf] = [out.detach()] + [t.grad for t in ins]
    for ref, got in zip(res[False], res[True]):
        scale = float(ref.abs().max())
        assert float((got - ref).abs().max()) < 2e-2 * scale, (float((got - ref).abs().max()), scale)
        assert float((got - ref).norm() / ref.norm()) < 6e-3


def test_hollow_score_elbo_training_step_matches_torch():
    """One ScoreElbo training step (loss value, clipped-Adam update) through the model wrapper: HIP training path against
    the torch module, dropout 0 (maze hollow configuration at reduced depth)."""
    import lib.models.models  # noqa: F401
    import lib.losses.losses  # noqa: F401
    import lib.models.model_utils as mu
    import lib.losses.losses_utils as lu
    from config.maze_config.config_hollow_maze import get_config
    res = {}
    for engine in ("torch", "hip"):
        cfg = get_config()
        cfg.device = "cuda"
        cfg.model.update(dropout_rate=0.0, attention_dropout_rate=0.0, num_layers=2, engine=engine, engine_train_precision="fp32")
        torch.manual_seed(0)
        model = mu.create_model(cfg, torch.device("cuda"))
        loss_fn = lu.get_loss(cfg)
        x = torch.randint(0, 3, (8, 225), device="cuda")
        torch.manual_seed(11)
        l = loss_fn.calc_loss(x, {"model": model, "n_iter": 0})
        for p in model.parameters():
            p.grad = None
        l.backward()
        res[engine] = (float(l.detach()), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None})
        if engine == "hip":
            assert model._trainer is not None
    assert abs(res["hip"][0] - res["torch"][0]) < 1e-4 * max(1.0, abs(res["torch"][0]))
    _compare(res["hip"][1], res["torch"][1], 2e-3)


def test_hollow_elementwise_training_kernels():
    """ctdd_hollow_colsum (two-stage column sums), ctdd_hollow_dropout (mask + residual + bf16 copy in one pass) and
    ctdd_hollow_relu_bf16 (bf16-only ReLU + dropout whose saved output is its own backward mask) against torch / against
    the fp32 kernel ctdd_hollow_act on the same Philox masks."""
    from ctdd import hollow_train as ht
    torch.manual_seed(2)
    rng = torch.tensor([77, 3], dtype=torch.int64, device="cuda")
    # column sums: fp32 and bf16 inputs, N not a multiple of the vector width's row split, padded leading dimension
    for rows, N, ld, bf in ((5000, 136, 136, False), (28800, 1024, 1024, True), (300, 3, 16, True), (777, 384, 384, False)):
        x = torch.randn((rows, ld), device="cuda")
        x[:, N:] = 0.0
        xo = x.to(torch.bfloat16) if bf else x
        got = ht._colsum(xo, rows, N, ld, bf)
        ref = xo.float()[:, :N].double().sum(0)
        assert float((got.double() - ref).abs().max()) < 1e-4 * max(1.0, float(ref.abs().max())) + 1e-3
    # dropout (+ residual, + bf16 copy): both outputs carry the same mask; p = 0 is the plain add / cast
    x, r = torch.randn((640, 128), device="cuda"), torch.randn((640, 128), device="cuda")
    p = 0.3
    o32, o16 = ht._dropout(x, p, rng, 9, res=r, want_f32=True, want_hi=True)
    kept = (o32 - r).abs() > 0
    assert abs(float(kept.float().mean()) - (1 - p)) < 0.01
    assert torch.allclose((o32 - r)[kept], (x / (1 - p))[kept], rtol=1e-5, atol=1e-5)
    assert torch.equal(o16, o32.to(torch.bfloat16))
    again, _ = ht._dropout(x, p, rng, 9)
    assert torch.equal(again != 0, kept)                                         # same (seed, step, layer) -> same mask
    other, _ = ht._dropout(x, p, rng, 10)
    assert not torch.equal(other != 0, kept)
    plain, cast = ht._dropout(x, 0.0, None, 0, res=r, want_f32=True, want_hi=True)
    assert torch.equal(plain, x + r) and torch.equal(cast, (x + r).to(torch.bfloat16))
    # bf16 ReLU + dropout: forward equals the fp32 kernel's result on the bf16-rounded input (same element -> mask mapping),
    # backward = gradient * [saved output != 0] / (1 - p) equals the fp32 kernel's backward
    pre = torch.randn((512, 1024), device="cuda").to(torch.bfloat16)
    u = pre.clone()
    ht._ck(ht.lib().ctdd_hollow_relu_bf16(u.data_ptr(), None, u.data_ptr(), u.numel(), p, rng.data_ptr(), 4, ht._st()), "relu_bf16")
    u_ref, _ = ht._act(pre.float().contiguous(), None, 1, p, rng, 4)
    assert torch.equal(u, u_ref.to(torch.bfloat16))
    du = torch.randn((512, 1024), device="cuda").to(torch.bfloat16)
    dpre = du.clone()
    ht._ck(ht.lib().ctdd_hollow_relu_bf16(dpre.data_ptr(), u.data_ptr(), dpre.data_ptr(), dpre.numel(), p, None, 0, ht._st()), "relu_bf16 bwd")
    d_ref, _ = ht._act(pre.float().contiguous(), du.float().contiguous(), 1, p, rng, 4)
    assert torch.equal(dpre, d_ref.to(torch.bfloat16))


@pytest.mark.parametrize("E", [32, 128, 256, 512])
@pytest.mark.parametrize("with_y,with_film,with_res", [(False, False, True), (True, True, False), (True, False, True)])
def test_hollow_layernorm_backward_kernel(E, with_y, with_film, with_res):
    """ctdd_hollow_layernorm_bwd (all three column-count instantiations: E <= 128, <= 256, <= 512) against autograd of
    FiLM(LayerNorm(x + y)): dx (+ the residual gradient added out of place), dgamma / dbeta through the replicated
    accumulators, the per-sample FiLM gradients."""
    from ctdd import hollow_train as ht
    torch.manual_seed(E + with_y)
    B, T = 3, 37
    x = torch.randn(B, T, E, device="cuda", requires_grad=True)
    y = torch.randn(B, T, E, device="cuda", requires_grad=True) if with_y else None
    gamma = (1 + 0.1 * torch.randn(E, device="cuda")).requires_grad_(True)
    beta = (0.1 * torch.randn(E, device="cuda")).requires_grad_(True)
    film = torch.randn(B, 2 * E, device="cuda", requires_grad=True) if with_film else None
    dout = torch.randn(B, T, E, device="cuda")
    dres = torch.randn(B, T, E, device="cuda") if with_res else None
    h = x + y if with_y else x
    z = torch.nn.functional.layer_norm(h, (E,), gamma, beta, 1e-5)
    if with_film:
        z = film[:, None, :E] * z + film[:, None, E:]
    z.backward(dout)
    out, _ = ht._layernorm(x.detach(), None if y is None else y.detach(), gamma.detach(), beta.detach(), None if film is None else film.detach(), 1e-5)
    assert float((out - z.detach()).abs().max()) < 1e-5 * max(1.0, float(z.detach().abs().max()))
    dx, dg, db, dfilm = ht._layernorm_bwd(x.detach(), None if y is None else y.detach(), gamma.detach(), beta.detach(),
                                          None if film is None else film.detach(), 1e-5, dout, dres=dres)
    ref_dx = x.grad + (dres if with_res else 0)
    tol = lambda r: 2e-5 * max(1.0, float(r.abs().max()))
    assert float((dx - ref_dx).abs().max()) < tol(ref_dx)
    if with_y:
        assert float((y.grad - x.grad).abs().max()) == 0.0                         # (the same gradient flows to both summands)
    assert float((dg - gamma.grad).abs().max()) < tol(gamma.grad)
    assert float((db - beta.grad).abs().max()) < tol(beta.grad)
    if with_film:
        assert float((dfilm - film.grad).abs().max()) < tol(film.grad)


def test_hollow_mnist_config_catrmnll_step_matches_torch():
    """BASELINE config 3 (MNIST SDDM hollow, D = 784, S = 256, E = 256, head dimension 32; CatRMNLL with reverse_prob logits) at
    reduced depth, batch 2, dropout 0: loss value and every parameter gradient of the HIP path (fp32 training mode; the
    objective's softmax @ q contractions on the exact-fp32 matrix instruction) against the torch module + torch objective."""
    import lib.models.models  # noqa: F401
    import lib.losses.losses  # noqa: F401
    import lib.models.model_utils as mu
    import lib.losses.losses_utils as lu
    from config.mnist_config.config_hollow_mnist import get_config
    from ctdd import native
    res = {}
    for engine in ("torch", "hip"):
        cfg = get_config()
        cfg.device = "cuda"
        cfg.loss.name = "CatRMNLL"
        cfg.loss.fused = engine == "hip"                       # torch side: the reference formula on device ops
        cfg.model.update(dropout_rate=0.0, attention_dropout_rate=0.0, num_layers=1, engine=engine, engine_train_precision="fp32")
        torch.manual_seed(0)
        model = mu.create_model(cfg, torch.device("cuda"))
        loss_fn = lu.get_loss(cfg)
        x = torch.randint(0, 256, (2, 784), device="cuda")
        torch.manual_seed(11)
        before = native.LAUNCH_COUNTS.get("ctdd_logprob_bwd", 0)
        l = loss_fn.calc_loss(x, {"model": model, "n_iter": 0})
        for p in model.parameters():
            p.grad = None
        l.backward()
        res[engine] = (float(l.detach()), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None})
        if engine == "hip":
            assert model._trainer is not None and native.LAUNCH_COUNTS.get("ctdd_logprob_bwd", 0) == before + 1
    assert abs(res["hip"][0] - res["torch"][0]) < 2e-4 * max(1.0, abs(res["torch"][0]))
    _compare(res["hip"][1], res["torch"][1], 5e-3, l2=True)
