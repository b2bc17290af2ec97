"""GPU: the hand-written U-Net BACKWARD (csrc/unet_train_kernels.hip + ctdd/unet_train.py) against torch autograd of the
same operations (the reference trains through autograd: lib/training/training.py:27, lib/networks/unet.py:100-140, 303-459).
Kernel-level cases first (weight gradient, transposed stride-2 convolution, GroupNorm + Swish backward, attention backward),
then every parameter gradient of whole networks: the reference's golden tiny U-Nets and the MNIST configuration."""
import ast
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

T = torch.from_numpy


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _check(rc, lib):
    assert rc == 0, lib.ctdd_last_error().decode()


# ---------------------------------------------------------------------------------------------- weight gradient
WGRAD_CASES = [
    (3, 28, 28, 96, 96, 0, 2, 5, 7),      # MNIST level-0 shape; chunks straddle images
    (5, 14, 14, 192, 96, 0, 2, 5, 3),
    (4, 7, 7, 64, 192, 0, 2, 11, 2),      # several images per chunk (zero rows between them)
    (2, 8, 8, 16, 32, 0, 1, 11, 1),       # golden tiny net: one n tile
    (2, 8, 8, 32, 256, 0, 4, 11, 2),      # C <= 32
    (3, 14, 14, 96, 192, 1, 2, 112, 4),   # linear skip (1x1)
    (2, 7, 7, 192, 576, 1, 2, 64, 1),     # attention qkv
    (3, 14, 14, 96, 96, 2, 2, 112, 3),    # Downsample: stride 2 from 28x28
    (2, 3, 3, 32, 32, 2, 2, 16, 1),       # stride 2 from an odd 7x7 grid
]


def _wgrad_case(case, f32, g):
    from ctdd import unet_train as ut
    B, H, W, Cc, N, kind, nwn, nlr, gx = case
    Hin, Win = (H, W) if kind != 2 else (2 * H + (1 if H == 3 else 0), 2 * W + (1 if W == 3 else 0))
    dt = torch.float32 if f32 else torch.bfloat16
    x = torch.randn((B, Hin, Win, Cc), generator=g, device="cuda").to(dt)
    dy = torch.randn((B, H, W, N), generator=g, device="cuda").to(dt)
    ntap = 1 if kind == 1 else 9
    koff, Ktot = 16, 16 + ntap * Cc + 8                        # the segment sits inside a wider packed matrix
    gw = torch.zeros((N, Ktot), dtype=torch.float32, device="cuda")
    gb = torch.zeros((N,), dtype=torch.float32, device="cuda")
    ents = []
    for tap in (range(9) if kind == 2 else (0,)):
        a = ut._WgradArgs()
        a.x, a.dy, a.gw = x.data_ptr(), dy.data_ptr(), gw.data_ptr()
        a.B, a.H, a.W, a.Hin, a.Win, a.N, a.ldy, a.C, a.Ktot, a.koff, a.kind, a.nwn = B, H, W, Hin, Win, N, N, Cc, Ktot, koff, kind, nwn
        epv = 4 if f32 else 8                                  # staging-slot limits: 8 + 10 sixteen-byte vectors per thread
        vn, vc = 32 * nwn // epv, 32 * (4 // nwn) // epv
        if kind == 0:
            a.nlr = max(1, min(nlr, 2048 // (W * vn), 2560 // (W * vc) - 2))
        else:
            a.nlr = max(16, min(nlr, 2048 // vn, 2560 // vc) // 16 * 16)
        a.nchunks = -(-(B * (H + 1)) // a.nlr) if kind == 0 else -(-(B * H * W) // a.nlr)
        a.grid_x, a.tap = min(gx, a.nchunks), tap
        a.gb = gb.data_ptr() if tap == 0 else None             # the bias gradient rides on one entry per convolution
        ents.append(a)
    # reference: autograd of the convolution on the same (rounded) operands
    w = torch.zeros((N, Cc, 3 if kind != 1 else 1, 3 if kind != 1 else 1), device="cuda", requires_grad=True)
    xin = x.float().permute(0, 3, 1, 2)
    out = F.conv2d(F.pad(xin, [0, 1, 0, 1]), w, stride=2) if kind == 2 else F.conv2d(xin, w, padding=1 if kind == 0 else 0)
    assert out.shape[2:] == (H, W)
    out.backward(dy.float().permute(0, 3, 1, 2))
    ref = w.grad.permute(0, 2, 3, 1).reshape(N, ntap * Cc)       # [n][tap][c]
    return ents, (x, dy, gb), gw, ref, (koff, ntap * Cc)


@pytest.mark.parametrize("f32", [False, True])
def test_wgrad_kernel(f32):
    """ctdd_unet_wgrad: ONE launch over a table of segments (every kind, ragged chunk ends, several images per chunk, a
    segment inside a wider packed matrix) against autograd's convolution weight gradients."""
    from ctdd import unet_train as ut
    lib = ut.lib()
    g = torch.Generator(device="cuda").manual_seed(11)
    cases = [_wgrad_case(c, f32, g) for c in WGRAD_CASES]
    for nine in (True, False):                      # a table holds nine-tap (3x3) entries or one-tap entries
        ents = [a for c in cases for a in c[0] if (a.kind == 0) == nine]
        tab = (ut._WgradArgs * len(ents))(*ents)
        dev_tab = torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8).cuda()
        _check(lib.ctdd_unet_wgrad(dev_tab.data_ptr(), C.addressof(tab), len(ents), int(f32), _stream()), lib)
    torch.cuda.synchronize()
    for case, (_, (_, dy, gb), gw, ref, (koff, width)) in zip(WGRAD_CASES, cases):
        got = gw[:, koff:koff + width]
        scale = ref.abs().max().item()
        assert (got - ref).abs().max().item() < 2e-5 * scale + 1e-6, (case, (got - ref).abs().max().item(), scale)
        bref = dy.float().sum((0, 1, 2))                       # ctdd_wgrad_args.gb: column sums of dy over all pixels
        assert (gb - bref).abs().max().item() < 2e-5 * bref.abs().max().item() + 1e-5, (case, "bias gradient")
        assert gw[:, :koff].abs().max().item() == 0 and gw[:, koff + width:].abs().max().item() == 0


# ---------------------------------------------------------------------------------------------- transposed stride-2 conv
@pytest.mark.parametrize("f32", [False, True])
@pytest.mark.parametrize("B,Hin,Cc,N", [(3, 28, 96, 96), (2, 7, 32, 64)])
def test_downsample_data_gradient(B, Hin, Cc, N, f32):
    """CTDD_SEG_3x3_S2T of the generic convolution kernel = the data gradient of the Downsample conv (unet.py:88-97)."""
    from ctdd import unet_engine as ue
    lib = ue._lib()
    g = torch.Generator(device="cuda").manual_seed(Hin)
    Ho = (Hin + 1 - 3) // 2 + 1
    dt = torch.float32 if f32 else torch.bfloat16
    dy = torch.randn((B, Ho, Ho, N), generator=g, device="cuda").to(dt)
    w = (torch.randn((N, Cc, 3, 3), generator=g, device="cuda") / (9 * Cc) ** 0.5).to(dt)
    xin = torch.zeros((B, Cc, Hin, Hin), device="cuda", requires_grad=True)
    F.conv2d(F.pad(xin, [0, 1, 0, 1]), w.float(), stride=2).backward(dy.float().permute(0, 3, 1, 2))
    ref = xin.grad.permute(0, 2, 3, 1).reshape(B * Hin * Hin, Cc)
    wd = w.permute(1, 2, 3, 0).reshape(Cc, 9 * N).contiguous()          # [c][tap][n], same tap index
    a = ue._ConvArgs()
    a.nseg = 1
    a.seg[0].C, a.seg[0].kind = N, 4
    if f32:
        a.seg[0].f32, a.w_f32 = dy.data_ptr(), wd.data_ptr()
    else:
        a.seg[0].hi, a.w_hi = dy.data_ptr(), wd.data_ptr()
    a.B, a.H, a.W, a.Hin, a.Win, a.N, a.Ktot = B, Hin, Hin, Ho, Ho, Cc, 9 * N
    out = torch.empty((B * Hin * Hin, Cc), dtype=torch.float32, device="cuda")
    a.out_f32 = out.data_ptr()
    rc = lib.ctdd_unet_conv(C.byref(a), 32 if N % 32 == 0 else 16, 1, int(f32), _stream())
    assert rc == 0, lib.ctdd_last_error().decode()
    torch.cuda.synchronize()
    assert (out - ref).abs().max().item() < 2e-5 * ref.abs().max().item() + 1e-6


# ---------------------------------------------------------------------------------------------- GroupNorm + Swish backward
@pytest.mark.parametrize("f32", [False, True])
@pytest.mark.parametrize("C1,C2,swish", [(96, 0, True), (192, 96, True), (64, 0, False)])
def test_groupnorm_backward(C1, C2, swish, f32):
    from ctdd import unet_train as ut
    lib = ut.lib()
    B, HW, Ct = 3, 49, C1 + C2
    G = min(Ct // 4, 32)
    g = torch.Generator(device="cuda").manual_seed(C1 + C2)
    dt = torch.float32 if f32 else torch.bfloat16
    x1 = (torch.randn((B, HW, C1), generator=g, device="cuda") * 1.5 + 0.3).to(dt)
    x2 = (torch.randn((B, HW, max(C2, 8)), generator=g, device="cuda") * 0.7).to(dt)
    da = torch.randn((B, HW, Ct), generator=g, device="cuda").to(dt)
    gamma = torch.randn(Ct, generator=g, device="cuda")
    beta = torch.randn(Ct, generator=g, device="cuda")
    old1 = torch.randn((B, HW, C1), generator=g, device="cuda").to(dt)
    d1, d2 = old1.clone(), torch.zeros((B, HW, max(C2, 8)), device="cuda", dtype=dt)
    xs = torch.cat([x1.float(), x2.float()[..., :C2]], -1)
    stats = torch.stack([xs.double().sum(1), (xs.double() ** 2).sum(1)], -1)           # [B][Ct][2]
    st1, st2 = stats[:, :C1].contiguous(), stats[:, C1:].contiguous()
    sums = torch.zeros((B, Ct, 2), dtype=torch.float32, device="cuda")
    a = ut._GnBwdArgs()
    if f32:
        a.s1_f32, a.s2_f32, a.da_f32, a.d1_f32, a.d2_f32 = x1.data_ptr(), x2.data_ptr(), da.data_ptr(), d1.data_ptr(), d2.data_ptr()
    else:
        a.s1_bf16, a.s2_bf16, a.da_bf16, a.d1_bf16, a.d2_bf16 = x1.data_ptr(), x2.data_ptr(), da.data_ptr(), d1.data_ptr(), d2.data_ptr()
    a.st1, a.C1, a.st2, a.C2 = st1.data_ptr(), C1, st2.data_ptr() if C2 else None, C2
    a.gamma, a.beta, a.B, a.HW, a.G, a.eps, a.swish = gamma.data_ptr(), beta.data_ptr(), B, HW, G, 1e-6, int(swish)
    a.sums, a.acc1, a.acc2, a.drop_p = sums.data_ptr(), 1, 0, 0.0
    dsum_bn = torch.zeros((B, C1 + 5), device="cuda")
    dsum_n = torch.zeros(C1, device="cuda")
    if C2 == 0:                                   # single source: per-(sample, channel) sums of dX in closed form
        a.dsum_bn, a.dsum_stride, a.dsum_n = dsum_bn.data_ptr(), C1 + 5, dsum_n.data_ptr()
    _check(lib.ctdd_unet_gn_bwd(C.byref(a), _stream()), lib)
    torch.cuda.synchronize()
    xr = xs.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    y = F.group_norm(xr.permute(0, 2, 1), G, gr, br, eps=1e-6).permute(0, 2, 1)
    if swish:
        y = y * torch.sigmoid(y)
    y.backward(da.float())
    tol = 2e-5 if f32 else 1.5e-2
    ref1 = xr.grad[..., :C1] + old1.float()
    assert (d1.float() - ref1).abs().max().item() < tol * ref1.abs().max().item()
    if C2 == 0:
        want = xr.grad.sum(1)
        assert (dsum_bn[:, :C1] - want).abs().max().item() < 3e-4 * xr.grad.abs().sum(1).max().item()
        assert (dsum_n - want.sum(0)).abs().max().item() < 3e-4 * xr.grad.abs().sum((0, 1)).max().item()
        assert dsum_bn[:, C1:].abs().max().item() == 0
    if C2:
        assert (d2.float() - xr.grad[..., C1:]).abs().max().item() < tol * xr.grad.abs().max().item()
    np.testing.assert_allclose(sums[..., 0].sum(0).cpu().numpy(), br.grad.cpu().numpy(), rtol=2e-4, atol=2e-4 * br.grad.abs().max().item())
    np.testing.assert_allclose(sums[..., 1].sum(0).cpu().numpy(), gr.grad.cpu().numpy(), rtol=2e-4, atol=2e-4 * gr.grad.abs().max().item())


def test_dropout_mask_is_shared_by_forward_and_backward():
    from ctdd import unet_train as ut
    lib = ut.lib()
    B, HW, Cc, p = 2, 64, 96, 0.25
    x = torch.ones((B, HW, Cc), device="cuda")
    rng = torch.tensor([12345, 7], dtype=torch.int64, device="cuda")
    _check(lib.ctdd_unet_dropout(x.data_ptr(), None, x.numel(), p, rng.data_ptr(), 3, _stream()), lib)
    keep = x != 0
    assert abs(keep.float().mean().item() - (1 - p)) < 0.02 and torch.allclose(x[keep], torch.tensor(1 / (1 - p), device="cuda"))
    # backward through GN (identity affine, no swish) with the same (rng, layer): dz is masked where the forward dropped
    xin = torch.randn((B, HW, Cc), device="cuda")
    da = torch.ones((B, HW, Cc), device="cuda")
    stats = torch.stack([xin.double().sum(1), (xin.double() ** 2).sum(1)], -1).contiguous()
    sums = torch.zeros((B, Cc, 2), device="cuda")
    d1 = torch.zeros_like(xin)
    gamma, beta = torch.ones(Cc, device="cuda"), torch.zeros(Cc, device="cuda")
    a = ut._GnBwdArgs()
    a.s1_f32, a.st1, a.C1, a.gamma, a.beta = xin.data_ptr(), stats.data_ptr(), Cc, gamma.data_ptr(), beta.data_ptr()
    a.B, a.HW, a.G, a.eps, a.swish, a.da_f32, a.sums, a.d1_f32 = B, HW, 24, 1e-6, 0, da.data_ptr(), sums.data_ptr(), d1.data_ptr()
    a.drop_p, a.rng, a.layer = p, rng.data_ptr(), 3
    _check(lib.ctdd_unet_gn_bwd(C.byref(a), _stream()), lib)
    torch.cuda.synchronize()
    want = (keep.float() / (1 - p)).sum(1)                                   # sum over pixels of dz = mask / (1 - p)
    torch.testing.assert_close(sums[..., 0], want, rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("B,T,Cc,heads", [(3, 49, 64, 4), (2, 49, 192, 8), (2, 50, 48, 2), (2, 16, 192, 1)])
def test_attention_backward(B, T, Cc, heads):
    """ctdd_unet_attention_bwd (k_attn_small_bwd_t4 where its tables fit the LDS, the thread-per-score kernel otherwise) against autograd
    through the reference's attention; token counts off the 4 x 4 tile size included."""
    from ctdd import unet_train as ut
    lib = ut.lib()
    g = torch.Generator(device="cuda").manual_seed(0)
    qkv = torch.randn((B, T, 3 * Cc), generator=g, device="cuda")
    do = torch.randn((B, T, Cc), generator=g, device="cuda")
    dq = torch.zeros_like(qkv)
    a = ut._AttnBwdArgs()
    a.qkv, a.d_out_f32, a.B, a.T, a.C, a.heads, a.d_qkv = qkv.data_ptr(), do.data_ptr(), B, T, Cc, heads, dq.data_ptr()
    _check(lib.ctdd_unet_attention_bwd(C.byref(a), _stream()), lib)
    torch.cuda.synchronize()
    qr = qkv.clone().requires_grad_(True)
    x = qr.permute(0, 2, 1).reshape(B * heads, 3 * Cc // heads, T)           # the reference's head split (unet.py:176-200)
    ch = Cc // heads
    q, k, v = torch.split(x, ch, dim=1)
    s = 1 / np.sqrt(np.sqrt(ch))
    w = torch.softmax(torch.einsum("bct,bcs->bts", q * s, k * s), dim=-1)
    out = torch.einsum("bts,bcs->bct", w, v).reshape(B, Cc, T).permute(0, 2, 1)
    out.backward(do)
    assert (dq - qr.grad).abs().max().item() < 2e-5 * qr.grad.abs().max().item()


# ---------------------------------------------------------------------------------------------- whole networks
def _tiny_model(golden, tag):
    import lib.models.models  # noqa: F401
    import lib.models.model_utils as mu
    from config.mnist_config.config_tauUnet_mnist import get_config
    g = golden("unet")
    meta = ast.literal_eval(str(g[f"{tag}__cfg"]))
    cfg = get_config()
    Cn, H, W = meta["data_shape"]
    cfg.data.S, cfg.data.image_size, cfg.data.shape = meta["S"], H, [Cn, H, W]
    cfg.model.update(ch=meta["ch"], ch_mult=meta["ch_mult"], num_res_blocks=meta["n_res_blocks"], num_heads=meta["num_heads"],
                     input_channels=Cn, data_min_max=meta["x_min_max"], model_output=meta["model_output"],
                     attn_resolutions=[int(meta["ch"] / 2)], concat_dim=Cn * H * W, dropout=0.0)
    model = mu.create_model(cfg, torch.device("cuda"))
    pre = f"{tag}__sd__"
    sd = {k[len(pre):]: T(v).cuda() for k, v in g.items() if k.startswith(pre)}
    missing, unexpected = torch.nn.Module.load_state_dict(model, sd, strict=False)
    assert not missing and not unexpected
    model.init_ema()
    return cfg, model, T(g[f"{tag}__x"]).cuda(), T(g[f"{tag}__t"]).cuda()


def _grads(model, cfg, x, t, weight, engine):
    cfg.model.engine = engine
    for p in model.parameters():
        p.grad = None
    logits = model(x.long(), t)
    (logits * weight).sum().backward()
    out = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
    return logits.detach(), out


def _compare(ga, gr, tol_rel, what):
    assert set(ga) == set(gr), set(gr) ^ set(ga)
    worst = 0.0
    for n in gr:
        ref, got = gr[n], ga[n]
        den = ref.abs().max().item()
        err = (got - ref).abs().max().item() / max(den, 1e-12)
        worst = max(worst, err)
        assert got.shape == ref.shape and err < tol_rel, (what, n, err, den)
    return worst


@pytest.mark.parametrize("tag", ["logits", "logistic"])
@pytest.mark.parametrize("precision,tol", [("fp32", 1e-4), ("bf16", 6e-2)])
def test_parameter_gradients_tiny_golden_net(golden, tag, precision, tol):
    """Every parameter gradient of the reference's golden tiny U-Nets (weights from tests/golden/unet.npz) against torch autograd
    of the module: 1e-4 of each tensor's largest gradient in the fp32 mode, the bf16 mode's own bar next to it."""
    cfg, model, x, t = _tiny_model(golden, tag)
    cfg.model.engine_precision = precision
    g = torch.Generator(device="cuda").manual_seed(4)
    B = x.shape[0]
    weight = torch.randn((B, cfg.model.concat_dim, cfg.data.S), generator=g, device="cuda")
    if tag == "logistic":                       # the saturated bins' log(1e-6 + noise) terms (test_gpu_unet) carry no usable gradient
        weight = weight * (torch.rand(weight.shape, generator=g, device="cuda") < 0.3)
    with torch.no_grad():                       # the reference zero-initialises two convolutions' scale: re-draw so every gradient is live
        gg = torch.Generator(device="cuda").manual_seed(5)
        for n_, p in model.named_parameters():
            if p.dim() > 1 and p.abs().max().item() < 1e-6:
                p.copy_(torch.randn(p.shape, generator=gg, device="cuda") / (p[0].numel() ** 0.5))
    lr, gr = _grads(model, cfg, x, t, weight, "torch")
    la, ga = _grads(model, cfg, x, t, weight, "hip")
    assert model._engine is not None and getattr(model._engine, "_train_plans", None), "the training plan did not run"
    assert (la - lr).abs().max().item() < (2e-4 if precision == "fp32" else 5e-2) * max(lr.abs().max().item(), 1.0)
    _compare(ga, gr, tol, (tag, precision))


@pytest.mark.parametrize("precision,tol", [("fp32", 1e-4), ("bf16", 6e-2)])
def test_parameter_gradients_mnist_size(precision, tol):
    """config_tauUnet_mnist (14.0 M parameters, 28x28, ch 96 / 192), batch 4: all parameter gradients vs autograd."""
    import lib.models.models  # noqa: F401
    import lib.models.model_utils as mu
    from config.mnist_config.config_tauUnet_mnist import get_config
    cfg = get_config()
    cfg.model.dropout = 0.0
    cfg.model.engine_precision = precision
    torch.manual_seed(0)
    model = mu.create_model(cfg, torch.device("cuda"))
    g = torch.Generator(device="cuda").manual_seed(1)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if p.dim() > 1:
                p.copy_(torch.randn(p.shape, generator=g, device="cuda") / (p[0].numel() ** 0.5))
            elif name.endswith("bias"):
                p.copy_(0.1 * torch.randn(p.shape, generator=g, device="cuda"))
    model.init_ema()
    B = 4
    x = torch.randint(0, 256, (B, 784), generator=g, device="cuda")
    t = torch.tensor([0.03, 0.4, 0.7, 0.99], device="cuda")
    weight = torch.randn((B, 784, 256), generator=g, device="cuda")
    lr, gr = _grads(model, cfg, x, t, weight, "torch")
    la, ga = _grads(model, cfg, x, t, weight, "hip")
    assert getattr(model._engine, "_train_plans", None)
    assert (la - lr).abs().max().item() < (2e-4 if precision == "fp32" else 5e-2) * lr.abs().max().item()
    worst = _compare(ga, gr, tol, ("mnist", precision))
    print(f"mnist-size parameter gradients, {precision}: worst relative error {worst:.2e}")


def test_train_step_through_the_engine_matches_torch_step():
    """Standard.step (zero_grad -> CT-ELBO -> backward -> clip -> Adam -> EMA) with the network on the training plan against
    the same step on torch autograd ops: same loss, same updated weights (fp32 mode, dropout off)."""
    import copy
    import lib.models.models  # noqa: F401
    import lib.models.model_utils as mu
    import lib.losses.losses  # noqa: F401
    import lib.losses.losses_utils as lu
    import lib.training.training  # noqa: F401
    import lib.training.training_utils as tu
    import lib.optimizers.optimizers  # noqa: F401
    import lib.optimizers.optimizers_utils as ou
    from config.mnist_config.config_tauUnet_mnist import get_config
    cfg = get_config()
    cfg.model.dropout, cfg.model.engine_precision = 0.0, "fp32"
    cfg.model.update(ch=32, ch_mult=[1, 2, 2], num_res_blocks=1, attn_resolutions=[48])
    torch.manual_seed(0)
    model = mu.create_model(cfg, torch.device("cuda"))
    g = torch.Generator(device="cuda").manual_seed(1)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if p.dim() > 1 and p.abs().max().item() < 1e-6:
                p.copy_(torch.randn(p.shape, generator=g, device="cuda") / (p[0].numel() ** 0.5))
    model.init_ema()
    ref = copy.deepcopy(model)
    ref.cfg = copy.deepcopy(cfg)
    ref.cfg.model.engine = "torch"
    ref.shadow_params = [s.clone() for s in model.shadow_params]
    ref._engine = None
    mb = torch.randint(0, 256, (4, 1, 28, 28), generator=g, device="cuda")
    outs = []
    for m_, c_ in ((model, cfg), (ref, ref.cfg)):
        state = {"model": m_, "optimizer": ou.get_optimizer(m_.parameters(), c_), "n_iter": 0}
        step, loss = tu.get_train_step(c_), lu.get_loss(c_)
        ls = []
        for it in range(2):
            torch.manual_seed(100 + it)
            ls.append(float(step.step(state, loss, mb)))
            state["n_iter"] += 1
        outs.append(ls)
    assert getattr(model._engine, "_train_plans", None) and ref._engine is None
    np.testing.assert_allclose(outs[0], outs[1], rtol=2e-4)
    lr = cfg.optimizer.lr
    for (n, a), b in zip(model.named_parameters(), ref.parameters()):
        # Adam's first steps move every weight by ~lr whatever the gradient's size: an entry whose gradient is rounding noise
        # can take the other sign, so isolated differences up to 2 steps x lr are allowed, everything else agrees closely
        d = (a - b).abs()
        assert d.max().item() <= 2.2 * 2 * lr, (n, d.max().item())
        assert (d > 2e-5 + 1e-3 * b.abs()).float().mean().item() < 2e-3, (n, (d > 2e-5 + 1e-3 * b.abs()).float().mean().item())
