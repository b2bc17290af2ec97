"""GPU: the hand-written U-Net inference engine (csrc/unet_kernels.hip + ctdd/unet_engine.py)
against golden logits frozen from the reference (tests/golden/unet.npz) and against the
differentiable module forward on the MNIST-sized network."""
import ast

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

T = torch.from_numpy


def _tiny_model(golden, tag):
    import lib.models.models  # noqa: F401
    import lib.models.model_utils as mu
    from config.mnist_config.config_tauUnet_mnist import get_config
    g = golden("unet")
    meta = ast.literal_eval(str(g[f"{tag}__cfg"]))
    cfg = get_config()
    C, H, W = meta["data_shape"]
    cfg.data.S, cfg.data.image_size, cfg.data.shape = meta["S"], H, [C, H, W]
    cfg.model.update(ch=meta["ch"], ch_mult=meta["ch_mult"], num_res_blocks=meta["n_res_blocks"], num_heads=meta["num_heads"],
                     input_channels=C, data_min_max=meta["x_min_max"], model_output=meta["model_output"],
                     attn_resolutions=[int(meta["ch"] / 2)], concat_dim=C * H * W)
    model = mu.create_model(cfg, torch.device("cuda"))
    pre = f"{tag}__sd__"
    sd = {k[len(pre):]: T(v).cuda() for k, v in g.items() if k.startswith(pre)}
    missing, unexpected = torch.nn.Module.load_state_dict(model, sd, strict=False)
    assert not missing and not unexpected
    model.init_ema()
    model.eval()
    return cfg, model, T(g[f"{tag}__x"]).cuda(), T(g[f"{tag}__t"]).cuda(), g[f"{tag}__out"]


@pytest.mark.parametrize("tag", ["logits", "logistic"])
def test_engine_matches_reference_golden(golden, tag):
    from ctdd.unet_engine import UNetEngine
    cfg, model, x, t, ref = _tiny_model(golden, tag)
    with torch.no_grad():
        eng = UNetEngine(model, precision="fp32")
        C, H, W = cfg.data.shape
        out = eng(x.view(-1, C, H, W), t).cpu().numpy()
        # BASELINE bar: logits within 1e-4 (fp32) of the reference.  The logistic head's saturated
        # bins are log1p(-exp(cl - cr) + 1e-6) with exp(..) -> 1: the value is ~log(1e-6 + d) where d
        # is fp32 rounding noise of the exp, so a 1-ulp libm difference moves it by up to ~0.06 on
        # ANY two fp32 implementations (ill-conditioned formula, models.py:273-279); in general the
        # logit error is ~1e-7 / p.  Bins with p < 1e-3 (logit < -7) are therefore checked through the
        # probabilities instead.
        well = ref > -7.0
        np.testing.assert_allclose(out[well], ref[well], rtol=0, atol=1e-4)
        assert well.mean() > 0.25
        np.testing.assert_allclose(torch.softmax(torch.from_numpy(out), -1).numpy(),
                                   torch.softmax(torch.from_numpy(ref), -1).numpy(), rtol=0, atol=2e-5)   # = 1e-4 on a logit at p <= 0.2
        fast = UNetEngine(model, precision="bf16")(x.view(-1, C, H, W), t).cpu().numpy()
        # bf16 activations/weights (the BASELINE config's dtype): reported separately, looser bar
        err = np.abs(fast - ref).max() / np.abs(ref).max()
        assert err < 5e-2, err


def test_engine_matches_module_mnist():
    """MNIST tauLDR U-Net (14.0 M parameters), re-drawn weights, batch 3: engine vs autograd module."""
    import lib.models.models  # noqa: F401
    import lib.models.model_utils as mu
    from config.mnist_config.config_tauUnet_mnist import get_config
    from ctdd.unet_engine import UNetEngine
    cfg = get_config()
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
    model.eval()
    x = torch.randint(0, 256, (3, 784), device="cuda")
    t = torch.tensor([0.03, 0.5, 0.99], device="cuda")
    with torch.no_grad():
        cfg.model.engine = "torch"
        ref = model(x, t).cpu()
        out = UNetEngine(model, precision="fp32")(x.view(3, 1, 28, 28), t).cpu()
        fast = UNetEngine(model, precision="bf16")(x.view(3, 1, 28, 28), t).cpu()
    assert ref.shape == out.shape == (3, 784, 256)
    scale = ref.abs().max().item()
    assert scale > 0.5
    assert (out - ref).abs().max().item() < 2e-4 * max(scale, 1.0)
    assert (fast - ref).abs().max().item() < 5e-2 * scale
    model.train()


def _mnist_model(seed=0, **model_over):
    import lib.models.models  # noqa: F401
    import lib.models.model_utils as mu
    from config.mnist_config.config_tauUnet_mnist import get_config
    cfg = get_config()
    cfg.model.update(model_over)
    torch.manual_seed(seed)
    model = mu.create_model(cfg, torch.device("cuda"))
    g = torch.Generator(device="cuda").manual_seed(seed + 1)
    with torch.no_grad():                           # (the reference zero-initialises the output conv: re-draw so logits vary)
        for name, p in model.named_parameters():
            if p.dim() > 1:
                p.copy_(torch.randn(p.shape, generator=g, device="cuda") / (p[0].numel() ** 0.5))
    model.init_ema()
    return cfg, model


def test_eval_forwards_do_not_alias():
    """Two eval/no_grad forwards of the same batch size return distinct tensors (the engine's plan owns ONE output buffer;
    the wrapper hands out copies unless a sampler loop borrows it), and the two-forward-pass CT-ELBO in eval mode
    (x_logits = model(x_t), then logits at x~) equals the torch-op model's value."""
    import lib.losses.losses  # noqa: F401
    import lib.losses.losses_utils as lu
    from lib.models.models import borrow_engine_output
    cfg, model = _mnist_model()
    model.eval()
    xa = torch.randint(0, 256, (4, 784), device="cuda")
    xb = torch.randint(0, 256, (4, 784), device="cuda")
    t = torch.tensor([0.1, 0.4, 0.7, 0.95], device="cuda")
    with torch.no_grad():
        a = model(xa, t)
        a0 = a.clone()
        b = model(xb, t)
        assert a.data_ptr() != b.data_ptr() and torch.equal(a, a0) and not torch.equal(a, b)
        with borrow_engine_output(model):           # the sampler loops: zero-copy, next forward overwrites
            c = model(xa, t)
            d = model(xb, t)
            assert c.data_ptr() == d.data_ptr()
        assert not getattr(model, "_borrow_engine_output", False)
        cfg.loss.one_forward_pass = False
        loss = lu.get_loss(cfg)
        state = {"model": model, "n_iter": 0}
        mb = torch.randint(0, 256, (4, 1, 28, 28), device="cuda")
        torch.manual_seed(5)
        l_eng = loss.calc_loss(state, mb, None).item()
        cfg.model.engine = "torch"
        torch.manual_seed(5)
        l_ref = loss.calc_loss(state, mb, None).item()
        cfg.model.engine = "hip"
    assert np.isfinite(l_ref) and abs(l_eng - l_ref) < 2e-2 * abs(l_ref) + 1e-3, (l_eng, l_ref)     # bf16 engine vs fp32 ops
    model.train()


def test_engine_sub_batch_streams_match_single_plan():
    """The headline path: a batch replayed as two HIP-graph sub-batches on parallel streams into slices of one logits
    buffer (engine_streams = 2) against one full-batch plan (engine_streams = 1), over consecutive calls with new inputs."""
    cfg, model = _mnist_model(seed=3)
    model.eval()
    _sub_batch_streams_case(cfg, model, 64)


def test_engine_sub_batch_streams_at_bench_size():
    """The same at the size bench.py runs: 256 samples = two 128-sample plans (the kernel selection of the headline)."""
    cfg, model = _mnist_model(seed=4)
    model.eval()
    _sub_batch_streams_case(cfg, model, 256, iters=3)


def _sub_batch_streams_case(cfg, model, B, iters=4):
    from ctdd.unet_engine import UNetEngine
    with torch.no_grad():
        cfg.model.engine_streams = 2
        e2 = UNetEngine(model, precision="bf16")
        cfg.model.engine_streams = 1
        e1 = UNetEngine(model, precision="bf16")
        for it in range(iters):
            x = torch.randint(0, 256, (B, 1, 28, 28), device="cuda")
            t = torch.rand(B, device="cuda") * 0.98 + 0.01
            cfg.model.engine_streams = 2
            o2 = e2(x, t).clone()
            cfg.model.engine_streams = 1
            o1 = e1(x, t).clone()
            assert any(len(k) == 5 for k in e2._plans) and all(len(k) == 4 for k in e1._plans)    # (B, dtype, streams, uniform t, bf16 logits) / (B, dtype, uniform t, bf16 logits)
            cfg.model.engine = "torch"
            ref = model(x.view(B, -1), t)
            cfg.model.engine = "hip"
            scale = ref.abs().max().item()
            # (kernel selection depends on the batch a plan runs -- 32 vs 64 samples here -- so bf16 roundings differ
            # between the two plans; both are held to the bf16 mode's bar against the fp32 module)
            assert (o2 - ref).abs().max().item() < 5e-2 * scale and (o1 - ref).abs().max().item() < 5e-2 * scale
            assert (o1 - o2).abs().max().item() < 5e-2 * scale
    cfg.model.engine_streams = 2
    model.train()


def test_engine_behind_ddp_wrapper():
    """cfg.distributed wraps the network in DistributedDataParallel (models.py:104-107); eval-mode sampling still runs the
    HIP engine on the module behind the wrapper."""
    import os
    import torch.distributed as dist
    import lib.models.models  # noqa: F401
    import lib.models.model_utils as mu
    from config.mnist_config.config_tauUnet_mnist import get_config
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    own = not dist.is_initialized()
    if own:
        dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        cfg = get_config()
        cfg.distributed = True
        torch.manual_seed(0)
        model = mu.create_model(cfg, torch.device("cuda"), rank=0)
        assert model.net.__class__.__name__ == "DistributedDataParallel"
        model.eval()
        x = torch.randint(0, 256, (2, 784), device="cuda")
        t = torch.tensor([0.2, 0.8], device="cuda")
        with torch.no_grad():
            out = model(x, t)
            assert model._engine is not None and model._engine._plans
            cfg.model.engine = "torch"
            ref = model(x, t)
        assert (out - ref).abs().max().item() < 5e-2 * max(ref.abs().max().item(), 1e-3) + 1e-4
        model.train()
    finally:
        if own:
            dist.destroy_process_group()


def _conv_case(B, H, W, segs, N, ksplit, seed):
    """Random NHWC bf16 segments + [N][K] weights; returns (args builder inputs, fp32 torch reference)."""
    import torch.nn.functional as F
    g = torch.Generator(device="cuda").manual_seed(seed)
    srcs, ws, ref = [], [], 0.0
    for C_, kind in segs:
        x = torch.randn((B, H, W, C_), generator=g, device="cuda").to(torch.bfloat16)
        k = 3 if kind == 0 else 1
        w = (torch.randn((N, C_, k, k), generator=g, device="cuda") / (C_ * k * k) ** 0.5).to(torch.bfloat16)
        ref = ref + F.conv2d(x.float().permute(0, 3, 1, 2), w.float(), padding=k // 2)
        srcs.append(x.contiguous())
        ws.append(w.float().permute(0, 2, 3, 1).reshape(N, -1))
    w2d = torch.cat(ws, 1).to(torch.bfloat16).contiguous()
    bias = torch.randn((N,), generator=g, device="cuda")
    tb = torch.randn((B, N), generator=g, device="cuda")
    res = torch.randn((B, H, W, N), generator=g, device="cuda").to(torch.bfloat16).contiguous()
    ref = ref.permute(0, 2, 3, 1) + bias + tb[:, None, None, :] + res.float()
    return srcs, w2d, bias, tb, res, ref


@pytest.mark.parametrize("B,H,W,segs,N,ksplit,bnt", [
    (5, 28, 28, [(96, 0)], 96, 1, 3),                       # 3x3, ragged last tile (3920 pixels)
    (3, 14, 14, [(192, 0), (96, 1), (96, 1)], 192, 2, 3),   # ResBlock conv2 with the folded 1x1 skip on two sources, split-K
    (4, 7, 7, [(192, 0)], 192, 3, 3),                       # 7x7 level, tiles spanning several samples
    (2, 28, 28, [(96, 0)], 256, 1, 4),                      # output conv shape (N = 256)
    (2, 16, 16, [(64, 0), (64, 1)], 64, 1, 2),
    (2, 28, 28, [(16, 0)], 96, 1, 3),                       # a single unit
    (3, 14, 14, [(48, 1), (32, 0)], 96, 1, 3),              # 1x1 units first, then 3x3
    (5, 28, 28, [(96, 0)], 96, 1, 13),                      # ring only: 256-pixel tiles, two workgroups per CU
    (3, 14, 14, [(192, 0), (96, 1), (96, 1)], 192, 1, 13),
    (4, 7, 7, [(64, 0)], 64, 2, 12),
])
@pytest.mark.parametrize("kernel", ["ctdd_unet_conv_res", "ctdd_unet_conv_ring"])
def test_slab_conv_kernels(kernel, B, H, W, segs, N, ksplit, bnt):
    """The slab convolution kernels (weights of all nine taps resident in LDS; register-staged 32-channel
    units or LDS-DMA ring of 16-channel units) against an fp32 torch convolution of the same bf16-rounded
    operands, with bias + per-sample bias + residual and the GroupNorm statistics."""
    if kernel == "ctdd_unet_conv_res" and (any(c % 32 for c, _ in segs) or bnt > 4):
        pytest.skip("register-staged kernel needs 32-channel units / has no 256-pixel variant")
    import ctypes as C
    from ctdd import unet_engine as ue
    lib = ue._lib()
    srcs, w2d, bias, tb, res, ref = _conv_case(B, H, W, segs, N, ksplit, seed=B * 100 + H)
    M = B * H * W
    a = ue._ConvArgs()
    a.nseg = len(segs)
    for i, ((C_, kind), x) in enumerate(zip(segs, srcs)):
        a.seg[i].hi, a.seg[i].C, a.seg[i].kind = x.data_ptr(), C_, kind
    a.w_hi, a.B, a.H, a.W, a.Hin, a.Win, a.N, a.Ktot = w2d.data_ptr(), B, H, W, H, W, N, w2d.shape[1]
    a.bias, a.tbias, a.tb_stride, a.res_bf16 = bias.data_ptr(), tb.data_ptr(), N, res.data_ptr()
    out = torch.empty((M, N), dtype=torch.float32, device="cuda")
    out_hi = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    stats = torch.zeros((B, N, 2), dtype=torch.float64, device="cuda")
    acc = torch.zeros((M, N), dtype=torch.float32, device="cuda")
    a.out_f32, a.out_hi, a.stats, a.ksplit, a.acc_buf = out.data_ptr(), out_hi.data_ptr(), stats.data_ptr(), ksplit, acc.data_ptr()
    rc = getattr(lib, kernel)(C.byref(a), bnt, torch.cuda.current_stream().cuda_stream)
    assert rc == 0, lib.ctdd_last_error().decode()
    torch.cuda.synchronize()
    ref2 = ref.reshape(M, N)
    # exact products (bf16 x bf16 in fp32), fp32 accumulation in a different order: ~1e-6 relative
    assert (out - ref2).abs().max().item() < 2e-5 * ref2.abs().max().item()
    assert torch.equal(out_hi, out.to(torch.bfloat16))
    want = torch.stack([out.double().view(B, H * W, N).sum(1), (out.double() ** 2).view(B, H * W, N).sum(1)], -1)
    # fp32 sums over blocks of eight rows, fp64 above that: ~1e-7 of sqrt(n * sum of squares)
    torch.testing.assert_close(stats, want, rtol=1e-5, atol=1e-3)


def test_engine_bf16_logits_on_request():
    """borrow_engine_output(model, bf16_logits=True): the output convolution writes the (B, D, S) logits in bf16 -- the same
    values as the fp32 logits rounded once (the two plans share every kernel; GroupNorm statistics meet in atomics, so the fp32
    values themselves may differ in their last bits)."""
    import lib.models.models  # noqa: F401
    import lib.models.model_utils as mu
    from lib.models.models import borrow_engine_output
    from config.mnist_config.config_tauUnet_mnist import get_config
    cfg = get_config()
    cfg.device = "cuda"
    torch.manual_seed(0)
    model = mu.create_model(cfg, torch.device("cuda"))
    model.eval()
    g = torch.Generator(device="cuda").manual_seed(5)
    for B in (3, 64):                                      # one plan / two sub-batch plans on parallel streams
        x = torch.randint(0, 256, (B, 784), device="cuda", generator=g)
        t = torch.rand(B, device="cuda", generator=g) * 0.9 + 0.05
        with torch.no_grad():
            f32 = model(x, t)
            with borrow_engine_output(model, bf16_logits=True):
                b16 = model(x, t)
                assert b16.dtype == torch.bfloat16 and b16.shape == f32.shape
                b16 = b16.float().clone()
            with borrow_engine_output(model):
                again = model(x, t)
                assert again.dtype == torch.float32
        assert f32.dtype == torch.float32
        err = (b16 - f32).abs()
        assert (err <= f32.abs() * 2.0 ** -8 + 1e-6).all(), float((err / f32.abs().clamp_min(1e-3)).max())


def test_engine_uniform_time_plan_matches_the_general_plan():
    """borrow_engine_output(model, uniform_time=True): the one-launch time path for a batch that shares one time value
    (k_time_uniform) gives the logits of the general three-launch path on the same inputs."""
    import lib.models.models  # noqa: F401
    import lib.models.model_utils as mu
    from lib.models.models import borrow_engine_output
    from config.mnist_config.config_tauUnet_mnist import get_config
    cfg = get_config()
    cfg.device = "cuda"
    torch.manual_seed(1)
    model = mu.create_model(cfg, torch.device("cuda"))
    model.eval()
    g = torch.Generator(device="cuda").manual_seed(6)
    for B, tval in ((5, 0.37), (64, 0.93)):
        x = torch.randint(0, 256, (B, 784), device="cuda", generator=g)
        t = torch.full((B,), tval, device="cuda")
        with torch.no_grad():
            general = model(x, t)
            with borrow_engine_output(model, uniform_time=True):
                uni = model(x, t).clone()
        scale = float(general.abs().max())
        # (the time path is fp32 in both plans and differs in summation order only; downstream the bf16 activations amplify
        #  that to a few 1e-3 of the logit range -- the bf16 mode's own bar against the fp32 module is 5e-2)
        assert float((uni - general).abs().max()) < 2e-2 * scale, float((uni - general).abs().max()) / scale


def test_engine_time_table_rows_replace_the_time_path():
    """UNetEngine.time_table: all grid times in one launch; a forward that is handed row i (the samplers' loops do, through
    `model._engine_time_row`) gives the logits of the general plan at time i -- and the table's rows are the projections the
    module itself computes (fp32 on both sides: 1e-5)."""
    import lib.models.models  # noqa: F401
    import lib.models.model_utils as mu
    from lib.models.models import borrow_engine_output, unwrap
    from config.mnist_config.config_tauUnet_mnist import get_config
    cfg = get_config()
    cfg.device = "cuda"
    torch.manual_seed(1)
    model = mu.create_model(cfg, torch.device("cuda"))
    model.eval()
    ts = torch.tensor([0.93, 0.37, 0.011, 1.0], device="cuda")
    with torch.no_grad():
        table = model.engine_time_table(ts)
        net = unwrap(model.net)
        blocks = [mod.resblocks for mod in list(net.down) + list(net.mid) + list(net.up) if hasattr(mod, "resblocks")]
        act = torch.nn.functional.silu(net.time(ts))
        want = torch.cat([rb.time[1](act) for rb in blocks], 1)
    assert table.shape == want.shape
    assert float((table - want).abs().max()) <= 1e-5 * float(want.abs().max())
    g = torch.Generator(device="cuda").manual_seed(8)
    for B, i in ((5, 1), (64, 0)):
        x = torch.randint(0, 256, (B, 784), device="cuda", generator=g)
        t = torch.full((B,), float(ts[i]), device="cuda")
        with torch.no_grad():
            general = model(x, t)
            with borrow_engine_output(model, uniform_time=True):
                model._engine_time_row = table[i]
                try:
                    row = model(x, t).clone()
                finally:
                    model._engine_time_row = None
        scale = float(general.abs().max())
        assert float((row - general).abs().max()) < 2e-2 * scale, float((row - general).abs().max()) / scale    # (as the uniform-time plan)
    # a plan built for rows refuses a call without one instead of running on a stale row
    from ctdd import native
    eng = model._engine
    st = next(v for k, v in eng._plans.items() if "row" in k and not isinstance(v, tuple))
    with pytest.raises(native.CtddError):
        eng._set_time(st, t, None)


@pytest.mark.parametrize("B,H,C1,C2,G,swish,slab", [
    (5, 28, 96, 0, 32, 1, 0), (3, 28, 192, 96, 32, 1, 0), (4, 14, 192, 192, 32, 1, 0), (7, 7, 192, 0, 32, 0, 0),
    (2, 28, 96, 96, 32, 1, 96), (130, 7, 192, 192, 32, 1, 0), (2, 4, 64, 0, 32, 1, 0), (3, 32, 128, 0, 32, 1, 0), (2, 14, 96, 0, 32, 1, 24)])
def test_gn_onepass_matches_groupnorm(B, H, C1, C2, G, swish, slab):
    """ctdd_unet_gn_onepass (statistics + GroupNorm + Swish of bf16 NHWC tensors in one pass, one or two concatenated sources,
    slabs of whole groups -- including groups and slabs that straddle the two sources) against torch's GroupNorm in fp32 on the
    same bf16-rounded values; the result is bf16: 2^-8 relative + the hardware exp2 / rcp of the Swish."""
    import ctypes as C_
    from ctdd import unet_engine as ue
    lib = ue._lib()
    g = torch.Generator(device="cuda").manual_seed(B * 1000 + H * 10 + C1)
    HW, Ct = H * H, C1 + C2
    srcs = [(torch.randn((B, HW, c), device="cuda", generator=g) * 1.7 + 0.4).to(torch.bfloat16) for c in (C1, C2) if c]
    gamma = torch.randn(Ct, device="cuda", generator=g)
    beta = torch.randn(Ct, device="cuda", generator=g)
    out = torch.zeros((B, HW, Ct), dtype=torch.bfloat16, device="cuda")
    a = ue._GnArgs()
    a.s1_bf16, a.C1 = srcs[0].data_ptr(), C1
    if C2:
        a.s2_bf16, a.C2 = srcs[1].data_ptr(), C2
    a.gamma, a.beta, a.B, a.HW, a.G, a.eps, a.swish, a.out_hi = gamma.data_ptr(), beta.data_ptr(), B, HW, G, 1e-5, swish, out.data_ptr()
    # training plans hand the kernel the sources' statistics buffers: it leaves the per-(sample, channel) sums / sums of squares there
    stats = [torch.full((B, c, 2), float("nan"), dtype=torch.float64, device="cuda") for c in (C1, C2) if c]
    a.st1 = stats[0].data_ptr()
    if C2:
        a.st2 = stats[1].data_ptr()
    rc = lib.ctdd_unet_gn_onepass(C_.byref(a), slab, 0, torch.cuda.current_stream().cuda_stream)
    assert rc == 0, lib.ctdd_last_error().decode()
    assert ue._onepass_slab(B, HW, Ct, G) > 0
    for s_, st_ in zip(srcs, stats):
        xs = s_.double()
        ref_st = torch.stack([xs.sum(1), (xs * xs).sum(1)], dim=2)                # (B, c, 2)
        assert torch.allclose(st_, ref_st, rtol=2e-5, atol=1e-3), float((st_ - ref_st).abs().max())
    x = torch.cat([s_.float() for s_ in srcs], 2).permute(0, 2, 1).reshape(B, Ct, H, H)
    want = torch.nn.functional.group_norm(x, G, gamma, beta, 1e-5)
    if swish:
        want = want * torch.sigmoid(want)
    want = want.reshape(B, Ct, HW).permute(0, 2, 1)
    err = (out.float() - want).abs()
    assert float((err - 2.0 ** -7 * want.abs()).max()) <= 2e-3, float(err.max())


def test_gn_onepass_refuses_what_it_cannot_hold():
    import ctypes as C_
    from ctdd import unet_engine as ue
    lib = ue._lib()
    x = torch.zeros((1, 64 * 64, 256), dtype=torch.bfloat16, device="cuda")      # 64x64 x (8 channels per group, slab >= 8): 4096 pixels / 1024 lanes > ... fits; G = 1 does not
    out = torch.empty_like(x)
    w = torch.ones(256, device="cuda")
    a = ue._GnArgs()
    a.s1_bf16, a.C1, a.gamma, a.beta, a.B, a.HW, a.G, a.eps, a.swish, a.out_hi = x.data_ptr(), 256, w.data_ptr(), w.data_ptr(), 1, 4096, 1, 1e-5, 1, out.data_ptr()
    assert lib.ctdd_unet_gn_onepass(C_.byref(a), 0, 0, torch.cuda.current_stream().cuda_stream) != 0      # one group of 256 channels x 4096 pixels: 128 vectors per lane
    assert ue._onepass_slab(1, 4096, 256, 1) == 0


def test_engine_onepass_groupnorm_plans_match_statistics_epilogue_plans():
    """cfg.model.gn_onepass (default on: the 7x7 / 14x14 levels' GroupNorms as one launch with the statistics inside, no
    statistics in their producers' epilogues) against the plan with statistics epilogues + k_gn_apply everywhere, and both
    against the fp32 module at the bf16 bar.  (The one-pass statistics are those of the bf16-rounded tensor the kernel
    normalises; the epilogue statistics those of the fp32 accumulators before rounding.)"""
    import lib.models.models  # noqa: F401
    import lib.models.model_utils as mu
    from ctdd.unet_engine import UNetEngine
    from config.mnist_config.config_tauUnet_mnist import get_config
    cfg = get_config()
    cfg.device = "cuda"
    cfg.model.engine = "torch"
    torch.manual_seed(2)
    model = mu.create_model(cfg, torch.device("cuda"))
    model.eval()
    g = torch.Generator(device="cuda").manual_seed(9)
    x = torch.randint(0, 256, (48, 784), device="cuda", generator=g)
    t = torch.rand(48, device="cuda", generator=g) * 0.98 + 0.01
    with torch.no_grad():
        want = model(x, t)
    outs = {}
    for flag in (1, 0):
        cfg.model.gn_onepass = flag
        eng = UNetEngine(model, precision="bf16")
        with torch.no_grad():
            outs[flag] = eng(x, t).float().clone()
        labels = [getattr(s_, "label", "") or "" for v in eng._plans.values() for s_ in (v.plan if not isinstance(v, tuple) else [p_ for sub in v[1] for p_ in sub.plan])]
        fns = [lb[0] for lb in labels if isinstance(lb, tuple)]                 # (entry point, label) per launch
        n1, n0 = fns.count("ctdd_unet_gn_onepass"), fns.count("ctdd_unet_gn_apply")
        assert (n1 > 0 and n0 > 0) if flag else (n1 == 0 and n0 > 0), (flag, n1, n0)      # small levels one-pass, 28x28 the statistics path
    scale = float(want.abs().max())
    for flag in (1, 0):
        assert float((outs[flag] - want).abs().max()) < 5e-2 * scale
    assert float((outs[1] - outs[0]).abs().max()) < 2e-2 * scale


@pytest.mark.parametrize("B,T,C,heads", [(5, 49, 192, 1), (3, 49, 192, 8), (2, 16, 64, 4), (2, 64, 32, 1), (3, 50, 96, 2)])
def test_mid_attention_kernel(B, T, C, heads):
    """ctdd_unet_attention (k_attn_small_t4: transposed q / k in LDS, 4 x 4 register tiles; token counts that are not multiples of
    four pad the tiles) against the reference's QKVAttention in torch (unet.py:176-200: per-head channel order [q | k | v], both q
    and k scaled by ch^-1/4, softmax over the keys)."""
    import ctypes as C_
    from ctdd import unet_engine as ue
    lib = ue._lib()
    g = torch.Generator(device="cuda").manual_seed(T * 100 + C + heads)
    qkv = torch.randn((B, T, 3 * C), device="cuda", generator=g)
    out = torch.full((B * T, C), float("nan"), device="cuda")
    a = ue._AttnArgs()
    a.qkv, a.B, a.T, a.C, a.heads, a.out_f32 = qkv.data_ptr(), B, T, C, heads, out.data_ptr()
    assert lib.ctdd_unet_attention(C_.byref(a), torch.cuda.current_stream().cuda_stream) == 0, lib.ctdd_last_error().decode()
    ch = C // heads
    x = qkv.view(B, T, heads, 3, ch).permute(0, 2, 3, 1, 4)              # (B, heads, 3, T, ch)
    q, k, v = x[:, :, 0], x[:, :, 1], x[:, :, 2]
    sc = ch ** -0.25
    w = torch.softmax(torch.einsum("bhtc,bhsc->bhts", q * sc, k * sc), dim=-1)
    ref = torch.einsum("bhts,bhsc->bhtc", w, v).permute(0, 2, 1, 3).reshape(B * T, C)
    assert torch.isfinite(out).all()
    assert float((out - ref).abs().max()) < 2e-5 * float(ref.abs().max()) + 1e-6
