"""GPU: the plain bf16 GEMM kernel (csrc/gemm_kernels.hip, `ctdd_gemm_bf16`) behind the hollow transformer's linear layers
(reference: every nn.Linear of lib/networks/hollow_networks.py:90-447) against torch on the same bf16-rounded operands:
every tile / chunk regime of the launcher, ragged M and N, one and three A segments, bias / activation / residual, and the
hi + lo split outputs of the inference engine."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(M, K, N, nseg=1, bias=True, act=0, res=False, f32=True, hi=False, lo=False, seed=0):
    from ctdd import hollow_train as ht
    from ctdd.hollow_engine import _GemmArgs
    g = torch.Generator().manual_seed(seed)
    xs = [torch.randn(M, K, generator=g).cuda().to(torch.bfloat16) for _ in range(nseg)]
    w = (torch.randn(N, nseg * K, generator=g) / (nseg * K) ** 0.5).cuda().to(torch.bfloat16)
    b = torch.randn(N, generator=g).cuda() if bias else None
    r = torch.randn(M, N, generator=g).cuda() if res else None
    out = torch.full((M, N), float("nan"), device="cuda") if f32 else None
    oh = torch.zeros((M, N), device="cuda", dtype=torch.bfloat16) if hi else None
    ol = torch.zeros((M, N), device="cuda", dtype=torch.bfloat16) if lo else None
    a = _GemmArgs()
    for i in range(nseg):
        a.a[i] = xs[i].data_ptr()
    a.nseg, a.w, a.bias, a.res = nseg, w.data_ptr(), ht._p(b), ht._p(r)
    a.out_f32, a.out_hi, a.out_lo, a.M, a.N, a.K, a.act = ht._p(out), ht._p(oh), ht._p(ol), M, N, K, act
    ht._ck(ht.lib().ctdd_gemm_bf16(C.byref(a), torch.cuda.current_stream().cuda_stream), "ctdd_gemm_bf16")
    ref = torch.cat([x.float() for x in xs], dim=1) @ w.float().t()
    if bias:
        ref = ref + b
    if act == 1:
        ref = torch.relu(ref)
    elif act == 2:
        ref = torch.nn.functional.gelu(ref)
    if res:
        ref = ref + r
    return ref, out, oh, ol


# (M, K, N, nseg): N <= 64 | N <= 128 with even / odd 64-chunk counts | K' <= 128 wide N | K' <= 256 | K' >= 512 (both chunk sizes)
SHAPES = [(1000, 64, 8, 1), (300, 128, 136 - 8, 1), (257, 192, 128, 1), (777, 128, 384, 1), (640, 128, 1024, 1), (513, 256, 512, 1),
          (900, 512, 256, 1), (333, 192, 264, 3), (1100, 256, 768, 3), (65, 64, 72, 3)]


@pytest.mark.parametrize("M,K,N,nseg", SHAPES)
def test_gemm_matches_torch(M, K, N, nseg):
    ref, out, _, _ = _run(M, K, N, nseg, seed=M + N)
    assert torch.isfinite(out).all()                                        # every element written (no tile skipped)
    scale = float(ref.abs().max())
    assert float((out - ref).abs().max()) < 2e-5 * scale * (nseg * K) ** 0.5 + 1e-6     # fp32 accumulation, another summation order


@pytest.mark.parametrize("act,res", [(1, False), (2, True), (0, True)])
def test_gemm_epilogue_variants(act, res):
    ref, out, oh, ol = _run(700, 128, 200, 1, bias=True, act=act, res=res, f32=True, hi=True, lo=True, seed=act)
    scale = float(ref.abs().max())
    assert float((out - ref).abs().max()) < 1e-4 * scale
    assert torch.equal(oh, out.to(torch.bfloat16))                          # hi = bf16(v), lo = bf16(v - hi)
    assert torch.equal(ol, (out - oh.float()).to(torch.bfloat16))


def test_gemm_bf16_only_output_and_no_bias():
    ref, out, oh, _ = _run(450, 256, 96, 1, bias=False, f32=False, hi=True)
    assert out is None
    assert float((oh.float() - ref).abs().max()) < 1e-2 * float(ref.abs().max())


def test_gemm_rejects_bad_shapes():
    from ctdd import native
    with pytest.raises(native.CtddError):
        _run(64, 48, 64)                                                    # K % 64 != 0
    with pytest.raises(native.CtddError):
        _run(64, 64, 6)                                                     # N % 4 != 0


@pytest.mark.parametrize("M,K,N", [(700, 128, 1024), (513, 1024, 128), (257, 192, 264)])
def test_gemm_training_epilogues_match_the_separate_passes(M, K, N):
    """drop_p / rng / layer: out = dropout(act(.)) + res with the keep flags of ctdd_hollow_dropout (Philox(seed, step * 4096 +
    layer, element / 4)): bit-identical to the plain product followed by that pass.  mask_u: the ReLU + dropout backward of
    ctdd_hollow_relu_bf16 (saved output as the mask) against the same rule in torch."""
    from ctdd import hollow_train as ht
    from ctdd.hollow_engine import _GemmArgs
    g = torch.Generator().manual_seed(M)
    x = torch.randn(M, K, generator=g).cuda().to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda().to(torch.bfloat16)
    b = torch.randn(N, generator=g).cuda()
    r = torch.randn(M, N, generator=g).cuda()
    rng = torch.tensor([1234, 7], dtype=torch.int64, device="cuda")
    p, layer = 0.25, 5
    st = torch.cuda.current_stream().cuda_stream

    def gemm(res, drop, mask=None, act=0):
        out = torch.full((M, N), float("nan"), device="cuda")
        a = _GemmArgs()
        a.a[0], a.nseg, a.w, a.bias, a.res = x.data_ptr(), 1, w.data_ptr(), b.data_ptr(), ht._p(res)
        a.out_f32, a.M, a.N, a.K, a.act = out.data_ptr(), M, N, K, act
        if drop:
            a.drop_p, a.rng, a.layer = p, (None if mask is not None else rng.data_ptr()), layer
        a.mask_u = ht._p(mask)
        ht._ck(ht.lib().ctdd_gemm_bf16(C.byref(a), st), "ctdd_gemm_bf16")
        return out

    plain = gemm(None, False, act=1)
    fused = gemm(r, True, act=1)
    sep, _ = ht._dropout(plain, p, rng, layer, res=r)                       # the pass the epilogue replaces
    assert torch.equal(fused, sep)
    kept = (fused - r) != 0
    assert 0.6 < float(kept.float().mean()) / float((plain != 0).float().mean()) < 0.9      # ~ 1 - p of the non-zero entries survive
    u = torch.relu(torch.randn(M, N, generator=g)).cuda().to(torch.bfloat16)             # a saved forward output: zeros = dropped / clipped
    got = gemm(None, True, mask=u)
    ref = torch.where(u != 0, gemm(None, False) / (1.0 - p), torch.zeros((), device="cuda"))
    assert torch.allclose(got, ref, rtol=1e-6, atol=0)


@pytest.mark.parametrize("M,K,N", [(700, 128, 384), (513, 256, 1024), (300, 1024, 256), (257, 128, 128), (640, 256, 72)])
def test_gemm_split_product_kernel(M, K, N):
    """The hi / lo split product [x_hi | x_lo | x_hi] . [w_hi | w_hi | w_lo] (first and third segment the same tensor): the launcher
    stages its four distinct operand tiles once per K chunk (k_gemm_bf16<..., SPLIT>); N <= 64 keeps the three-segment kernel.
    Against x_hi w_hi + x_lo w_hi + x_hi w_lo in float64, and: the result is within 2^-15 of the fp32 product of the unsplit operands."""
    from ctdd import hollow_train as ht
    from ctdd.hollow_engine import _GemmArgs
    g = torch.Generator().manual_seed(M + K)
    x = torch.randn(M, K, generator=g).cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    b = torch.randn(N, generator=g).cuda()
    xh = x.to(torch.bfloat16); xl = (x - xh.float()).to(torch.bfloat16)
    wh = w.to(torch.bfloat16); wl = (w - wh.float()).to(torch.bfloat16)
    wcat = torch.cat([wh, wh, wl], dim=1).contiguous()
    out = torch.full((M, N), float("nan"), device="cuda")
    a = _GemmArgs()
    a.a[0], a.a[1], a.a[2] = xh.data_ptr(), xl.data_ptr(), xh.data_ptr()
    a.nseg, a.w, a.bias, a.out_f32, a.M, a.N, a.K = 3, wcat.data_ptr(), b.data_ptr(), out.data_ptr(), M, N, K
    ht._ck(ht.lib().ctdd_gemm_bf16(C.byref(a), torch.cuda.current_stream().cuda_stream), "ctdd_gemm_bf16")
    ref = (xh.double() @ wh.double().t() + xl.double() @ wh.double().t() + xh.double() @ wl.double().t() + b.double())
    assert torch.isfinite(out).all()
    scale = float(ref.abs().max())
    assert float((out.double() - ref).abs().max()) < 2e-6 * scale * K ** 0.5
    full = x.double() @ w.double().t() + b.double()
    assert float((out.double() - full).abs().max()) < 2.0 ** -15 * scale * 4
