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
