"""the hollow blocks' forward / data-gradient GEMMs (rows = 28800, bf16 operands), graph-timed."""
import sys, os
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_R, os.path.join(_R, 'continuous-time-diffusion-models-for-discrete-data_amd')]
import torch
from bench_kernels import timeit
from ctdd import hollow_train as ht
R = 28800
dev = torch.device("cuda")
cases = [("qkv fwd", 128, 384, True, False, 0, False), ("out fwd (+res)", 128, 128, True, False, 0, True), ("fc1 fwd relu, bf16 out", 128, 1024, False, True, 1, False),
         ("fc2 fwd (+res)", 1024, 128, True, False, 0, True), ("datt dgrad", 128, 128, True, False, 0, False), ("dz(in) dgrad", 384, 128, True, False, 0, False),
         ("du dgrad bf16 out", 128, 1024, False, True, 0, False), ("dz(fc1) dgrad", 1024, 128, True, False, 0, False)]
for name, K, N, f32, hi, act, res in cases:
    x = torch.randn((R, K), device=dev).to(torch.bfloat16); w = torch.randn((N, K), device=dev).to(torch.bfloat16)
    b = torch.randn((N,), device=dev); r = torch.randn((R, N), device=dev) if res else None
    t = timeit(lambda: ht._gemm(x, w, b, r, R, K, N, True, act=act, want_hi=hi, want_f32=f32), 20)
    byts = R * K * 2 + N * K * 2 + (R * N * 4 if f32 else 0) + (R * N * 2 if hi else 0) + (R * N * 4 if res else 0)
    print(f"{name:26s} K={K:5d} N={N:5d}: {t*1e6:7.1f} us  {2.0*R*K*N/t/1e12:6.1f} TF/s  {byts/t/1e9:7.0f} GB/s ({byts/1e6:.0f} MB)", flush=True)
