"""matrix-core training attention kernels alone, graph-timed, over batch size (latency vs throughput)."""
import sys, os
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_R, os.path.join(_R, 'continuous-time-diffusion-models-for-discrete-data_amd')]
import torch
from bench_kernels import timeit
from ctdd import hollow_train as ht
D, H, hd = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (225, 8, 16)
E = H * hd
for B in ((int(sys.argv[4]),) if len(sys.argv) > 4 else (4, 16, 32, 64, 128, 256)):
    for p in (0.0, 0.1):
        qkv = torch.randn((B * D, 3 * E), device="cuda")
        rng = torch.tensor([5, 9], dtype=torch.int64, device="cuda")
        out, out_hi, stats = ht._attention_fwd(qkv, None, None, B, D, D, H, hd, 0, p, rng if p > 0 else None, 3, True)
        dout = torch.randn((B * D, E), device="cuda")
        tf = timeit(lambda: ht._attention_fwd(qkv, None, None, B, D, D, H, hd, 0, p, rng if p > 0 else None, 3, True), 20)
        tb = timeit(lambda: ht._attention_bwd(qkv, None, None, out, stats, dout, B, D, D, H, hd, 0, p, rng if p > 0 else None, 3, True, want_f32=False), 20)
        wgs = B * H * 2
        print(f"B={B:4d} p={p}: fwd {tf*1e6:7.1f} us  bwd (dQ + dK/dV) {tb*1e6:7.1f} us   ({wgs} workgroups per launch)", flush=True)
