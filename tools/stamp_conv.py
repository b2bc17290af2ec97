"""Phase durations of k_conv_res (stamps build)."""
import os, sys
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_R, os.path.join(_R, "continuous-time-diffusion-models-for-discrete-data_amd")]
import sys, os, ctypes as C
import torch, numpy as np
from ctdd import unet_engine as ue
l = C.CDLL(os.path.join(_R, "continuous-time-diffusion-models-for-discrete-data_amd", "libres_stamps.so"))
l.ctdd_unet_conv_res.argtypes, l.ctdd_unet_conv_res.restype = [C.c_void_p, C.c_int, C.c_void_p], C.c_int
l.ctdd_unet_conv_ring.argtypes, l.ctdd_unet_conv_ring.restype = [C.c_void_p, C.c_int, C.c_void_p], C.c_int
l.ctdd_last_error.restype = C.c_char_p
def run(B, H, W, Cs, N, bnt, kern='ctdd_unet_conv_ring'):
    M = B * H * W
    xs = [torch.randn((M, c), device="cuda").to(torch.bfloat16) for c in Cs]
    K = sum(9 * c for c in Cs)
    w = (torch.randn((N, K), device="cuda") / K ** 0.5).to(torch.bfloat16)
    a = ue._ConvArgs(); a.nseg = len(Cs)
    for i, x in enumerate(xs):
        a.seg[i].hi, a.seg[i].C, a.seg[i].kind = x.data_ptr(), Cs[i], 0
    out = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    nwg = -(-M // 512) * -(-N // (32 * bnt))
    buf = torch.zeros((3 * nwg * 8 * 8,), dtype=torch.int64, device="cuda")      # (the ring kernel writes three tables: tools/stamp_ring_timeline.py)
    stats = torch.zeros((B, N, 2), dtype=torch.float64, device="cuda")
    a.w_hi, a.B, a.H, a.W, a.Hin, a.Win, a.N, a.Ktot = w.data_ptr(), B, H, W, H, W, N, K
    a.out_hi, a.stats, a.ksplit, a.acc_buf = out.data_ptr(), stats.data_ptr(), 1, buf.data_ptr()
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        assert getattr(l, kern)(C.byref(a), bnt, st) == 0, l.ctdd_last_error().decode()
    torch.cuda.synchronize()
    d = buf.cpu().numpy()[: nwg * 64].reshape(nwg, 8, 8)
    names = ["epi phase1", "epi phase2", "epi phase3", "stats begin", "flush"]
    tot = d[:, :, 7].astype(np.float64)
    print(kern, f"B={B} {H}x{W} C={Cs} N={N}: {nwg} WGs; wave total ticks mean {tot.mean():.0f} min {tot.min():.0f} max {tot.max():.0f}")
    for i, nm in enumerate(names):
        v = d[:, :, i].astype(np.float64)
        print(f"   {nm:10s} mean {v.mean():9.0f}  ({100*v.mean()/tot.mean():5.1f}%)  min {v.min():.0f} max {v.max():.0f}")
    rest = tot - d[:, :, :5].sum(-1)
    print(f"   other    mean {rest.mean():9.0f}  ({100*rest.mean()/tot.mean():5.1f}%)")
    span = d[:, :, 6].max() - d[:, :, 5].min()
    print(f"   kernel span ticks {span}")
run(83, 28, 28, [96], 96, 3)
run(256, 28, 28, [96], 96, 3)
run(83, 28, 28, [192], 192, 3)
