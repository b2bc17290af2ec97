"""Time line of k_conv_ring workgroups (stamps build, tools/build_conv_stamps.sh): when each 16-channel unit's barrier opens,
when the main loop and the epilogue end, and where the main loop's cycles go (DMA wait, barrier wait, unit set-up, taps).

    python tools/stamp_ring_timeline.py [B H W Cin N]
"""
import ctypes as C
import os
import sys

_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_R, os.path.join(_R, "continuous-time-diffusion-models-for-discrete-data_amd")]
import numpy as np
import torch
from ctdd import unet_engine as ue

l = C.CDLL(os.path.join(_R, "continuous-time-diffusion-models-for-discrete-data_amd", "libres_stamps.so"))
l.ctdd_unet_conv_ring.argtypes, l.ctdd_unet_conv_ring.restype = [C.c_void_p, C.c_int, C.c_void_p], C.c_int
l.ctdd_last_error.restype = C.c_char_p


def run(B, H, W, Cin, N, bnt=3):
    M = B * H * W
    x = torch.randn((M, Cin), device="cuda").to(torch.bfloat16)
    K = 9 * Cin
    w = (torch.randn((N, K), device="cuda") / K ** 0.5).to(torch.bfloat16)
    a = ue._ConvArgs()
    a.nseg = 1
    a.seg[0].hi, a.seg[0].C, a.seg[0].kind = x.data_ptr(), Cin, 0
    out = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    tile = 256 if bnt > 10 else 512
    nw = 4 if bnt > 10 else 8
    nwg = -(-M // tile) * -(-N // (32 * (bnt % 10)))
    buf = torch.zeros((3, nwg, nw, 8), dtype=torch.int64, device="cuda")
    stats = torch.zeros((B, N, 2), dtype=torch.float64, device="cuda")
    a.w_hi, a.B, a.H, a.W, a.Hin, a.Win, a.N, a.Ktot = w.data_ptr(), B, H, W, H, W, N, K
    a.out_hi, a.stats, a.ksplit, a.acc_buf = out.data_ptr(), stats.data_ptr(), 1, buf.data_ptr()
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        assert l.ctdd_unet_conv_ring(C.byref(a), bnt, st) == 0, l.ctdd_last_error().decode()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        l.ctdd_unet_conv_ring(C.byref(a), bnt, st)
    e1.record()
    e1.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 5
    d = buf.cpu().numpy().astype(np.float64)
    epi, tl, ml = d[0], d[1], d[2]
    tentry = ml[:, :, 5]
    print(f"B={B} {H}x{W} Cin={Cin} N={N} bnt={bnt}: {nwg} workgroups, {us:.1f} us per launch (events, stamps build)")
    print(f"  workgroup entry spread: {tentry.max() - tentry.min():.0f} ticks; wave total {epi[:, :, 7].mean():.0f} (max {epi[:, :, 7].max():.0f})")
    print("  unit barrier opens at (ticks after entry, mean over waves): " + " ".join(f"{tl[:, :, i].mean():.0f}" for i in range(6)))
    print(f"  main loop ends {tl[:, :, 6].mean():.0f}, epilogue rows end {tl[:, :, 7].mean():.0f}, done {epi[:, :, 7].mean():.0f}")
    print(f"  main loop sums: dma wait {ml[:, :, 0].mean():.0f}, barrier wait {ml[:, :, 1].mean():.0f}, prologue {ml[:, :, 2].mean():.0f}, "
          f"unit set-up {ml[:, :, 3].mean():.0f}, taps {ml[:, :, 4].mean():.0f}")
    print(f"  epilogue: acc->LDS {epi[:, :, 0].mean():.0f}, rows {epi[:, :, 1].mean():.0f}, column stats {epi[:, :, 2].mean():.0f}, "
          f"stats begin {epi[:, :, 3].mean():.0f}, flush {epi[:, :, 4].mean():.0f}")
    span = (tentry + epi[:, :, 7]).max() - tentry.min()
    print(f"  kernel span {span:.0f} ticks -> {span / us:.0f} ticks per us")


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run(*[int(v) for v in sys.argv[1:]])
    else:
        run(128, 28, 28, 96, 96)
        run(128, 28, 28, 192, 96)
        run(128, 28, 28, 96, 96, 13)
        run(128, 14, 14, 192, 192)
