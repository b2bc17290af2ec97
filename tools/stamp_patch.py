"""Phase stamps of k_conv_patch (diagnostic build -DCTDD_PATCH_STAMPS, tools/build_conv_stamps.sh): per wave, cycles from the
kernel's first instruction to: each chunk's tiles staged (one stamp per chunk), main loop done, epilogue rows done, statistics
flushed.  python tools/stamp_patch.py [B H W C N bk bnt wm]"""
import ctypes as C
import os
import sys

_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_R, os.path.join(_R, "continuous-time-diffusion-models-for-discrete-data_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402
from ctdd import unet_engine as ue  # noqa: E402

lib = C.CDLL(os.path.join(_R, "continuous-time-diffusion-models-for-discrete-data_amd", "libres_stamps.so"))
lib.ctdd_unet_conv_patch.argtypes, lib.ctdd_unet_conv_patch.restype = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p], C.c_int
lib.ctdd_last_error.restype = C.c_char_p


def run(B, H, W, Cin, N, bk, bnt, wm):
    M = B * H * W
    x = torch.randn((M, Cin), device="cuda").to(torch.bfloat16)
    K = 9 * Cin
    w = (torch.randn((N, K), device="cuda") / K ** 0.5).to(torch.bfloat16)
    a = ue._ConvArgs()
    a.nseg = 1
    a.seg[0].hi, a.seg[0].C, a.seg[0].kind = x.data_ptr(), Cin, 0
    out = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    nwg = -(-M // (4 * wm)) * -(-N // (32 * bnt))
    buf = torch.zeros((nwg * 4 * 8,), dtype=torch.int64, device="cuda")
    stats = torch.zeros((B, N, 2), dtype=torch.float64, device="cuda")
    a.w_hi, a.B, a.H, a.W, a.Hin, a.Win, a.N, a.Ktot = w.data_ptr(), B, H, W, H, W, N, K
    a.out_hi, a.stats, a.ksplit, a.acc_buf = out.data_ptr(), stats.data_ptr(), 1, buf.data_ptr()
    st = torch.cuda.current_stream().cuda_stream
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        e0.record()
        assert lib.ctdd_unet_conv_patch(C.byref(a), bk, bnt, wm, st) == 0, lib.ctdd_last_error().decode()
        e1.record()
        torch.cuda.synchronize()
    d = buf.cpu().numpy().reshape(nwg * 4, 8).astype(np.float64)
    n = int((d[0] > 0).sum())
    rel = d[:, 1:n] - d[:, :1]
    names = [f"chunk{i} staged" for i in range(n - 4)] + ["main loop done", "epilogue rows", "stats flushed"]
    print(f"B={B} {H}x{W} C={Cin} N={N} bk={bk} bnt={bnt} wm={wm}: {nwg} workgroups, launch {e0.elapsed_time(e1) * 1e3:.1f} us; median cycles since kernel start:")
    for i, nm in enumerate(names):
        print(f"   {nm:16s} {np.median(rel[:, i]):9.0f}   (p90 {np.percentile(rel[:, i], 90):9.0f})")
    span = d[:, n - 1].max() - d[:, 0].min()
    print(f"   first start -> last end: {span:.0f} cycles")


if __name__ == "__main__":
    args = [int(v) for v in sys.argv[1:]] or [128, 7, 7, 192, 192, 64, 1, 32]
    run(*args)
