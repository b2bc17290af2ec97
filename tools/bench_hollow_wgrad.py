"""the hollow blocks' weight-gradient launches (rows = 28800), graph-timed, over M-split widths."""
import sys, os
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_R, os.path.join(_R, 'continuous-time-diffusion-models-for-discrete-data_amd')]
import torch
from bench_kernels import timeit
from ctdd import hollow_train as ht
import ctypes as C
from ctdd import unet_train
R = 28800
dev = torch.device("cuda")
def run(N, K, bias, wgs, nwn=None):
    x = torch.randn((R, K), device=dev).to(torch.bfloat16); dy = torch.randn((R, N), device=dev).to(torch.bfloat16)
    dw = torch.zeros((N, K), device=dev); dbs = torch.zeros((N, 8), device=dev)
    ents = []
    for xs, gw, Kc in ((x, dw, K),) + (((torch.ones((R, 8), dtype=torch.bfloat16, device=dev), dbs, 8),) if bias else ()):
        a = unet_train._WgradArgs()
        a.x, a.dy, a.gw = xs.data_ptr(), dy.data_ptr(), gw.data_ptr()
        a.B, a.H, a.W, a.Hin, a.Win, a.N, a.ldy, a.C, a.Ktot, a.koff, a.kind = 1, R, 1, R, 1, N, N, Kc, Kc, 0, unet_train.WG_1x1
        a.nwn, a.nlr = ht._wgrad_geometry(R, N, Kc, True)
        if nwn and Kc == K:
            a.nwn = nwn
            tb, epv = 64, 8
            nwc = 4 // nwn
            a.nlr = min(144 * 1024 // ((nwn + nwc) * tb), 2048 // (32 * nwn // epv), 2560 // (32 * nwc // epv)) // 16 * 16
        a.nchunks = -(-R // a.nlr)
        groups = -(-N // (32 * a.nwn)) * -(-Kc // (32 * (4 // a.nwn)))
        a.grid_x, a.tap = max(1, min(a.nchunks, -(-wgs // groups))), 0
        ents.append(a)
    tab = (unet_train._WgradArgs * len(ents))(*ents)
    dtab = ht._device_table(bytes(tab), dev)
    l = ht.lib()
    def f():
        l.ctdd_unet_wgrad(dtab.data_ptr(), C.addressof(tab), len(ents), 0, torch.cuda.current_stream().cuda_stream)
    t = timeit(f, 20)
    return t * 1e6, ents[0].nwn, ents[0].nlr, ents[0].grid_x
for N, K, name in ((128, 128, "out_proj"), (384, 128, "in_proj"), (128, 1024, "fc2"), (1024, 128, "fc1")):
    fl = 2.0 * R * N * K
    for bias in (False, True):
        for wgs in (256, 512, 768, 1536):
            for nwn in (None, 1, 2, 4):
                try:
                    us, a_nwn, a_nlr, gx = run(N, K, bias, wgs, nwn)
                except Exception as e:
                    print(name, "failed", e); continue
                print(f"{name:9s} N={N:5d} K={K:5d} bias={int(bias)} wgs={wgs:5d} nwn={a_nwn} nlr={a_nlr:4d} grid_x={gx:3d}: {us:7.1f} us  {fl/us/1e6:6.1f} TF/s", flush=True)
