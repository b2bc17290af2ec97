"""Phase stamps (s_memtime, per wave and 128-row tile) of the two S = 256 fused-step kernels on synthetic inputs.

    tools/build_stamps.sh && python tools/stamps_s256.py [--batch 256] [--scale 1.0]

Prints, per (kernel, t): the launch time and the median cycles a wave spends in phase 1 (loads + softmax + w), phase 2
(contraction), phase 3 (rates) and phase 4 (draw); `span` = first stamp of the launch to the last, in microseconds of the
100 MHz real-time counter, i.e. what the stamps themselves add is visible against the un-stamped kernel's time.
"""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "continuous-time-diffusion-models-for-discrete-data_amd")
sys.path[:0] = [ROOT, PKG]
from ctdd import native  # noqa: E402
from ctdd.process import DeviceForwardProcess  # noqa: E402


def bench_regime(a):
    import bench
    lib = ctypes.CDLL(os.path.join(PKG, "libctdd_stamps.so"))
    fn = lib.ctdd_tauleap_step_s256
    fn.argtypes = native._SIGS["ctdd_tauleap_step_s256"][0]
    fn.restype = ctypes.c_int
    cfg, model, sampler = bench.build_model(torch.device("cuda"))
    sampler.seed, sampler.rank_stream = 42, 0
    N, D = a.batch, 784
    nt = (N * D + 127) // 128
    with torch.no_grad():
        st = sampler.begin(model, N, pipeline=False)
        done = 0
        for target in (5, 45, 500):
            for i in range(done, target):
                sampler.advance(st, i)
            done = i = target
            t_ones = sampler._t_ones(st.t32, i, st.N, st.dev)
            with sampler._borrow(model):
                lg = sampler._net_logits(model, st.x, t_ones, st.fast).clone()
            h = float(np.float32(st.ts[i] - st.ts[i + 1]))
            for name, flag, src, hot in (("3-product", 0, lg.float(), False), ("bf16", native.STEP_BF16, lg.float(), False),
                                         ("bf16/l16", native.STEP_BF16 | native.STEP_LOGITS_BF16, lg.to(torch.bfloat16), False),
                                         ("l16 hot", native.STEP_BF16 | native.STEP_LOGITS_BF16, lg.to(torch.bfloat16), True)):
                dbg = torch.zeros(nt * 4 * 8, dtype=torch.int64, device="cuda")
                out = torch.empty(N, D, dtype=torch.int32, device="cuda")
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                for _ in range(2):
                    if hot:                                # right behind three network forwards, as in the sampler loop
                        with sampler._borrow(model):
                            for _ in range(3):
                                model(st.x.long(), t_ones)
                    e0.record()
                    rc = fn(src.data_ptr(), st.x.data_ptr(), None, st.fast.step_ptr(i), st.fast.RT0.data_ptr(), st.fast.R0.data_ptr(),
                            float(st.betas[i]), h, st.flags | flag, st.key, 12345, N, D, None, out.data_ptr(), dbg.data_ptr(), None)
                    e1.record()
                    torch.cuda.synchronize()
                raw = dbg.cpu().numpy().reshape(nt * 4, 8)
                d = raw[:, :5]
                ph = np.diff(d, axis=1)
                clk = ""
                if flag:
                    dt = (raw[:, 5] - raw[:, 6]).astype(np.float64) / 100.0          # us per wave (100 MHz counter)
                    clk = " ; clock %.2f GHz" % np.median((d[:, 4] - d[:, 0]) / np.maximum(dt, 1e-3) / 1e3)
                print(f"step {i} t={st.ts[i]:.3f} {name:10s} rc={rc} launch {e0.elapsed_time(e1) * 1e3:7.1f} us  median cycles: p1 %6d p2 %6d p3 %6d p4 %6d (p90 %6d) | wave total %6d"
                      % (*np.median(ph, axis=0), np.percentile(ph[:, 3], 90), np.median(d[:, 4] - d[:, 0])) + clk, flush=True)
            # what the rows look like at this step: total rate * h per dimension
            rates = native.tauleap_step_s256(lg.float().contiguous(), st.x, st.fast, i, st.betas[i], h, st.flags, st.key, 1, want_rates=True, want_x=False)[1]
            lam = rates.sum(-1) * h
            print(f"   Lambda = h sum_s rate: median {float(lam.median()):.3g}, p90 {float(lam.quantile(0.9)):.3g}, max {float(lam.max()):.3g}; "
                  f"share > 12: {float((lam > 12).float().mean()):.3f}, > 64: {float((lam > 64).float().mean()):.3f}", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--scale", type=float, default=1.0, help="std of the synthetic logits")
    ap.add_argument("--h", type=float, default=1e-3)
    ap.add_argument("--bench", action="store_true", help="bench.py's regime: real network logits and sampler state at steps 5 / 45 / 500")
    a = ap.parse_args()
    if a.bench:
        return bench_regime(a)
    lib = ctypes.CDLL(os.path.join(PKG, "libctdd_stamps.so"))
    fn = lib.ctdd_tauleap_step_s256
    fn.argtypes = native._SIGS["ctdd_tauleap_step_s256"][0]
    fn.restype = ctypes.c_int
    N, D, S = a.batch, 784, 256
    dev = torch.device("cuda")
    pr = DeviceForwardProcess("gaussian", S, dev, rate_sigma=6.0, Q_sigma=512.0, time_exp=100.0, time_base=3.0)
    g = torch.Generator(device=dev).manual_seed(1234)
    logits = torch.randn((N, D, S), device=dev, generator=g) * a.scale
    x = torch.randint(0, S, (N, D), device=dev, generator=g, dtype=torch.int32)
    nt = (N * D + 127) // 128
    for t in (0.01, 0.5, 0.99):
        tt = torch.tensor([t])
        qt0 = pr.tables(tt, want_qt0=True)[0]
        beta = float(pr.beta(tt)[0])
        tabs = native.S256Tables(qt0, pr.base_rate, 1e-9)
        lg16 = logits.to(torch.bfloat16)
        for name, flag in (("3-product", 0), ("bf16", native.STEP_BF16), ("bf16/l16", native.STEP_BF16 | native.STEP_LOGITS_BF16)):
            src = lg16 if flag & native.STEP_LOGITS_BF16 else logits
            dbg = torch.zeros(nt * 4 * 8, dtype=torch.int64, device=dev)
            out = torch.empty(N, D, dtype=torch.int32, device=dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(2):
                e0.record()
                rc = fn(src.data_ptr(), x.data_ptr(), None, tabs.step_ptr(0), tabs.RT0.data_ptr(), tabs.R0.data_ptr(), beta, a.h,
                        1 | flag, 42, 12345, N, D, None, out.data_ptr(), dbg.data_ptr(), None)
                e1.record()
                torch.cuda.synchronize()
            raw = dbg.cpu().numpy().reshape(nt * 4, 8)
            d = raw[:, :5]
            ph = np.diff(d, axis=1)
            span = (raw[:, 5].max() - raw[:, 5].min()) / 100.0
            extra = ""
            if flag:                                           # bf16 kernel: stamp 5 = end of phase 1's single wait for its loads
                extra = " ; p1 wait %d" % np.median(raw[:, 7] - raw[:, 0])
            print(f"t={t} {name:10s} rc={rc} launch {e0.elapsed_time(e1) * 1e3:7.1f} us  median cycles: p1 %6d p2 %6d p3 %6d p4 %6d | wave total %6d (p90 %6d) ; span %.1f us"
                  % (*np.median(ph, axis=0), np.median(d[:, 4] - d[:, 0]), np.percentile(d[:, 4] - d[:, 0], 90), span) + extra, flush=True)


if __name__ == "__main__":
    main()
