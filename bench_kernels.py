"""Kernel-level roofline figures of the rate / sampling kernels: synthetic inputs resident in HBM,
one JSON line per (kernel, shape, t).

    logits ~ N(0,1) fp32 (B, D, S) from seed 1234, x ~ randint(0, S) int32 (B, D) from seed 1235, h = 1e-3,
    eps_ratio = 1e-9, is_ordinal, Philox seed 42; GaussianTargetRate(S=256, rate_sigma=6, Q_sigma=512, time_exp=100,
    time_base=3) at S = 256, the uniform-variant processes of the maze / synthetic configs at S = 3 / 2.

Algorithmic bytes per sample-step: D*S*4 (logits) + D*4 (x in) + D*4 (x out); a kernel that also writes a
(B, D, S) tensor (rates, probabilities) counts it.  `frac` = achieved GB/s / 8000 (HBM3E peak, MI355X_MICROARCH.md); the
S = 256 CT-ELBO / reverse_prob steps carry a D*S*S contraction and are priced against the matrix peak as well
(`mfma_frac`, 3 split-bf16 products vs the dense bf16 peak of 2500 TFLOP/s).  Durations: HIP events around `reps`
back-to-back launches on the launch stream, after warm-up.

    python bench_kernels.py [--quick] [--out profiles/r01_kernel_roofline.jsonl]
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "continuous-time-diffusion-models-for-discrete-data_amd"))

HBM_PEAK_GBS = 8000.0
MFMA_BF16_PEAK_TFLOPS = 2500.0


def timeit(fn, reps, warm=3):
    """Seconds per launch: `reps` launches captured into ONE HIP graph (back-to-back on the device: a ctypes launch costs the
    host ~13 us, more than the small-S kernels run), HIP events around its replay; eager launches if capture fails."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    graph = None
    try:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            with torch.cuda.graph(g, stream=side):
                for _ in range(reps):
                    fn()
        torch.cuda.current_stream().wait_stream(side)
        g.replay()
        torch.cuda.synchronize()
        graph = g
    except Exception:                                   # (a kernel that allocates outside the graph pool, a host sync ...)
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    if graph is not None:
        graph.replay()
    else:
        for _ in range(reps):
            fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true", help="one batch size per shape")
    ap.add_argument("--out", default=None, help="also append the lines to this file")
    ap.add_argument("--shapes", default=None, help="comma-separated subset of mnist,cifar10,maze,synthetic")
    a = ap.parse_args()
    if not torch.cuda.is_available():
        raise SystemExit("bench_kernels.py needs a GPU: libctdd has no CPU path")
    from ctdd import native
    from ctdd.process import DeviceForwardProcess
    dev = torch.device("cuda")
    sink = open(a.out, "a") if a.out else None

    def emit(rec):
        line = json.dumps(rec)
        print(line, flush=True)
        if sink:
            sink.write(line + "\n")

    def inputs(B, D, S):
        g = torch.Generator(device=dev).manual_seed(1234)
        logits = torch.randn((B, D, S), device=dev, generator=g)
        g2 = torch.Generator(device=dev).manual_seed(1235)
        x = torch.randint(0, S, (B, D), device=dev, generator=g2, dtype=torch.int32)
        return logits, x

    shapes = [  # (name, D, S, process kind, params, batches)
        ("mnist", 784, 256, "gaussian", dict(rate_sigma=6.0, Q_sigma=512.0, time_exp=100.0, time_base=3.0), [64, 256, 1024, 4096]),
        ("cifar10", 3072, 256, "gaussian", dict(rate_sigma=6.0, Q_sigma=512.0, time_exp=100.0, time_base=3.0), [16, 64, 256, 1024]),
        ("maze", 225, 3, "univar", dict(rate_const=1.7, t_func="sqrt_cos"), [4096, 65536]),
        ("synthetic", 32, 2, "univar", dict(rate_const=1.0, t_func="sqrt_cos"), [65536, 1048576]),
    ]
    h, eps, seed = 1e-3, 1e-9, 42
    for name, D, S, kind, params, batches in shapes:
        if a.shapes and name not in a.shapes.split(","):
            continue
        pr = DeviceForwardProcess(kind, S, dev, **params)
        if a.quick:
            batches = batches[1:2]
        for B in batches:
            logits, x = inputs(B, D, S)
            scan = B * (D * S * 4 + 8 * D)                       # algorithmic bytes of a fused step
            reps = max(3, min(50, int(2e9 // scan)))
            for t in ((0.5,) if a.quick else (0.01, 0.5, 0.99)):
                tt = torch.tensor([t], dtype=torch.float32)
                qt0 = pr.tables(tt, want_qt0=True)[0]
                beta = float(pr.beta(tt)[0])
                base = dict(shape=name, B=B, D=D, S=S, t=t, reps=reps)

                def rec(kernel, secs, nbytes, flops=0.0):
                    r = dict(base, kernel=kernel, us=round(secs * 1e6, 2), bytes=nbytes, GBps=round(nbytes / secs / 1e9, 1),
                             frac=round(nbytes / secs / 1e9 / HBM_PEAK_GBS, 4), sample_steps_per_s=round(B / secs, 1))
                    if flops:
                        r["TFLOPs"] = round(flops / secs / 1e12, 1)
                        r["mfma_frac"] = round(flops / secs / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4)
                    emit(r)

                out = torch.empty((B, D), dtype=torch.int32, device=dev)
                # K5 + K6 fused, CRM branch, direct logits: a pure scan of the logit tensor (HBM-bound for every S)
                rec("tauleap_step crm/direct (k_rows)",
                    timeit(lambda: native.tauleap_step(native.BRANCH_CRM, "direct", logits, x, None, pr.base_rate, beta, eps, h, 1,
                                                       seed, 0, out=out), reps), scan)
                # K7 LBJF step, same branch
                rec("lbjf_step crm/direct (k_rows)",
                    timeit(lambda: native.lbjf_step(native.BRANCH_CRM, "direct", logits, x, None, pr.base_rate, beta, eps, h, 0,
                                                    None, seed, 0), reps), scan)
                # K10 final argmax
                rec("argmax (k_argmax)", timeit(lambda: native.argmax(logits), reps), B * (D * S * 4 + 4 * D))
                if S == 256:
                    contraction = 2.0 * B * D * S * S * 3            # three split-bf16 products
                    tabs = native.S256Tables(qt0, pr.base_rate, eps)
                    rec("tauleap_step ctelbo (k_tauleap_s256)",
                        timeit(lambda: native.tauleap_step_s256(logits, x, tabs, 0, beta, h, 1, seed, 0, out=out), reps), scan, contraction)
                    tabb = native.S256Tables(qt0, pr.base_rate, eps, bf16=True)     # single bf16 product (the bf16 network's mode)
                    rec("tauleap_step ctelbo, single bf16 product (k_tauleap_s256_b16)",
                        timeit(lambda: native.tauleap_step_s256(logits, x, tabb, 0, beta, h, 1, seed, 0, out=out), reps), scan, contraction / 3)
                    lg16 = logits.to(torch.bfloat16)
                    scan16 = B * (D * S * 2 + 8 * D)                     # SURVEY 8d's bf16 figure: 407 680 B per MNIST sample-step
                    rec("tauleap_step ctelbo, single bf16 product, bf16 logits (k_tauleap_s256_b16)",
                        timeit(lambda: native.tauleap_step_s256(lg16, x, tabb, 0, beta, h, 1, seed, 0, out=out), reps), scan16, contraction / 3)
                    del lg16
                    tabc = native.S256Tables(qt0, pr.base_rate, 0.0, crm=True)
                    rec("tauleap_step crm/reverse_prob (k_tauleap_s256)",
                        timeit(lambda: native.tauleap_step_s256(logits, x, tabc, 0, beta, h, 1, seed, 0, out=out), reps), scan, contraction)
                    if B * D <= 256 * 784:                           # the generic fp32-FMA contraction, for scale
                        rec("tauleap_step ctelbo (k_rows, fp32 FMA contraction)",
                            timeit(lambda: native.tauleap_step(native.BRANCH_CTELBO, "direct", logits, x, qt0[0], pr.base_rate, beta,
                                                               eps, h, 1, seed, 0, out=out), max(3, reps // 4)), scan)
                else:
                    rec("tauleap_step ctelbo (k_rows)",
                        timeit(lambda: native.tauleap_step(native.BRANCH_CTELBO, "direct", logits, x, qt0[0], pr.base_rate, beta, eps,
                                                           h, 1, seed, 0, out=out), reps), scan)
                    rec("tauleap_step crm/reverse_prob (k_rows)",
                        timeit(lambda: native.tauleap_step(native.BRANCH_CRM, "reverse_prob", logits, x, qt0[0], pr.base_rate, beta,
                                                           eps, h, 1, seed, 0, out=out), reps), scan)
                # K5 rates only (reads the logits, writes a (B, D, S) rate tensor)
                if B * D * S * 4 <= 2 ** 31:
                    rate_t = pr.rate(tt)
                    rec("reverse_rates crm/direct (k_rows)",
                        timeit(lambda: native.reverse_rates(native.BRANCH_CRM, "direct", logits, x, None, rate_t, eps, want_ratio=False),
                               max(3, reps // 2)), B * (2 * D * S * 4 + 4 * D))
            # K2 noising: rows of q_{t|0}[x0] raced against Philox exponentials (one table per sample)
            if t is not None and B <= 65536 and S <= 256:
                Bn = min(B, 1024 if S == 256 else 4096)
                ts = torch.rand(Bn, device=dev) * 0.9 + 0.05
                q = pr.tables(ts.cpu(), want_qt0=True)[0]
                x0 = x[:Bn].contiguous()
                secs = timeit(lambda: native.noise_categorical(q, x0, None, None, seed, 0), max(3, reps // 2))
                emit(dict(shape=name, B=Bn, D=D, S=S, kernel="noise_categorical (Philox race)", us=round(secs * 1e6, 2),
                          bytes=Bn * D * (S * 4 + 8), GBps=round(Bn * D * (S * 4 + 8) / secs / 1e9, 1),
                          note="bytes = gathered table rows (L2-resident tables: not an HBM figure)", draws_per_s=round(Bn * D / secs, 1)))
            del logits, x
            torch.cuda.empty_cache()
    if sink:
        sink.close()


if __name__ == "__main__":
    main()
