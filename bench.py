"""bench.py -- tau-leaping sample-steps/s on the MNIST tauLDR config (D=784, S=256).

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one pass of the hot path over one batch: the exact body of TauL.sample's loop
(reference lib/sampling/sampling.py:119-160) for `--batch` samples per GPU -- score-network
forward, then ONE fused libctdd launch (softmax -> S x S ratio contraction -> forward-rate gather
-> Poisson jump draw -> state update).  Inputs (state x, per-step q_{t|0} tables, random-init
weights) are resident in HBM before the timed region.  Sampling shards over GPUs with no
data-path collective: weak scaling, value = all ranks' sample-steps / max-rank time.

The JSON line also carries
  roofline          the fused tau-leap kernel (k_tauleap_s256_b16 in the bf16 mode the headline runs: one bf16 product,
                    bf16 logits straight from the output convolution; k_tauleap_s256 in the fp32 parity mode):
                    algorithmic bytes per sample-step (407 680 B with bf16 logits, 809 088 B with fp32
                    logits) x batch / its mean launch duration (HIP events on the launch stream); `mfma_view` prices
                    the same launches by their matrix FLOPs; `fp32_logit_equivalent_frac` is the same duration priced
                    with the fp32-logit byte count rounds 1 and 2 reported;
  roofline_network  the score-network forward (the larger share of the step): 2 M N K matrix FLOPs of its convolutions /
                    the forward's wall time as it runs in the loop (graph replay, two sub-batches on parallel streams, GroupNorm
                    etc. included) vs the dense bf16 MFMA peak; the convolution launches replayed one by one are reported too;
  cpu_baseline      the CPU oracle (checker, never the product) running the same step on host cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "continuous-time-diffusion-models-for-discrete-data_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense bf16 (no sparsity)
D, S = 784, 256
ALGO_BYTES_PER_SAMPLE_STEP = D * S * 4 + D * 4 + D * 4      # fp32 logits + x in + x out
ALGO_BYTES_PER_SAMPLE_STEP_BF16 = D * S * 2 + D * 4 + D * 4  # bf16 logits + x in + x out


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="samples per GPU (SURVEY 8d: 256)")
    ap.add_argument("--cpu-batch", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fp32-mode", action="store_true", help="skip the fp32 parity-mode timing")
    ap.add_argument("--no-train-step", action="store_true", help="skip the training-step timing")
    ap.add_argument("--no-configs", action="store_true", help="skip the BASELINE configs 3-5 block")
    ap.add_argument("--train", action="store_true",
                    help="time the data-parallel TRAINING step instead (MNIST tauLDR CT-ELBO, 64 samples per GPU, DDP over RCCL)")
    ap.add_argument("--engine-streams", type=int, default=None, help="sub-batches of the U-Net engine on parallel streams")
    ap.add_argument("--model-opt", action="append", default=[], help="integer engine option, key=value (tuning sweeps)")
    ap.add_argument("--sampler-opt", action="append", default=[], help="integer sampler option, key=value (tuning sweeps)")
    return ap.parse_args()


def build_model(device):
    import lib.models.models  # noqa: F401
    import lib.models.model_utils as mu
    import lib.sampling.sampling  # noqa: F401
    import lib.sampling.sampling_utils as su
    from config.mnist_config.config_tauUnet_mnist import get_config
    cfg = get_config()
    cfg.device = str(device)
    torch.manual_seed(0)
    model = mu.create_model(cfg, device)
    model.eval()
    sampler = su.get_sampler(cfg)
    return cfg, model, sampler


def kernel_roofline(sampler, st, steps, batch, restore_state=False):
    """Mean duration of the fused tau-leap launch alone, HIP events on the launch stream, over the
    same steps (logits recomputed outside the event bracket, exactly as the sampler loop gets them)."""
    model = st.model
    times = []
    x_keep = st.x
    l16 = False
    with sampler._borrow(model):
        for i in steps:
            t_ones = sampler._t_ones(st.t32, i, st.N, st.dev)
            logits = sampler._net_logits(model, st.x, t_ones, st.fast)
            l16 = logits.dtype == torch.bfloat16
            h = float(np.float32(st.ts[i] - st.ts[i + 1]))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            st.x = sampler._leap(model, logits, st.x, st.qt0[i], st.fast, i, st.betas[i], h, st.flags, st.key,
                                 10_000_000 + i)
            e1.record()
            e1.synchronize()
            times.append(e0.elapsed_time(e1) * 1e-3)
    dur = float(np.mean(times))
    if restore_state:
        st.x = x_keep
    bf16_step = st.fast is not None and st.fast.bf16
    per = ALGO_BYTES_PER_SAMPLE_STEP_BF16 if l16 else ALGO_BYTES_PER_SAMPLE_STEP
    traffic, traffic_src = None, None
    pmc = os.path.join(ROOT, "profiles", "r03_pmc_traffic_tauleap_s256_b16.json" if bf16_step else "r01_pmc_traffic_tauleap_s256.json")
    if batch == 256 and os.path.exists(pmc):          # HBM bytes per launch from the committed rocprofv3 --pmc passes (same workload)
        with open(pmc) as f:
            rec = json.load(f)
        traffic, traffic_src = round(rec["hbm_bytes_per_launch"]), os.path.relpath(pmc, ROOT) + " (FETCH_SIZE x2 + WRITE_SIZE, separate --pmc passes)"
    achieved = per * batch / dur / 1e9
    nprod = 1 if bf16_step else 3                               # bf16 mode: one product; parity mode: hi*hi + hi*lo + lo*hi
    mfma_flops = nprod * 2 * D * S * S * batch
    kernel = ("ctdd k_tauleap_s256_b16 (LDS-DMA rows -> softmax in MFMA fragment order -> one bf16 MFMA contraction -> rates -> Poisson draw + update; "
              + ("bf16" if l16 else "fp32") + " logits)") if bf16_step else \
        "ctdd k_tauleap_s256 (fused softmax + split-bf16 MFMA contraction + Poisson draw + update)"
    return {"kernel": kernel, "bound": "hbm", "achieved": round(achieved, 2),
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_src,
            "avg_launch_us": round(dur * 1e6, 2), "algorithmic_bytes_per_launch": per * batch, "logits_dtype": "bf16" if l16 else "f32",
            "fp32_logit_equivalent_frac": round(ALGO_BYTES_PER_SAMPLE_STEP * batch / dur / 1e9 / HBM_PEAK_GBS, 5),
            "mfma_view": {"achieved": round(mfma_flops / dur / 1e12, 2), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                          "frac": round(mfma_flops / dur / 1e12 / MFMA_BF16_PEAK_TFLOPS, 5), "products": nprod}}


def network_roofline(model, batch, sampler=None):
    """The U-Net engine's convolution launches replayed one by one (eager, outside the HIP graph) with HIP
    events on the launch stream: summed 2*M*N*K matrix FLOPs / summed durations."""
    eng = getattr(model, "_engine", None)
    if eng is None:
        return None
    # the forward exactly as the sampler loop issues it: int32 states, the plan's own output buffer, and (when the sampler
    # precomputes its grid's time projections) a time-projection row instead of the time path
    from lib.models.models import borrow_engine_output
    dev = next(model.parameters()).device
    x = torch.randint(0, S, (batch, D), device=dev, dtype=torch.int32)
    t = torch.full((batch,), 0.5, device=dev)
    use_row = sampler is not None and getattr(sampler.cfg.sampler, "time_table", True) and hasattr(model, "engine_time_table")
    row = model.engine_time_table(t[:1])[0] if use_row else None

    def forward():
        with (sampler._borrow(model) if sampler is not None else borrow_engine_output(model)):
            model._engine_time_row = row
            try:
                return model(x, t)
            finally:
                model._engine_time_row = None

    before = set(eng._plans.keys())
    forward()
    torch.cuda.synchronize()
    lb = bool(forward().dtype == torch.bfloat16)
    cand = [k for k in eng._plans if k[0] == batch and k[1] == torch.int32 and k[-1] == lb and (("row" in k) == bool(use_row)) and "slot" not in k]
    new_keys = [k for k in cand if k not in before]
    key = (new_keys or cand or [None])[0]
    if key is None:
        return None
    st = eng._plans[key]
    plans = list(st[1]) if isinstance(st, tuple) else [st]       # (logits, sub-plans, streams) when the batch runs as sub-batches
    flops, secs, n = 0, 0.0, 0
    for st in plans:
        eng._run_plan(st)
        torch.cuda.synchronize()
        for step in st.plan:
            if not getattr(step, "flops", 0):
                continue
            step()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                step()
            e1.record()
            e1.synchronize()
            secs += e0.elapsed_time(e1) * 1e-3 / 3
            flops += step.flops
            n += 1
    seq = flops / secs / 1e12
    # the forward as it runs in the timed loop (HIP-graph replay, sub-batches on parallel streams, all kernels of the network)
    forward()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        forward()
    e1.record()
    e1.synchronize()
    fwd = e0.elapsed_time(e1) * 1e-3 / 10
    ach = flops / fwd / 1e12
    return {"kernel": "score-network forward: ctdd k_conv_ring / k_conv_patch / k_conv_igemm bf16 implicit-GEMM convolutions (+ GroupNorm, attention, time MLP)",
            "bound": "mfma", "achieved": round(ach, 2), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / MFMA_BF16_PEAK_TFLOPS, 5),
            "traffic": None, "forward_ms": round(fwd * 1e3, 3), "matrix_gflop_per_forward": round(flops / 1e9, 1), "sub_batches": len(plans),
            "conv_launches_per_forward": n, "conv_launches_sequential_tflops": round(seq, 2), "conv_launches_sequential_us": round(secs * 1e6, 1)}


def fp32_parity_mode(cfg, model, sampler, batch, steps=6, warmup=2):
    """The same sampler step with the score network in its parity mode (`engine_precision="fp32"`: exact-fp32 matrix
    instructions, the mode that meets the 1e-4 logit bar against the reference's golden logits, tests/test_gpu_unet.py)."""
    prev_prec, prev_eng = getattr(cfg.model, "engine_precision", "bf16"), model._engine
    cfg.model.engine_precision, model._engine = "fp32", None
    try:
        st = sampler.begin(model, batch)
        for i in range(warmup):
            sampler.advance(st, i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(warmup, warmup + steps):
            sampler.advance(st, i)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        del st
    finally:
        cfg.model.engine_precision, model._engine = prev_prec, prev_eng
    return {"value": round(batch * steps / el, 2), "unit": "sample-steps/s", "ms_per_step": round(el / steps * 1e3, 3), "steps": steps,
            "dtype": "f32", "note": "score network on v_mfma_f32_32x32x2_f32 (logits within 1e-4 of the reference's goldens); tau-leap step on the three-product "
                    "split-bf16 kernel k_tauleap_s256 (rates within 1e-4 of the oracle)"}


def train_step_timing(batch=64, steps=10, warmup=4):
    """One MNIST tauLDR CT-ELBO training step (config_tauUnet_mnist, batch 64: zero_grad -> noising + CT-ELBO (K2/K3/K11) ->
    backward -> clip + Adam + EMA (K28)) with the score network on the hand-written training plan (ctdd/unet_train.py) and,
    next to it, on torch autograd device ops (what round 1 trained with)."""
    import lib.losses.losses  # noqa: F401
    import lib.losses.losses_utils as lu
    import lib.models.model_utils as mu
    import lib.optimizers.optimizers  # noqa: F401
    import lib.optimizers.optimizers_utils as ou
    import lib.training.training  # noqa: F401
    import lib.training.training_utils as tu
    from config.mnist_config.config_tauUnet_mnist import get_config
    out = {}
    for tag, engine, prec in (("hip_plan_bf16", "hip", "bf16"), ("hip_plan_fp32", "hip", "fp32"), ("torch_autograd_fp32", "torch", None)):
        cfg = get_config()
        cfg.model.engine = engine
        if prec:
            cfg.model.engine_precision = prec
        n_steps = 3 if prec == "fp32" else steps
        torch.manual_seed(0)
        model = mu.create_model(cfg, torch.device("cuda"))
        state = {"model": model, "optimizer": ou.get_optimizer(model.parameters(), cfg), "n_iter": 0}
        step, loss = tu.get_train_step(cfg), lu.get_loss(cfg)
        mb = torch.randint(0, S, (batch, 1, 28, 28), device="cuda")
        for _ in range(warmup):
            step.step(state, loss, mb)
            state["n_iter"] += 1
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n_steps):
            step.step(state, loss, mb)
            state["n_iter"] += 1
        torch.cuda.synchronize()
        out[tag] = round((time.perf_counter() - t0) / n_steps * 1e3, 3)
        del model, state
    return {"workload": f"MNIST tauLDR CT-ELBO training step, batch {batch} (Standard.step: loss + backward + clip + Adam + EMA)",
            "ms_per_step": out, "speedup": round(out["torch_autograd_fp32"] / out["hip_plan_bf16"], 2),
            "speedup_like_for_like_fp32": round(out["torch_autograd_fp32"] / out["hip_plan_fp32"], 2),
            "note": "speedup = torch autograd fp32 / HIP plan bf16 (kernels AND precision change); speedup_like_for_like_fp32 = the same torch step / "
                    "the HIP plan on exact-fp32 matrix instructions (1/16 of the bf16 matrix rate)"}


def hollow_train_step_timing(batch=128, steps=10, warmup=4):
    """One training step of the maze hollow transformer (config_hollow_maze: D = 225, S = 3, E = 128, 2 x 8 blocks, dropout 0.1,
    ScoreElbo with reverse_prob logits, batch 128): the network forward / backward on the HIP training path
    (ctdd/hollow_train.py, bf16 GEMM / attention operands) and on torch autograd device ops."""
    import lib.losses.losses  # noqa: F401
    import lib.losses.losses_utils as lu
    import lib.models.model_utils as mu
    import lib.optimizers.optimizers  # noqa: F401
    import lib.optimizers.optimizers_utils as ou
    import lib.training.training  # noqa: F401
    import lib.training.training_utils as tu
    from config.maze_config.config_hollow_maze import get_config
    out = {}
    for tag, engine, prec in (("hip_bf16", "hip", "bf16"), ("hip_fp32", "hip", "fp32"), ("torch_autograd_fp32", "torch", None)):
        cfg = get_config()
        cfg.device, cfg.model.engine = "cuda", engine
        if prec:
            cfg.model.engine_train_precision = prec
        torch.manual_seed(0)
        model = mu.create_model(cfg, torch.device("cuda"))
        state = {"model": model, "optimizer": ou.get_optimizer(model.parameters(), cfg), "n_iter": 0}
        step, loss = tu.get_train_step(cfg), lu.get_loss(cfg)
        mb = torch.randint(0, 3, (batch, 1, 15, 15), device="cuda")
        for _ in range(warmup):
            step.step(state, loss, mb)
            state["n_iter"] += 1
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step.step(state, loss, mb)
            state["n_iter"] += 1
        torch.cuda.synchronize()
        out[tag] = round((time.perf_counter() - t0) / steps * 1e3, 3)
        del model, state
    return {"workload": f"maze hollow-transformer ScoreElbo training step, batch {batch} (Standard.step)", "ms_per_step": out,
            "speedup": round(out["torch_autograd_fp32"] / out["hip_bf16"], 2),
            "speedup_like_for_like_fp32": round(out["torch_autograd_fp32"] / out["hip_fp32"], 2)}


def baseline_configs():
    """BASELINE.json configs 3-5 next to the headline (config 2), one MI355X, random-init weights, a few steps each:
    sampler throughput in sample-steps/s (a whole `sampler.sample(model, N)` call: tables, initial draw, the loop, final
    arg-max where the sampler has one, device-to-host copy) and training steps in ms (Standard.step on the HIP plans)."""
    import lib.losses.losses  # noqa: F401
    import lib.losses.losses_utils as lu
    import lib.models.model_utils as mu
    import lib.optimizers.optimizers  # noqa: F401
    import lib.optimizers.optimizers_utils as ou
    import lib.sampling.sampling_utils as su
    import lib.training.training  # noqa: F401
    import lib.training.training_utils as tu
    from config.cifar10_config.config_tauUnet_cifar10 import get_config as c10
    from config.maze_config.config_hollow_maze import get_config as maze
    from config.mnist_config.config_hollow_mnist import get_config as hmnist
    dev = torch.device("cuda")
    out = {}

    def sample_rate(key, what, get_config, N, steps, model_over=None, **over):
        cfg = get_config()
        cfg.device = "cuda"
        cfg.sampler.num_steps = steps
        for k_, v_ in over.items():
            setattr(cfg.sampler, k_, v_)
        for k_, v_ in (model_over or {}).items():
            setattr(cfg.model, k_, v_)
        torch.manual_seed(0)
        model = mu.create_model(cfg, dev)
        model.eval()
        smp = su.get_sampler(cfg)
        smp.seed = 1
        smp.sample(model, N)                       # warm-up: plans, graphs, tables
        torch.cuda.synchronize()
        els = []
        for _ in range(3):                         # median of three whole calls (a single 20-step call is ~60 ms: one host hiccup
            t0 = time.perf_counter()               #  or a first-use allocation moved it by 30 % between runs)
            smp.sample(model, N)
            torch.cuda.synchronize()
            els.append(time.perf_counter() - t0)
        el = sorted(els)[1]
        calls = steps * (2 if cfg.sampler.name == "MidPointTauL" else 1)
        out[key] = {"workload": what, "sampler": cfg.sampler.name, "N": N, "steps": steps, "value": round(N * steps / el, 1),
                    "unit": "sample-steps/s", "ms_per_step": round(el / steps * 1e3, 3), "network_calls_per_step": calls // steps,
                    "timing": "median of 3 whole sample() calls after one warm-up call"}
        del model, smp
        torch.cuda.empty_cache()

    def train_ms(key, what, get_config, shape, S_, B, loss_name=None, steps=5, warmup=5):
        cfg = get_config()
        cfg.device = "cuda"
        if loss_name:
            cfg.loss.name = loss_name
        torch.manual_seed(0)
        model = mu.create_model(cfg, dev)
        state = {"model": model, "optimizer": ou.get_optimizer(model.parameters(), cfg), "n_iter": 0}
        step, loss = tu.get_train_step(cfg), lu.get_loss(cfg)
        mb = torch.randint(0, S_, (B,) + tuple(shape), device="cuda")
        for _ in range(warmup):
            step.step(state, loss, mb)
            state["n_iter"] += 1
        torch.cuda.synchronize()
        groups = []
        for _ in range(3):                         # median of three groups of `steps` steps (the first steps after the plans are
            t0 = time.perf_counter()               #  built still pay allocator and graph warm-up: 11.7 vs 9.6 ms seen for CIFAR)
            for _ in range(steps):
                step.step(state, loss, mb)
                state["n_iter"] += 1
            torch.cuda.synchronize()
            groups.append((time.perf_counter() - t0) / steps * 1e3)
        out[key] = {"workload": what, "loss": cfg.loss.name, "batch": B, "ms_per_step": round(sorted(groups)[1], 3),
                    "timing": f"median of 3 groups of {steps} steps after {warmup} warm-up steps"}
        del model, state
        torch.cuda.empty_cache()

    sample_rate("cifar10_unet_taul", "config 5 sampler: CIFAR-10 tauLDR U-Net (D=3072, S=256, logistic head), TauL", c10, 64, 20)
    sample_rate("maze_hollow_midpoint", "config 4: maze hollow transformer (D=225, S=3), MidPointTauL", maze, 128, 50, name="MidPointTauL")
    sample_rate("mnist_hollow_taul", "config 3 sampler: MNIST hollow transformer (D=784, S=256), TauL; engine_precision bf16x3 (default: logits "
                "within 4e-6 of the fp32 module's range, three bf16 products per contraction)", hmnist, 32, 10, name="TauL")
    sample_rate("mnist_hollow_taul_bf16", "the same with engine_precision bf16 (single bf16 operands: logits within 2e-3 of the range; bf16 step kernel)",
                hmnist, 32, 10, model_over={"engine_precision": "bf16"}, name="TauL")
    train_ms("mnist_hollow_catrmnll_train", "config 3: MNIST hollow transformer training step (CatRMNLL, reverse_prob logits)", hmnist,
             (1, 28, 28), 256, 32, loss_name="CatRMNLL")
    train_ms("cifar10_unet_ctelbolambda_train", "config 5: CIFAR-10 tauLDR U-Net training step (CTElboLambda)", c10, (3, 32, 32), 256, 32)
    return out


def cpu_baseline(model_gpu, cfg, batch, budget_s=12.0):
    """The CPU oracle (restatement of the reference's CPU path, pinned by tests/golden) doing the
    same tau-leaping step on the host cores: oracle U-Net forward + reverse rates + torch.poisson
    + update.  Bounded sample: `batch` samples, as many steps as fit in ~budget_s."""
    from oracle import ctmc_ops as ops, nets
    from oracle.forward_process import ForwardProcess
    # the GPU box gives one GPU's CPU share (16 cores) of a 256-thread host: use what we may run on
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))
    torch.set_num_threads(cores)
    m = cfg.model
    proc = ForwardProcess("gaussian", S, rate_sigma=m.rate_sigma, Q_sigma=m.Q_sigma, time_exp=m.time_exp, time_base=m.time_base)
    sd = {k: v.detach().float().cpu() for k, v in model_gpu.state_dict().items() if isinstance(v, torch.Tensor)}
    kw = dict(data_shape=list(cfg.data.shape), S=S, model_output=m.model_output, ch=m.ch, ch_mult=list(m.ch_mult),
              n_res_blocks=m.num_res_blocks, num_heads=m.num_heads, x_min_max=list(m.data_min_max))
    ts = ops.taul_time_grid(cfg.training.max_t, cfg.sampler.min_t, cfg.sampler.num_steps)
    g = torch.Generator().manual_seed(1)
    x = torch.randint(0, S, (batch, D), generator=g)
    done, t0 = 0, time.perf_counter()
    with torch.no_grad():
        while True:
            i = done
            t_ones = ts[i] * torch.ones((batch,))
            h = ts[i] - ts[i + 1]
            logits = nets.image_model_forward(sd, x, t_ones, **kw)
            rr, _ = ops.reverse_rates_ctelbo(logits, x, proc.transition(t_ones), proc.rate(t_ones), cfg.sampler.eps_ratio)
            jumps = torch.poisson(ops.zero_own_state(rr, x) * h)
            x = ops.tauleap_apply(x, jumps, cfg.sampler.is_ordinal)
            done += 1
            el = time.perf_counter() - t0
            if el > budget_s or done >= 600:
                break
    return {"value": round(batch * done / el, 2), "unit": "sample-steps/s", "cores": cores, "kind": "port",
            "sample": f"{done} tau-leaping steps of {batch} samples (oracle U-Net fwd + rates + torch.poisson + update), {el:.1f} s"}


def ddp_train_bench(a, world, rank, dev, dist):
    """`bench.py --gpus N --train`: K data-parallel training steps of the MNIST tauLDR U-Net (CT-ELBO, per-rank batch 64 =
    config_tauUnet_mnist's batch, global batch 64 N; cfg.distributed wraps the network in DistributedDataParallel: one gradient
    all-reduce per step over RCCL).  Same timing protocol as the sampling bench: W warm-up steps, barrier + synchronize, K
    steps, barrier + synchronize, max over ranks."""
    import lib.losses.losses  # noqa: F401
    import lib.losses.losses_utils as lu
    import lib.models.models  # noqa: F401
    import lib.models.model_utils as mu
    import lib.optimizers.optimizers  # noqa: F401
    import lib.optimizers.optimizers_utils as ou
    import lib.training.training  # noqa: F401
    import lib.training.training_utils as tu
    from config.mnist_config.config_tauUnet_mnist import get_config
    cfg = get_config()
    cfg.device = str(dev)
    cfg.distributed = world > 1
    B = 64
    torch.manual_seed(0)                                    # same initial weights on every rank (DDP broadcasts rank 0's anyway)
    model = mu.create_model(cfg, dev, rank=rank)
    state = {"model": model, "optimizer": ou.get_optimizer(model.parameters(), cfg), "n_iter": 0}
    step, loss = tu.get_train_step(cfg), lu.get_loss(cfg)
    g = torch.Generator(device=dev).manual_seed(100 + rank)  # each rank its own shard of the global minibatch
    mb = torch.randint(0, S, (B, 1, 28, 28), device=dev, generator=g)
    K, W = a.steps, max(a.warmup, 3)
    for _ in range(W):
        step.step(state, loss, mb)
        state["n_iter"] += 1
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        step.step(state, loss, mb)
        state["n_iter"] += 1
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
        torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if dist:
        tt = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt.item())
    if rank == 0:
        nparam = sum(p.numel() for p in model.parameters())
        print(json.dumps({
            "metric": "data-parallel training samples/s (MNIST tauLDR U-Net, CT-ELBO)", "value": round(B * world * K / el, 2), "unit": "samples/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(el / K * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic (random-init weights, random uint8 images resident in HBM)",
            "config": {"workload": "MNIST tauLDR CT-ELBO training step (Standard.step: noising, loss, backward, all-reduce, clip, Adam, EMA)",
                       "batch_per_gpu": B, "global_batch": B * world,
                       "parallelism": f"dp{world}: minibatch-sharded, one {nparam * 4 / 2 ** 20:.0f} MB fp32 gradient all-reduce per step (one bucket, after backward)"}}),
              flush=True)


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libctdd has no CPU path")
    ndev = torch.cuda.device_count()
    if local >= ndev and os.environ.get("CTDD_BENCH_SHARE_DEVICE") != "1":
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local} but only {ndev} GPU(s) visible")
    local_dev = local % ndev                 # (CTDD_BENCH_SHARE_DEVICE=1: rehearsal of the N>1 path on a one-GPU box)
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("CTDD_BENCH_BACKEND", "nccl")      # nccl = RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    if a.train:
        ddp_train_bench(a, world, rank, dev, dist)
        if dist:
            dist.barrier()
            dist.destroy_process_group()
        return
    cfg, model, sampler = build_model(dev)
    if a.engine_streams is not None:
        cfg.model.engine_streams = a.engine_streams
    for kv in a.model_opt:                     # e.g. --model-opt ring_min_tiles=160
        k_, v_ = kv.split("=")
        setattr(cfg.model, k_, int(v_))
    for kv in a.sampler_opt:                   # e.g. --sampler-opt pipeline_sub_batches=1
        k_, v_ = kv.split("=")
        setattr(cfg.sampler, k_, int(v_))
    sampler.seed = 42
    sampler.rank_stream = rank              # distinct Philox key per rank; no data-path collective
    K, W = a.steps, a.warmup
    assert W + K + 1 <= sampler.num_steps
    with torch.no_grad():
        st = sampler.begin(model, a.batch)
        for i in range(W):
            sampler.advance(st, i)
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(W, W + K):
            sampler.advance(st, i)
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
            torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if dist:
            tt = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        if rank == 0 and getattr(st, "parts", 1) > 1:
            # the roofline launches are timed one by one on the whole batch: a joined state advanced to the same step
            st = sampler.begin(model, a.batch, pipeline=False)
            for i in range(W + K):
                sampler.advance(st, i)
            torch.cuda.synchronize()
        roof = kernel_roofline(sampler, st, range(W + K, min(W + K + 20, sampler.num_steps - 1)), a.batch) if rank == 0 else None
        # the same kernel mid-trajectory (t ~ 0.5: fewer dimensions jump than next to t = 1, where the timed steps run)
        mid = sampler.num_steps // 2
        roof_mid = kernel_roofline(sampler, st, range(mid, mid + 20), a.batch, restore_state=True) if rank == 0 else None
        roof_net = network_roofline(model, a.batch, sampler) if rank == 0 else None
        fp32_mode = fp32_parity_mode(cfg, model, sampler, a.batch) if (rank == 0 and world == 1 and not a.no_fp32_mode) else None
    if rank == 0:
        value = a.batch * K * world / el
        line = {
            "metric": "tau-leaping sample-steps/s", "value": round(value, 2), "unit": "sample-steps/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(el / K * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "dtype_detail": "bf16 score network (fp32 accumulate, bf16 logits) + tau-leap step with f32 softmax / rates and ONE bf16 MFMA product for the S x S ratio contraction (relative rate error <= 3 * 2^-8; the three-product f32-parity step is timed in fp32_parity_mode)", "data": "synthetic (random-init weights, Gaussian initial state, resident in HBM)",
            "config": {"workload": "MNIST tauLDR U-Net TauL step (config_tauUnet_mnist: D=784, S=256, 1000-step grid)",
                       "batch_per_gpu": a.batch, "global_batch": a.batch * world, "D": D, "S": S,
                       "parallelism": f"sample-sharded x{world}, no collective in the loop"},
            "dims_per_s": round(value * D, 1),
            "roofline": roof,
            "roofline_network": roof_net,
            "roofline_mid_trajectory": None if roof_mid is None else {k_: roof_mid[k_] for k_ in ("bound", "achieved", "peak", "unit", "frac", "avg_launch_us", "fp32_logit_equivalent_frac", "mfma_view")},
            "fp32_parity_mode": fp32_mode,
        }
        if world == 1 and not a.no_train_step:
            line["train_step"] = train_step_timing()
            line["train_step_hollow"] = hollow_train_step_timing()
        if world == 1 and not a.no_configs:
            line["configs"] = baseline_configs()
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(model, cfg, a.cpu_batch)
        print(json.dumps(line), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
