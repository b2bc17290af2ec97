"""Per-launch table of the U-Net inference plan (one sub-batch plan as the sampler loop replays it): label, microseconds when the
launch runs alone (HIP events, 5 repetitions), matrix TFLOP/s, share of the summed time.

    python tools/conv_table.py [--batch 128] [--config mnist|cifar10]
"""
import argparse
import os
import sys

_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_R, os.path.join(_R, "continuous-time-diffusion-models-for-discrete-data_amd")]
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--config", default="mnist")
    ap.add_argument("--model-opt", action="append", default=[])
    a = ap.parse_args()
    import lib.models.models  # noqa: F401
    import lib.models.model_utils as mu
    from ctdd.unet_engine import UNetEngine
    if a.config == "mnist":
        from config.mnist_config.config_tauUnet_mnist import get_config
    else:
        from config.cifar10_config.config_tauUnet_cifar10 import get_config
    cfg = get_config()
    cfg.device = "cuda"
    for kv in a.model_opt:
        k_, v_ = kv.split("=")
        setattr(cfg.model, k_, int(v_))
    cfg.model.engine_streams = 1
    torch.manual_seed(0)
    model = mu.create_model(cfg, torch.device("cuda"))
    model.eval()
    eng = UNetEngine(model, precision="bf16")
    C_, H_, W_ = cfg.data.shape
    x = torch.randint(0, cfg.data.S, (a.batch, C_ * H_ * W_), device="cuda")
    t = torch.full((a.batch,), 0.5, device="cuda")
    with torch.no_grad():
        eng(x, t, logits_bf16=(cfg.model.model_output == "logits"))
    (st,) = [v for v in eng._plans.values() if not isinstance(v, tuple)]
    rows = []
    for step in st.plan:
        step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            step()
        e1.record()
        e1.synchronize()
        rows.append((step.label, e0.elapsed_time(e1) * 1e3 / 5, getattr(step, "flops", 0)))
    tot = sum(r[1] for r in rows)
    # graph replay of the whole plan for comparison
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.no_grad():
        eng(x, t, logits_bf16=(cfg.model.model_output == "logits"))
        e0.record()
        for _ in range(10):
            eng(x, t, logits_bf16=(cfg.model.model_output == "logits"))
        e1.record()
        e1.synchronize()
    print(f"batch {a.batch}: {len(rows)} launches, summed alone {tot:.0f} us, graph replay {e0.elapsed_time(e1) * 100:.0f} us, "
          f"matrix {sum(r[2] for r in rows) / 1e9:.1f} GFLOP")
    agg = {}
    for (fn, lab), us, fl in rows:
        key = (fn.replace("ctdd_unet_", ""), lab)
        c = agg.setdefault(key, [0, 0.0, 0])
        c[0] += 1; c[1] += us; c[2] += fl
    for (fn, lab), (n, us, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{us:8.1f} us {100 * us / tot:5.1f}%  x{n:<2d} {us / n:7.1f} us each  {fl / us / 1e6 if fl else 0:7.1f} TF/s  {fn:12s} {lab}")


if __name__ == "__main__":
    main()
