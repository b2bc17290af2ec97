"""Per-launch table of the hollow transformer's inference plan: label, microseconds alone (HIP events), share.

    python tools/hollow_table.py [--config mnist|maze] [--batch 32] [--precision bf16x3|bf16|fp32|mixed]
"""
import argparse
import os
import sys

_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_R, os.path.join(_R, "continuous-time-diffusion-models-for-discrete-data_amd")]
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="mnist")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--precision", default="bf16x3")
    ap.add_argument("--single", default="", help="comma-separated linear names that take one bf16 product in the bf16x3 mode")
    ap.add_argument("--quiet", action="store_true")
    a = ap.parse_args()
    import lib.models.models  # noqa: F401
    import lib.models.model_utils as mu
    from ctdd.hollow_engine import HollowEngine
    if a.config == "mnist":
        from config.mnist_config.config_hollow_mnist import get_config
    else:
        from config.maze_config.config_hollow_maze import get_config
    cfg = get_config()
    cfg.device = "cuda"
    if a.single:
        cfg.model.engine_bf16_linears = [v.strip() for v in a.single.split(",")]
    torch.manual_seed(0)
    model = mu.create_model(cfg, torch.device("cuda"))
    model.eval()
    eng = HollowEngine(model, precision=a.precision)
    D, S = int(cfg.model.concat_dim), int(cfg.data.S)
    x = torch.randint(0, S, (a.batch, D), device="cuda")
    t = torch.rand(a.batch, device="cuda") * 0.9 + 0.05
    with torch.no_grad():
        out = eng(x, t).float().clone()
        cfg.model.engine = "torch"
        ref = model(x, t).float()
        cfg.model.engine = "hip"
    rng = float(ref.max() - ref.min())
    print(f"{a.config} batch {a.batch} precision {a.precision}: max |logit error| vs the fp32 module = {float((out - ref).abs().max()):.3e} "
          f"({float((out - ref).abs().max()) / rng:.2e} of the logit range {rng:.3f})")
    (st,) = list(eng._plans.values())
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
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.no_grad():
        eng(x, t)
        e0.record()
        for _ in range(10):
            eng(x, t)
        e1.record()
        e1.synchronize()
    print(f"{len(rows)} launches, summed alone {tot:.0f} us, forward (graph replay) {e0.elapsed_time(e1) * 100:.0f} us  [single: {a.single or '-'}]")
    if a.quiet:
        return
    agg = {}
    for (fn, lab), us, fl in rows:
        import re
        key = (fn.replace("ctdd_", ""), re.sub(r"(l2r|r2l)\.\d+", r"\1.*", str(lab)))
        c = agg.setdefault(key, [0, 0.0, 0])
        c[0] += 1; c[1] += us; c[2] += fl
    for (fn, lab), (n, us, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
        print(f"{us:8.1f} us {100 * us / tot:5.1f}%  x{n:<3d} {us / n:7.1f} us each  {fl / us / 1e6 if fl else 0:7.1f} TF/s  {fn:22s} {lab}")


if __name__ == "__main__":
    main()
