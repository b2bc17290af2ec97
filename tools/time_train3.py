"""MNIST tauLDR CT-ELBO training step (B = 64): network on torch autograd ops vs the hand-written training plan."""
import sys, time
import os
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_R, os.path.join(_R, 'continuous-time-diffusion-models-for-discrete-data_amd')]
import torch
import lib.models.models, lib.losses.losses, lib.training.training, lib.optimizers.optimizers  # noqa
import lib.models.model_utils as mu, lib.losses.losses_utils as lu, lib.training.training_utils as tu, lib.optimizers.optimizers_utils as ou
from config.mnist_config.config_tauUnet_mnist import get_config

def run(tag, engine, precision="bf16", B=64, steps=10, detail=False, **over):
    cfg = get_config()
    cfg.model.engine, cfg.model.engine_precision = engine, precision
    for k, v in over.items():
        setattr(cfg.model, k, v)
    torch.manual_seed(0)
    model = mu.create_model(cfg, torch.device("cuda"))
    state = {"model": model, "optimizer": ou.get_optimizer(model.parameters(), cfg), "n_iter": 0}
    step, loss = tu.get_train_step(cfg), lu.get_loss(cfg)
    mb = torch.randint(0, 256, (B, 1, 28, 28), device="cuda")
    for _ in range(4): step.step(state, loss, mb); state["n_iter"] += 1
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): l = step.step(state, loss, mb); state["n_iter"] += 1
    torch.cuda.synchronize()
    print(tag, "%.2f ms/step, loss %.4f" % ((time.perf_counter() - t0) / steps * 1e3, float(l)), flush=True)
    if detail and engine == "hip":
        eng = model._engine
        st = next(iter(eng._train_plans.values()))[0]
        x = torch.randint(0, 256, (B, 784), device="cuda"); t = torch.rand(B, device="cuda")
        def timeit(f, n=10):
            f(); torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(n): f()
            torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
        if st.fgraph is not None:
            print("  forward graph replay %.2f ms, backward graph replay %.2f ms" % (timeit(st.fgraph.replay), timeit(st.bgraph.replay)))
        if "-q" in sys.argv:
            return
        print("  forward plan (eager launches) %.2f ms" % timeit(lambda: eng._run_plan(st)))
        def bwd():
            st.tc.zbuf.zero_(); st.tc.bzpool.zero_(); st.tc.gflat.zero_()
            for s in st.bwd_plan: s()
        print("  backward plan (eager launches) %.2f ms, %d launches fwd / %d bwd" % (timeit(bwd), len(st.plan), len(st.bwd_plan)))
        # per-launch times of the backward plan
        rows = []
        for s in st.bwd_plan:
            s(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3): s()
            e1.record(); e1.synchronize()
            rows.append((e0.elapsed_time(e1) / 3 * 1e3, s.label, getattr(s, "flops", 0)))
        agg = {}
        for us, lab, fl in rows:
            k = lab[0]
            a = agg.setdefault(k, [0, 0.0, 0]); a[0] += 1; a[1] += us; a[2] += fl
        for k, (n, us, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            print("    %-40s %3d launches %8.1f us  %s" % (k, n, us, ("%.0f TFLOP/s" % (fl / us / 1e6)) if fl else ""))
        if "-v" in sys.argv:
            for us, lab, fl in sorted(rows, key=lambda r: -r[0])[:25]:
                print("      %8.1f us %s %s" % (us, lab, ("%.0f TF" % (fl / us / 1e6)) if fl else ""))

which = sys.argv[1] if len(sys.argv) > 1 else "all"
_over = {a.split("=")[0]: int(a.split("=")[1]) for a in sys.argv[2:] if "=" in a}     # cfg.model knobs, e.g. gn_onepass_train=0
if which in ("all", "torch"):
    run("torch autograd network      ", "torch")
if which in ("all", "hip"):
    run("HIP training plan, bf16     ", "hip", "bf16", detail=True, **_over)
if which in ("all", "fp32"):
    run("HIP training plan, fp32     ", "hip", "fp32")
