"""Forward-only: two 128-sample plans replayed on two streams -- joined every iteration (as the engine does), free-running
import os, sys
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_R, os.path.join(_R, "continuous-time-diffusion-models-for-discrete-data_amd")]
in lockstep, free-running with stream 2 delayed by a fraction of a forward."""
import sys, time
import torch
import lib.models.models  # noqa
import lib.models.model_utils as mu
from config.mnist_config.config_tauUnet_mnist import get_config
from ctdd.unet_engine import UNetEngine
cfg = get_config()
model = mu.create_model(cfg, torch.device("cuda")); model.eval()
eng = UNetEngine(model, precision="bf16")
Bs, IT = 128, 40
x = torch.randint(0, 256, (Bs, 784), device="cuda"); t = torch.full((Bs,), 0.5, device="cuda")
subs = [eng._prepare(Bs, torch.int64, x, t) for _ in range(2)]
s = [torch.cuda.Stream(), torch.cuda.Stream()]
def one(st):
    st.graph.replay()
# single forward duration alone
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(IT): one(subs[0])
torch.cuda.synchronize(); alone = (time.perf_counter() - t0) / IT
print(f"one 128-sample forward alone: {alone*1e3:.3f} ms", flush=True)
def run(mode, delay_frac=0.0):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if mode == "join":
        for _ in range(IT):
            evs = []
            for k in range(2):
                with torch.cuda.stream(s[k]):
                    one(subs[k]); e = torch.cuda.Event(); e.record(s[k]); evs.append(e)
            for k in range(2):
                for e in evs: s[k].wait_event(e)
    else:
        if delay_frac > 0:
            with torch.cuda.stream(s[1]):
                torch.cuda._sleep(int(delay_frac * alone / 0.68e-9))
        for _ in range(IT):
            for k in range(2):
                with torch.cuda.stream(s[k]):
                    one(subs[k])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / IT
# calibrate _sleep
torch.cuda.synchronize(); t0 = time.perf_counter(); torch.cuda._sleep(10_000_000); torch.cuda.synchronize()
print(f"_sleep(1e7) = {(time.perf_counter()-t0)*1e3:.2f} ms", flush=True)
for rep in range(2):
    print(f"join every iteration: {run('join')*1e3:.3f} ms per pair", flush=True)
    print(f"free, lockstep:       {run('free')*1e3:.3f}", flush=True)
    for f in (0.25, 0.5, 0.75):
        print(f"free, delay {f}:     {run('free', f)*1e3:.3f}", flush=True)
