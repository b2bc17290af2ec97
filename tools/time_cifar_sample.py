"""CIFAR-10 tauLDR U-Net TauL sampling (BASELINE config 5's sampler), timed like bench.py's `configs` block, three repeats."""
import os, sys, time
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_R, os.path.join(_R, "continuous-time-diffusion-models-for-discrete-data_amd")]
import torch
import lib.models.models, lib.sampling.sampling  # noqa
import lib.models.model_utils as mu, lib.sampling.sampling_utils as su
from config.cifar10_config.config_tauUnet_cifar10 import get_config
N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
cfg = get_config(); cfg.device = "cuda"; cfg.sampler.num_steps = steps
torch.manual_seed(0)
model = mu.create_model(cfg, torch.device("cuda")); model.eval()
s = su.get_sampler(cfg); s.seed = 1
s.sample(model, N); torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter(); s.sample(model, N); torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(f"CIFAR TauL N={N} steps={steps}: {el*1e3:.1f} ms -> {N*steps/el:.0f} sample-steps/s ({el/steps*1e3:.2f} ms/step)", flush=True)
