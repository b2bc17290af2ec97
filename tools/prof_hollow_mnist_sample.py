"""profile target -- MNIST hollow transformer (BASELINE config 3) TauL sampling, N = 32, 10 steps, default engine precision (bf16x3)."""
import os, sys
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_R, os.path.join(_R, "continuous-time-diffusion-models-for-discrete-data_amd")]
import torch
import lib.models.models, lib.sampling.sampling  # noqa
import lib.models.model_utils as mu, lib.sampling.sampling_utils as su
from config.mnist_config.config_hollow_mnist import get_config
cfg = get_config(); cfg.device = "cuda"; cfg.sampler.num_steps = 10; cfg.sampler.name = "TauL"
if len(sys.argv) > 1:
    cfg.model.engine_precision = sys.argv[1]
torch.manual_seed(0)
model = mu.create_model(cfg, torch.device("cuda")); model.eval()
s = su.get_sampler(cfg); s.seed = 1
for _ in range(3):
    s.sample(model, 32)
torch.cuda.synchronize()
