"""Sampling throughput of the other BASELINE configs (random-init weights), sample-steps/s."""
import os, sys
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_R, os.path.join(_R, "continuous-time-diffusion-models-for-discrete-data_amd")]
import sys, time
import torch, numpy as np
import lib.models.models, lib.sampling.sampling  # noqa
import lib.models.model_utils as mu, lib.sampling.sampling_utils as su
def run(name, get_config, N, steps, **over):
    cfg = get_config(); cfg.device = "cuda"
    cfg.sampler.num_steps = steps
    for k, v in over.items(): setattr(cfg.sampler, k, v)
    torch.manual_seed(0)
    model = mu.create_model(cfg, torch.device("cuda")); model.eval()
    s = su.get_sampler(cfg)
    s.sample(model, N); torch.cuda.synchronize()
    t0 = time.perf_counter(); s.sample(model, N); torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(f"{name}: sampler {cfg.sampler.name} N={N} steps={steps}: {el:.2f} s -> {N*steps/el:.0f} sample-steps/s ({el/steps*1e3:.2f} ms/step)")
from config.cifar10_config.config_tauUnet_cifar10 import get_config as c10
from config.maze_config.config_hollow_maze import get_config as maze
from config.synthetic_config.config_hollow_synthetic import get_config as syn
from config.mnist_config.config_hollow_mnist import get_config as hm
run("CIFAR-10 tauLDR U-Net (D=3072, S=256, logistic head)", c10, 64, 20)
run("maze hollow (D=225, S=3)", maze, 128, 50)
run("synthetic hollow (D=32, S=2)", syn, 1024, 50)
run("MNIST hollow (D=784, S=256)", hm, 32, 10)
