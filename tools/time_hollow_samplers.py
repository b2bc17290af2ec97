"""Sampling throughput of the two hollow BASELINE configs (MNIST TauL N = 32, maze MidPointTauL N = 128), median of three calls."""
import os, sys, time
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_R, os.path.join(_R, "continuous-time-diffusion-models-for-discrete-data_amd")]
import torch
import lib.models.models, lib.sampling.sampling  # noqa
import lib.models.model_utils as mu, lib.sampling.sampling_utils as su
from config.maze_config.config_hollow_maze import get_config as maze
from config.mnist_config.config_hollow_mnist import get_config as hm
def run(name, get_config, N, steps, sampler):
    cfg = get_config(); cfg.device = "cuda"; cfg.sampler.num_steps = steps; cfg.sampler.name = sampler
    torch.manual_seed(0)
    model = mu.create_model(cfg, torch.device("cuda")); model.eval()
    s = su.get_sampler(cfg); s.seed = 1
    s.sample(model, N); torch.cuda.synchronize()
    els = []
    for _ in range(3):
        t0 = time.perf_counter(); s.sample(model, N); torch.cuda.synchronize(); els.append(time.perf_counter() - t0)
    el = sorted(els)[1]
    print(f"{name}: {sampler} N={N} steps={steps}: {N*steps/el:.0f} sample-steps/s ({el/steps*1e3:.2f} ms/step)", flush=True)
run("MNIST hollow", hm, 32, 10, "TauL")
run("maze hollow", maze, 128, 50, "MidPointTauL")
