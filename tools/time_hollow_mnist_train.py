"""MNIST hollow (E=256, 2x9 blocks, D=784, S=256; BASELINE config 3 with CatRMNLL) training step: torch vs HIP."""
import sys, time
import os
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_R, os.path.join(_R, "continuous-time-diffusion-models-for-discrete-data_amd")]
import torch
import lib.models.models, lib.losses.losses, lib.training.training, lib.optimizers.optimizers  # noqa
import lib.models.model_utils as mu, lib.losses.losses_utils as lu, lib.training.training_utils as tu, lib.optimizers.optimizers_utils as ou
from config.mnist_config.config_hollow_mnist import get_config
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
for engine in ("hip", "torch"):
    cfg = get_config(); cfg.device = "cuda"; cfg.model.engine = engine
    cfg.loss.name = "CatRMNLL"
    torch.manual_seed(0)
    model = mu.create_model(cfg, torch.device("cuda"))
    state = {"model": model, "optimizer": ou.get_optimizer(model.parameters(), cfg), "n_iter": 0}
    step, loss = tu.get_train_step(cfg), lu.get_loss(cfg)
    mb = torch.randint(0, 256, (B, 1, 28, 28), device="cuda")
    for _ in range(3): l = step.step(state, loss, mb); state["n_iter"] += 1
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): l = step.step(state, loss, mb); state["n_iter"] += 1
    torch.cuda.synchronize()
    print("%-5s B=%d: %.2f ms/step  loss %.3f  trainer=%s" % (engine, B, (time.perf_counter() - t0) / 5 * 1e3, float(l), getattr(model, "_trainer", None) is not None), flush=True)
    del model, state
