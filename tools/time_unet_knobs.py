"""MNIST U-Net training step (B=64, bf16 plan) over weight-gradient scheduling knobs."""
import sys, os, time
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_R, os.path.join(_R, 'continuous-time-diffusion-models-for-discrete-data_amd')]
import torch
import lib.models.models, lib.losses.losses, lib.training.training, lib.optimizers.optimizers  # noqa
import lib.models.model_utils as mu, lib.losses.losses_utils as lu, lib.training.training_utils as tu, lib.optimizers.optimizers_utils as ou
from config.mnist_config.config_tauUnet_mnist import get_config
def run(**over):
    cfg = get_config()
    for k, v in over.items(): setattr(cfg.model, k, v)
    torch.manual_seed(0)
    model = mu.create_model(cfg, torch.device("cuda"))
    state = {"model": model, "optimizer": ou.get_optimizer(model.parameters(), cfg), "n_iter": 0}
    step, loss = tu.get_train_step(cfg), lu.get_loss(cfg)
    mb = torch.randint(0, 256, (64, 1, 28, 28), device="cuda")
    for _ in range(4): step.step(state, loss, mb); state["n_iter"] += 1
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): l = step.step(state, loss, mb); state["n_iter"] += 1
    torch.cuda.synchronize()
    print(over, "%.3f ms/step" % ((time.perf_counter() - t0) / 20 * 1e3), flush=True)
import sys
if len(sys.argv) > 1:                                   # knob=value pairs, one run per argument (e.g. gn_onepass_train=0)
    for arg in sys.argv[1:]:
        k, v = arg.split("=")
        run(**{k: int(v)})
else:
    for w in (1, 2, 3, 4):
        for ov in (8, 24, 48):
            run(wgrad_wgs_per_cu=w, wgrad_chunk_overhead=ov)
