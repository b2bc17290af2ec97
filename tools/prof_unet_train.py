"""profile target -- MNIST tauLDR CT-ELBO training steps (B = 64) on the HIP training plan, bf16."""
import sys, os
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_R, os.path.join(_R, 'continuous-time-diffusion-models-for-discrete-data_amd')]
import torch
import lib.models.models, lib.losses.losses, lib.training.training, lib.optimizers.optimizers  # noqa
import lib.models.model_utils as mu, lib.losses.losses_utils as lu, lib.training.training_utils as tu, lib.optimizers.optimizers_utils as ou
from config.mnist_config.config_tauUnet_mnist import get_config
cfg = get_config()
cfg.model.engine_precision = sys.argv[1] if len(sys.argv) > 1 else "bf16"
torch.manual_seed(0)
model = mu.create_model(cfg, torch.device("cuda"))
state = {"model": model, "optimizer": ou.get_optimizer(model.parameters(), cfg), "n_iter": 0}
step, loss = tu.get_train_step(cfg), lu.get_loss(cfg)
mb = torch.randint(0, 256, (64, 1, 28, 28), device="cuda")
for _ in range(14):
    l = step.step(state, loss, mb); state["n_iter"] += 1
torch.cuda.synchronize()
print(float(l))
