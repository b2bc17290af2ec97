"""profile target -- maze hollow ScoreElbo training steps on the HIP path."""
import sys
import os
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_R, os.path.join(_R, "continuous-time-diffusion-models-for-discrete-data_amd")]
import torch
import lib.models.models  # noqa
import lib.losses.losses  # noqa
import lib.models.model_utils as mu
import lib.losses.losses_utils as lu
from config.maze_config.config_hollow_maze import get_config
B = 128
cfg = get_config(); cfg.device = "cuda"
cfg.model.engine_train_precision = sys.argv[1] if len(sys.argv) > 1 else "bf16"
import os
if "P_ATT" in os.environ: cfg.model.attention_dropout_rate = float(os.environ["P_ATT"])
if "P_DROP" in os.environ: cfg.model.dropout_rate = float(os.environ["P_DROP"])
torch.manual_seed(0)
model = mu.create_model(cfg, torch.device("cuda"))
loss_fn = lu.get_loss(cfg)
x = torch.randint(0, 3, (B, 225), device="cuda")
state = {"model": model, "n_iter": 0}
for _ in range(10):
    for p in model.parameters(): p.grad = None
    l = loss_fn.calc_loss(x, state); l.backward()
torch.cuda.synchronize()
print(float(l))
