"""fp32 HIP gradients and fp32 torch gradients of the maze hollow net against a float64 evaluation of the module."""
import copy, sys
import os
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_R, os.path.join(_R, "continuous-time-diffusion-models-for-discrete-data_amd")]
import torch
import lib.models.models  # noqa
import lib.models.model_utils as mu
from config.maze_config.config_hollow_maze import get_config
from ctdd.hollow_train import HollowTrainer

cfg = get_config(); cfg.device = "cuda"
cfg.model.update(dropout_rate=0.0, attention_dropout_rate=0.0)
torch.manual_seed(0)
model = mu.create_model(cfg, torch.device("cuda"))
x = torch.randint(0, 3, (4, 225), device="cuda"); t = torch.tensor([0.02, 0.3, 0.5, 0.99], device="cuda")
wgt = torch.randn((4, 225, 3), device="cuda")
def grads(f, w=wgt):
    for p in model.parameters(): p.grad = None
    o = f(x, t); (o * w).sum().backward()
    return {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
cfg.model.engine = "torch"
g_t = grads(model)
g_h = grads(HollowTrainer(model, precision="fp32"))
g_b = grads(HollowTrainer(model, precision="bf16"))
net64 = copy.deepcopy(model.net).double()
net64.input_embedding.register_forward_pre_hook(lambda m, a: (a[0].double(),))
torch.set_default_dtype(torch.float64)
o = net64(x, t.double()); (o * wgt.double()).sum().backward()
g64 = {"net." + n: p.grad for n, p in net64.named_parameters() if p.grad is not None}
torch.set_default_dtype(torch.float32)
floor = 1e-3 * max(float(v.abs().max()) for v in g64.values())
rows = []
for n, r in g64.items():
    sc = max(float(r.abs().max()), floor)
    rows.append((float((g_h[n].double() - r).abs().max()) / sc, float((g_t[n].double() - r).abs().max()) / sc,
                 float((g_b[n].double() - r).norm() / max(float(r.norm()), floor)), n))
rows.sort(reverse=True)
print("worst HIP fp32 err, torch fp32 err, bf16 l2 err")
for r in rows[:12]: print("%.3e %.3e %.3e %s" % r)
l2 = lambda g: max(float((g[n].double() - r).norm() / max(float(r.norm()), floor)) for n, r in g64.items())
print("fp32 relative L2 errors: hip %.3e torch %.3e" % (l2(g_h), l2(g_t)))
print("max torch err %.3e, max hip err %.3e, max bf16 l2 %.3e" % (max(r[1] for r in rows), max(r[0] for r in rows), max(r[2] for r in rows)))
