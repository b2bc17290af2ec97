"""GPU: K28 fused clip + Adam + EMA update (csrc/optim.hip through lib/optimizers + lib/training) against
the reference's literal sequence on torch ops: clip_grad_norm_ -> torch.optim.Adam.step -> EMA lerp
(lib/training/training.py:17-40, lib/models/models.py:745-758)."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _make(seed, shapes):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return [torch.nn.Parameter(torch.randn(s, generator=g, device="cuda")) for s in shapes]


@pytest.mark.parametrize("max_norm,use_ema", [(1.0, True), (0.0, True), (0.05, False)])
def test_fused_adam_ema_matches_torch_sequence(max_norm, use_ema):
    from lib.optimizers.optimizers import FusedAdam
    shapes = [(96, 96, 3, 3), (96,), (17,), (40000,), (3, 5, 7), (1,)]          # > one 16 Ki chunk, ragged tails, a scalar
    pa, pb = _make(0, shapes), _make(0, shapes)
    pa.append(torch.nn.Parameter(torch.ones(5, device="cuda")))                 # a parameter that never gets a gradient
    pb.append(torch.nn.Parameter(torch.ones(5, device="cuda")))
    ref_opt = torch.optim.Adam(pa, 2e-3)
    opt = FusedAdam(pb, 2e-3)
    sh_a = [p.detach().clone() for p in pa]
    sh_b = [p.detach().clone() for p in pb]
    g = torch.Generator(device="cuda").manual_seed(7)
    n_upd = 0
    for it in range(4):
        grads = [torch.randn(s, generator=g, device="cuda") * (10.0 if it == 1 else 0.1) for s in shapes]
        for plist in (pa, pb):
            for p, gr in zip(plist, grads):
                p.grad = gr.clone()
        n_upd += 1
        decay = min(0.999, (1 + n_upd) / (10 + n_upd))
        # reference sequence
        if max_norm > 0:
            torch.nn.utils.clip_grad_norm_(pa, max_norm)
        ref_opt.step()
        if use_ema:
            with torch.no_grad():
                for s_, p in zip(sh_a, pa):
                    s_.sub_((1 - decay) * (s_ - p))                             # models.py:755-758
        # fused
        opt.fused_step(max_norm, sh_b if use_ema else None, decay, pb)
    torch.cuda.synchronize()
    for a, b in zip(pa, pb):
        torch.testing.assert_close(b, a, rtol=2e-5, atol=1e-7)
    for a, b in zip(sh_a, sh_b):
        torch.testing.assert_close(b, a, rtol=2e-5, atol=1e-7)
    # optimizer state has torch.optim.Adam's layout and values
    sa, sb = ref_opt.state_dict(), opt.state_dict()
    assert sa["param_groups"][0].keys() == sb["param_groups"][0].keys()
    for k in sa["state"]:
        assert float(sa["state"][k]["step"]) == float(sb["state"][k]["step"]) == 4.0
        torch.testing.assert_close(sb["state"][k]["exp_avg"], sa["state"][k]["exp_avg"], rtol=2e-5, atol=1e-8)
        torch.testing.assert_close(sb["state"][k]["exp_avg_sq"], sa["state"][k]["exp_avg_sq"], rtol=2e-5, atol=1e-10)
    assert 6 not in sb["state"]                                                 # the gradient-less parameter has no state, as in torch


def test_train_step_uses_fused_update_on_gpu():
    """Standard.step on a GPU model: same weights / EMA as the torch-op sequence after three steps."""
    import lib.models.models  # noqa: F401
    import lib.models.model_utils as mu
    import lib.losses.losses  # noqa: F401
    import lib.training.training  # noqa: F401
    import lib.training.training_utils as tu
    import lib.optimizers.optimizers  # noqa: F401
    import lib.optimizers.optimizers_utils as ou
    from config.synthetic_config.config_hollow_synthetic import get_config

    class SqLoss:
        def calc_loss(self, state, minibatch, label=None):
            return sum((p.float() ** 2).sum() for p in state["model"].parameters()) * 1e-3

    cfg = get_config()
    cfg.device = "cuda"
    torch.manual_seed(0)
    model = mu.create_model(cfg, torch.device("cuda"))
    ref = copy.deepcopy(model)
    ref.shadow_params = [s.clone() for s in model.shadow_params]
    state = {"model": model, "optimizer": ou.get_optimizer(model.parameters(), cfg), "n_iter": 0}
    step = tu.get_train_step(cfg)
    ref_opt = torch.optim.Adam(ref.parameters(), cfg.optimizer.lr)
    mb = torch.zeros(4, 32, dtype=torch.long, device="cuda")
    for it in range(3):
        step.step(state, SqLoss(), mb)
        state["n_iter"] += 1
        ref_opt.zero_grad()
        (sum((p.float() ** 2).sum() for p in ref.parameters()) * 1e-3).backward()
        if cfg.training.clip_grad:
            torch.nn.utils.clip_grad_norm_(ref.parameters(), cfg.training.grad_norm)
        if cfg.training.warmup > 0:
            for g_ in ref_opt.param_groups:
                g_["lr"] = cfg.optimizer.lr * np.minimum(it / cfg.training.warmup, 1.0)
        ref_opt.step()
        ref.update_ema()
    assert model.num_updates == ref.num_updates == 3
    for a, b in zip(ref.parameters(), model.parameters()):
        torch.testing.assert_close(b, a, rtol=5e-5, atol=1e-7)
    for a, b in zip(ref.shadow_params, model.shadow_params):
        torch.testing.assert_close(b, a, rtol=5e-5, atol=1e-7)


def test_fused_adam_mixed_step_counts_and_param_groups():
    """torch.optim.Adam keeps `step` per parameter and clip_grad_norm_ takes ONE norm over all parameters: a parameter
    that gets its first gradient later (a sub-module that becomes used, a tensor unfrozen mid-run) and a second param
    group with another learning rate must still follow the torch sequence."""
    from lib.optimizers.optimizers import FusedAdam
    shapes = [(64, 33), (129,), (20000,), (7, 9)]
    pa, pb = _make(3, shapes), _make(3, shapes)
    groups = lambda ps: [{"params": ps[:2]}, {"params": ps[2:], "lr": 5e-4}]
    ref_opt = torch.optim.Adam(groups(pa), 2e-3)
    opt = FusedAdam(groups(pb), 2e-3)
    g = torch.Generator(device="cuda").manual_seed(11)
    for it in range(5):
        grads = [torch.randn(s, generator=g, device="cuda") * 0.3 for s in shapes]
        for plist in (pa, pb):
            for i, (p, gr) in enumerate(zip(plist, grads)):
                late = i in (1, 3) and it < 2                                   # these two see their first gradient at it = 2
                p.grad = None if late else gr.clone()
        torch.nn.utils.clip_grad_norm_(pa, 0.5)
        ref_opt.step()
        opt.fused_step(0.5, None, -1.0, pb)
    torch.cuda.synchronize()
    for a, b in zip(pa, pb):
        torch.testing.assert_close(b, a, rtol=3e-5, atol=1e-7)
    sa, sb = ref_opt.state_dict()["state"], opt.state_dict()["state"]
    assert [float(sb[k]["step"]) for k in sorted(sb)] == [float(sa[k]["step"]) for k in sorted(sa)] == [5.0, 3.0, 5.0, 3.0]
