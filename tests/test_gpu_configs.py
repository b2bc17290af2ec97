"""GPU: the five BASELINE configurations end to end through the registries -- build the
model, draw samples with the configured sampler, take one training step with the configured loss."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _registries():
    import lib.models.models  # noqa: F401
    import lib.models.model_utils as mu
    import lib.sampling.sampling  # noqa: F401
    import lib.sampling.sampling_utils as su
    import lib.losses.losses  # noqa: F401
    import lib.losses.losses_utils as lu
    import lib.training.training  # noqa: F401
    import lib.training.training_utils as tu
    import lib.optimizers.optimizers  # noqa: F401
    import lib.optimizers.optimizers_utils as ou
    return mu, su, lu, tu, ou


def _one_step(cfg, model, minibatch):
    _, _, lu, tu, ou = _registries()
    state = {"model": model, "optimizer": ou.get_optimizer(model.parameters(), cfg), "n_iter": 0}
    w0 = [p.detach().clone() for p in model.parameters()]
    out = tu.get_train_step(cfg).step(state, lu.get_loss(cfg), minibatch)
    assert out.dim() == 0 and torch.isfinite(out) and float(out) < 1e8
    moved = sum(int(not torch.equal(a, b)) for a, b in zip(w0, model.parameters()))
    assert moved > 0.5 * len(w0) and model.num_updates == 1
    return float(out)


def test_cifar10_unet_logistic_head_engine_and_sampling():
    """config_tauUnet_cifar10: 34.4 M parameters, logistic head, D=3072; engine vs autograd module, then TauL."""
    mu, su, _, _, _ = _registries()
    from config.cifar10_config.config_tauUnet_cifar10 import get_config
    from ctdd.unet_engine import UNetEngine
    cfg = get_config()
    cfg.sampler.num_steps = 3
    torch.manual_seed(0)
    model = mu.create_model(cfg, torch.device("cuda"))
    assert sum(p.numel() for p in model.parameters()) == 34_431_366          # 34.43 M
    g = torch.Generator(device="cuda").manual_seed(1)
    with torch.no_grad():
        for name, p in model.named_parameters():                 # the reference zero-scales some convs at init: re-draw
            if p.dim() > 1:
                p.copy_(torch.randn(p.shape, generator=g, device="cuda") / (p[0].numel() ** 0.5))
    model.init_ema()
    model.eval()
    x = torch.randint(0, 256, (2, 3072), device="cuda")
    t = torch.tensor([0.2, 0.9], device="cuda")
    with torch.no_grad():
        cfg.model.engine = "torch"
        ref = model(x, t).float().cpu()
        out = UNetEngine(model, precision="fp32")(x.view(2, 3, 32, 32), t).cpu()
        fast = UNetEngine(model, precision="bf16")(x.view(2, 3, 32, 32), t).cpu()
    assert ref.shape == out.shape == (2, 3072, 256)
    pr, po, pf = torch.softmax(ref, -1), torch.softmax(out, -1), torch.softmax(fast, -1)
    assert (po - pr).abs().max().item() < 5e-5                   # logistic head: compare bin probabilities (see test_gpu_unet)
    assert (pf - pr).abs().max().item() < 5e-2
    cfg.model.engine = "hip"
    samples, change_dim = su.get_sampler(cfg).sample(model, 4)
    assert samples.shape == (4, 3072) and samples.min() >= 0 and samples.max() <= 255 and len(change_dim) == 3
    model.train()


@pytest.mark.parametrize("which", ["maze", "synthetic", "hollow_mnist"])
def test_hollow_configs_sample_and_train(which):
    mu, su, _, _, _ = _registries()
    if which == "maze":
        from config.maze_config.config_hollow_maze import get_config
    elif which == "synthetic":
        from config.synthetic_config.config_hollow_synthetic import get_config
    else:
        from config.mnist_config.config_hollow_mnist import get_config
    cfg = get_config()
    cfg.device = "cuda"
    cfg.sampler.num_steps = 4
    if which == "hollow_mnist":                                  # BASELINE config (3): CatRMNLL with nll_weight 0.01
        cfg.loss.name, cfg.loss.nll_weight = "CatRMNLL", 0.01
    torch.manual_seed(0)
    model = mu.create_model(cfg, torch.device("cuda"))
    D, S = int(np.prod(cfg.data.shape)), cfg.data.S
    model.eval()
    out = su.get_sampler(cfg).sample(model, 4)
    samples = out[0] if isinstance(out, tuple) else out
    assert samples.shape == (4, D) and samples.min() >= 0 and samples.max() < S
    model.train()
    mb = torch.randint(0, S, (3, D), device="cuda")
    _one_step(cfg, model, mb)


def test_mnist_unet_ctelbo_train_step():
    mu, _, _, _, _ = _registries()
    from config.mnist_config.config_tauUnet_mnist import get_config
    cfg = get_config()
    torch.manual_seed(0)
    model = mu.create_model(cfg, torch.device("cuda"))
    mb = torch.randint(0, 256, (2, 1, 28, 28), device="cuda")
    _one_step(cfg, model, mb)


@pytest.mark.parametrize("sampler", ["MidPointTauL", "TauL_corrector"])
def test_maze_hollow_full_size_samplers(sampler):
    """BASELINE config (4) at its real size: config_hollow_maze (D = 225, S = 3, 7.8 M-parameter hollow transformer) with
    MidPointTauL, and with TauL + corrector steps.  The oracle covers these loops at D = 15 (test_gpu_samplers); here the
    properties that do not depend on the size: shapes, state range, per-step counters consistent with each other, the
    corrector entered, reproducibility under a fixed Philox key, and the engine (not torch ops) behind model(x, t)."""
    mu, su, _, _, _ = _registries()
    from config.maze_config.config_hollow_maze import get_config
    cfg = get_config()
    cfg.device = "cuda"
    cfg.sampler.num_steps = 6
    if sampler == "MidPointTauL":
        cfg.sampler.name, cfg.sampler.is_ordinal = "MidPointTauL", True
    else:
        cfg.sampler.name = "TauL"
        cfg.sampler.corrector_entry_time, cfg.sampler.num_corrector_steps = 0.5, 2
    torch.manual_seed(0)
    model = mu.create_model(cfg, torch.device("cuda"))
    model.eval()
    N, D, S = 64, 225, 3
    smp = su.get_sampler(cfg)
    smp.seed = 99
    out = smp.sample(model, N)
    assert model._engine is not None and model._engine._plans            # the HIP engine ran the network
    x = out[0]
    assert x.shape == (N, D) and x.min() >= 0 and x.max() < S
    if sampler == "MidPointTauL":
        assert len(out) == 5
        change_jump, change_dim, change_first, change_1to2 = (np.asarray(o) for o in out[1:])
        nst = len(change_dim)
        assert nst >= 5 and change_jump.shape == change_first.shape == change_1to2.shape == (nst,)
        assert ((change_dim >= 0) & (change_dim <= 1)).all() and ((change_first >= 0) & (change_first <= 1)).all()
        fin = np.isfinite(change_jump)
        assert ((change_jump[fin] >= 0) & (change_jump[fin] <= 1)).all()
        assert change_dim.max() > 0                                       # the chain moves
    else:
        assert len(out) == 2 and len(out[1]) == cfg.sampler.num_steps and max(out[1]) > 0
    smp2 = su.get_sampler(cfg)
    smp2.seed = 99
    assert np.array_equal(smp2.sample(model, N)[0], x)                    # same key, same draws
    model.train()


def test_cifar10_unet_ctelbo_lambda_train_step():
    """BASELINE config (5): config_tauUnet_cifar10 (CTElboLambda, logistic head, D = 3072): one training step at the
    configuration's size (K11 sees 3072 rows of 256 states per sample; clip + Adam + EMA in K28)."""
    mu, _, _, _, _ = _registries()
    from config.cifar10_config.config_tauUnet_cifar10 import get_config
    cfg = get_config()
    assert cfg.loss.name == "CTElboLambda"
    torch.manual_seed(0)
    model = mu.create_model(cfg, torch.device("cuda"))
    g = torch.Generator(device="cuda").manual_seed(1)
    with torch.no_grad():
        for name, p in model.named_parameters():                 # the reference zero-scales some convs at init: re-draw
            if p.dim() > 1:
                p.copy_(torch.randn(p.shape, generator=g, device="cuda") / (p[0].numel() ** 0.5))
    model.init_ema()
    mb = torch.randint(0, 256, (2, 3, 32, 32), device="cuda")
    mu_, _, lu, tu, ou = _registries()
    state = {"model": model, "optimizer": ou.get_optimizer(model.parameters(), cfg), "n_iter": cfg.training.n_iters // 2}
    out = tu.get_train_step(cfg).step(state, lu.get_loss(cfg), mb)
    assert out.dim() == 0 and torch.isfinite(out) and float(out) < 1e8 and model.num_updates == 1
