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
