"""MNIST tauLDR U-Net (D=784, S=256), CT-ELBO loss, tau-leaping 1000 steps -- the headline
benchmark config (reference config/mnist_config/config_tauUnet_mnist.py)."""
from config._common import skeleton, image_data, tau_unet


def get_config():
    c = skeleton("SavedModels/MNIST/")
    c.experiment_name = "mnist"
    c.loss.update(name="CTElbo", eps_ratio=1e-9, nll_weight=0, min_time=0.01, one_forward_pass=True)
    c.training.update(n_iters=600000, grad_norm=2, max_t=1)
    image_data(c, "DiscreteMNIST", 256, 28, 1, 64)
    c.data.random_flips = True
    tau_unet(c, 96, [1, 2, 2], 1, 28, "logits")
    c.saving.checkpoint_freq = 1000
    c.sampler.update(name="TauL", num_steps=1000, min_t=0.01, initial_dist="gaussian", sample_freq=1000)
    return c
