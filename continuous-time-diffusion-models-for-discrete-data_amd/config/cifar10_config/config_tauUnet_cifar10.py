"""CIFAR-10 tauLDR U-Net (D=3072, S=256), logistic head, CTElboLambda loss
(reference config/cifar10_config/config_tauUnet_cifar10.py)."""
from config._common import skeleton, image_data, tau_unet


def get_config():
    c = skeleton("SavedModels/CIFAR10/")
    c.experiment_name = "cifar10"
    c.loss.update(name="CTElboLambda", eps_ratio=1e-9, nll_weight=0, min_time=0.01, one_forward_pass=True)
    c.training.update(n_iters=500000, grad_norm=1, max_t=1)
    image_data(c, "DiscreteCIFAR10", 256, 32, 3, 64)
    c.data.random_flips = True
    tau_unet(c, 128, [1, 2, 2, 2], 3, 32, "logistic_pars")
    c.model.fix_logistic = False
    c.saving.checkpoint_freq = 5000
    c.sampler.update(name="TauL", num_steps=1000, min_t=0.01, initial_dist="gaussian", sample_freq=5000)
    return c
