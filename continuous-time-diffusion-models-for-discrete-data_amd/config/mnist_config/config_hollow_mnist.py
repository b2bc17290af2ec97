"""MNIST SDDM hollow transformer (E=256, 9 layers), ratio-matching family losses
(reference config/mnist_config/config_hollow_mnist.py; BASELINE config 3 sets loss.name="CatRMNLL")."""
from config._common import skeleton, image_data, hollow


def get_config():
    c = skeleton("SavedModels/MNISTHollow")
    c.loss.update(name="ScoreElbo", logit_type="reverse_prob", loss_type="rm", ce_coeff=0, eps_ratio=1e-9,
                  min_time=0.007, one_forward_pass=True, nll_weight=0.01)
    c.training.update(n_iters=600000, grad_norm=1, max_t=0.99999, resume=True)
    image_data(c, "DiscreteMNIST", 256, 28, 1, 32)
    c.data.is_img = True
    c.model.name = "GaussianHollowEMA"
    hollow(c, 256, 9, 512, 28 * 28, 256)
    c.model.update(out_dim=256, rate_sigma=6.0, Q_sigma=512.0, time_exp=100.0, time_base=3.0, rate_const=2.1)
    c.saving.checkpoint_freq = 10000
    c.sampler.update(name="TauL", num_steps=1000, min_t=0.007, initial_dist="gaussian", sample_freq=22000000)
    return c
