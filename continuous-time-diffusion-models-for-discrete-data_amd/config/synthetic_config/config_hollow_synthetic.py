"""Synthetic 2-D toy as two 16-bit Gray-coded integers (D=32, S=2), small hollow transformer
(reference config/synthetic_config/config_hollow_synthetic.py).  "CRMLBJF" resolves to LBJF
(the reference's train_synthetic.py:74 overrides the name by hand)."""
from config._common import skeleton, hollow


def get_config():
    c = skeleton("SavedModels/Synthetic/")
    c.loss.update(name="ScoreElbo", logit_type="reverse_prob", loss_type="rm", ce_coeff=0, eps_ratio=1e-9,
                  min_time=0.007, one_forward_pass=True, nll_weight=0.01)
    c.training.update(n_iters=200000, grad_norm=1, max_t=0.99999)
    c.data.update(name="SyntheticData", type="2spirals", is_img=False, S=2, batch_size=128, shuffle=True,
                  binmode="gray", int_scale=6003.0107336488345, plot_size=4.458594271092115, shape=[32],
                  location="lib/datasets/Synthetic/data_2spirals.npy")
    c.model.update(name="UniVarHollowEMA", rate_const=2.0, Q_sigma=512.0, t_func="sqrt_cos")
    hollow(c, 64, 2, 256, 32, 2)
    c.model.out_dim = 2
    c.optimizer.lr = 1.5e-4
    c.saving.checkpoint_freq = 10000
    c.sampler.update(name="CRMLBJF", num_steps=500, min_t=0.007, initial_dist="uniform", sample_freq=200000000)
    return c
