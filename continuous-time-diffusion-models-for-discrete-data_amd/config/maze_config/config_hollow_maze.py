"""Maze 15x15 (D=225, S=3: wall / path / free) hollow transformer
(reference config/maze_config/config_hollow_maze.py).  The shipped sampler name "CRMLBJF" is
not a registered class in the reference; it resolves to LBJF here."""
from config._common import skeleton, hollow


def get_config():
    c = skeleton("SavedModels/MAZEelbo/")
    c.loss.update(name="ScoreElbo", logit_type="reverse_prob", loss_type="rm", ce_coeff=0, eps_ratio=1e-9,
                  min_time=0.007, one_forward_pass=True, nll_weight=0.01)
    c.training.update(n_iters=300000, grad_norm=3, max_t=0.99999, resume=True)
    c.data.update(name="Maze3S", S=3, is_img=True, batch_size=128, shuffle=True, image_size=15,
                  shape=[1, 15, 15], use_augm=False, crop_wall=False, limit=1, random_transform=True)
    c.model.update(name="UniVarHollowEMA", rate_const=1.7, Q_sigma=512.0, t_func="sqrt_cos")
    hollow(c, 128, 8, 1024, 15 * 15, 3)
    c.saving.checkpoint_freq = 10000
    c.sampler.update(name="CRMLBJF", num_steps=750, min_t=0.007, initial_dist="uniform", sample_freq=5000000000)
    return c
