"""Golden vectors for SURVEY 8(f).2-4 (checkpoint IO, dataset adapters, evaluation metrics)  --  runs ONLY in the build
container (needs /root/reference).

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_aux.py

Imports the reference read-only.  Third-party modules the reference imports but never calls on these paths are stood in
for by the build-owned stubs of oracle/stubs (`absl`, `ml_collections.config_flags`: command-line helpers of
lib/datasets/synthetic.py) and by an empty `torch.utils.tensorboard` module object (lib/utils/bookkeeping.py:7 imports it
for legacy writers nothing calls).  Outputs, all data:

  tests/golden/aux_maze.npz        mazes drawn by the reference's `maze_gen` under python `random.seed`, their walls, what
                                   the reference's `find_path` makes of them, `maze_acc` on clean + corrupted samples, and
                                   the state statistics of 400 reference mazes
  tests/golden/aux_synthetic.npz   `float2bin` / `bin2float` of the reference (gray and normal bin maps) on seeded points
  tests/golden/aux_metrics.npz     `binary_exp_hamming_mmd` / `binary_exp_hamming_sim` of the reference on seeded bit arrays
  tests/golden/aux_checkpoint/     `model_3.pt` written by the reference's `save_state` after three reference training steps
                                   (warm-up on: the optimizer's lr is numpy's float64) of a tiny hollow-transformer model,
                                   plus `aux_checkpoint.npz`: the inputs and the reference model's logits (live and EMA
                                   weights) the restored mirror model must reproduce

`save_config` / `load_config` (bookkeeping.py:374-394) go through ruamel.yaml, which is absent from this image: they are
NOT run here and stay parity unpinned (the mirror's YAML round trip is tested on its own files only).
"""
import os
import random
import sys
import types
import warnings

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.dont_write_bytecode = True
sys.path[:0] = [os.path.join(HERE, "stubs"), "/root/reference/TAUnSDDM", ROOT]
warnings.filterwarnings("ignore")

import numpy as np  # noqa: E402
import torch  # noqa: E402

_tb = types.ModuleType("torch.utils.tensorboard")          # imported by bookkeeping.py:7, used only by writers nothing calls
_tb.SummaryWriter = object
sys.modules["torch.utils.tensorboard"] = _tb
torch.utils.tensorboard = _tb

from ml_collections import ConfigDict  # noqa: E402  (stub)

OUT = os.path.join(ROOT, "tests", "golden")
torch.set_num_threads(1)


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrs.items()})
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def gen_maze():
    import lib.datasets.maze as rm
    out = {}
    for tag, rt, crop in (("fixed", False, False), ("random", True, False)):
        random.seed(11 if rt else 7)
        np.random.seed(3)
        mz = rm.maze_gen(limit=12, crop=crop, random_transform=rt, dim_x=7, dim_y=7, pixelSizeOfTile=1, weightHigh=99,
                         weightLow=97, device="cpu")
        mz = mz.numpy().reshape(-1, 15, 15).astype(np.int64)
        out[f"{tag}__solved"] = mz
        walls = mz.copy()
        walls[walls == 1] = 2
        out[f"{tag}__walls"] = walls
        # what the reference's own solver makes of the bare walls (rt = entries searched on the border)
        re = np.stack([rm.find_path(w.copy(), rt) for w in walls], 0)
        out[f"{tag}__resolved"] = re
    # maze_acc (maze.py:866-898) on clean, corrupted and unsolvable samples
    good = out["random__solved"]
    rng = np.random.default_rng(5)
    bad = good.copy()
    for i in range(bad.shape[0]):
        k = i % 4
        if k == 0:                                          # a path cell turned into floor: not the solved form
            ys, xs = np.nonzero(bad[i] == 1)
            j = rng.integers(len(ys))
            bad[i, ys[j], xs[j]] = 2
        elif k == 1:                                        # a wall punched through: a shorter path may exist
            ys, xs = np.nonzero(bad[i][1:-1, 1:-1] == 0)
            j = rng.integers(len(ys))
            bad[i, ys[j] + 1, xs[j] + 1] = 2
        elif k == 2:                                        # an opening closed: no two entries
            border = np.zeros((15, 15), dtype=bool)
            border[0, :] = border[-1, :] = border[:, 0] = border[:, -1] = True
            ys, xs = np.nonzero((bad[i] != 0) & border)
            bad[i, ys[0], xs[0]] = 0
        # k == 3: left intact
    samples = np.concatenate([good, bad], 0)
    kept = rm.maze_acc(samples.copy())
    out["acc__samples"], out["acc__kept"] = samples, kept
    # state statistics of the generator (distribution-level pin for the mirror's own generator)
    random.seed(123)
    big = rm.maze_gen(limit=400, crop=False, random_transform=True, dim_x=7, dim_y=7, device="cpu").numpy().reshape(-1, 15, 15)
    out["stats__counts"] = np.stack([(big == s).sum(axis=(1, 2)) for s in (0, 1, 2)], 1)
    out["stats__rot_share"] = np.array([(big[:, 0, :] == 0).all(axis=1).mean()])     # rotated mazes have their openings on the sides
    save("aux_maze", **out)


def gen_synthetic():
    import lib.datasets.synthetic as rs
    out = {}
    rng = np.random.default_rng(21)
    for D, scale in ((32, 5461.76), (16, 21.3)):
        lim = (1 << (D // 2 - 1)) - 1
        pts = (rng.uniform(-1, 1, (64, 2)) * lim / scale * 0.999).astype(np.float64)
        pts[0] = [0.0, -0.0]
        pts[1] = [lim / scale, -lim / scale]
        for mode in ("gray", "normal"):
            bm, inv_bm = rs.get_binmap(D, mode)
            bits = rs.float2bin(pts, bm, D, scale)
            back = rs.bin2float(bits.astype(np.int32), inv_bm, D, scale)
            out[f"D{D}__{mode}__points"], out[f"D{D}__{mode}__bits"], out[f"D{D}__{mode}__back"] = pts, bits, back
            out[f"D{D}__{mode}__scale"] = np.array([scale])
    save("aux_synthetic", **out)


def gen_metrics():
    import lib.datasets.metrics as rmx
    out = {}
    g = torch.Generator().manual_seed(4)
    for tag, (n, m, D, p, q) in {"a": (40, 56, 32, 0.5, 0.5), "b": (64, 64, 32, 0.3, 0.6), "c": (17, 5, 8, 0.5, 0.2)}.items():
        x = (torch.rand(n, D, generator=g) < p).to(torch.int64)
        y = (torch.rand(m, D, generator=g) < q).to(torch.int64)
        out[f"{tag}__x"], out[f"{tag}__y"] = x.numpy(), y.numpy()
        for bw in (0.1, 0.5):
            out[f"{tag}__mmd_bw{bw}"] = np.array([float(rmx.binary_exp_hamming_mmd(x, y, None, bandwidth=bw))])
        out[f"{tag}__sim_bw0.1"] = rmx.binary_exp_hamming_sim(x.float(), y.float(), 0.1).numpy()
    save("aux_metrics", **out)


def gen_checkpoint():
    import lib.models.models  # noqa: F401  (registers the models)
    import lib.models.model_utils as mu
    import lib.losses.losses  # noqa: F401
    import lib.losses.losses_utils as lu
    import lib.training.training  # noqa: F401
    import lib.training.training_utils as tu
    import lib.optimizers.optimizers  # noqa: F401
    import lib.optimizers.optimizers_utils as ou
    import lib.utils.bookkeeping as rb
    from config.synthetic_config.config_hollow_synthetic import get_config
    cfg = get_config()
    cfg.device = "cpu"
    cfg.distributed = False
    S, D = 3, 12
    cfg.data.S = S
    cfg.data.batch_size = 8
    cfg.model.name = "UniVarHollowEMA"
    cfg.model.concat_dim, cfg.model.embed_dim, cfg.model.num_layers, cfg.model.num_heads = D, 16, 1, 2
    cfg.model.mlp_dim, cfg.model.qkv_dim, cfg.model.readout_dim = 32, 16, S
    cfg.model.dropout_rate = 0.0
    cfg.model.attention_dropout_rate = 0.0
    cfg.model.ema_decay = 0.9
    cfg.loss.name = "CatRM"
    cfg.loss.logit_type = "reverse_prob"
    cfg.training.warmup = 10                                  # -> numpy float64 learning rate in the optimizer state (training.py:31-33)
    cfg.optimizer.lr = 1e-2
    torch.manual_seed(0)
    model = mu.create_model(cfg, torch.device("cpu"))
    state = {"model": model, "optimizer": ou.get_optimizer(model.parameters(), cfg), "n_iter": 0}
    loss, step = lu.get_loss(cfg), tu.get_train_step(cfg)
    g = torch.Generator().manual_seed(1)
    for _ in range(3):
        mb = torch.randint(0, S, (8, D), generator=g)
        step.step(state, loss, mb)
        state["n_iter"] += 1
    ck_dir = os.path.join(OUT, "aux_checkpoint")
    os.makedirs(ck_dir, exist_ok=True)
    tmp = os.path.join(ck_dir, "_tmp")
    rb.save_state(state, tmp)                                 # <tmp>/<date>/model_3.pt, exactly as the reference writes it
    (date,) = os.listdir(tmp)
    os.replace(os.path.join(tmp, date, "model_3.pt"), os.path.join(ck_dir, "model_3.pt"))
    os.rmdir(os.path.join(tmp, date))
    os.rmdir(tmp)
    x = torch.randint(0, S, (4, D), generator=g)
    t = torch.tensor([0.1, 0.4, 0.7, 0.95])
    with torch.no_grad():                                     # (the module is in train mode; dropout rates are 0)
        live = model(x, t)
    model.eval()                                              # EMA swap (models.py:806-823)
    with torch.no_grad():
        ema = model(x, t)
    model.train()
    fields = {k: cfg.model[k] for k in ("concat_dim", "embed_dim", "num_layers", "num_heads", "mlp_dim", "qkv_dim", "readout_dim",
                                        "ema_decay", "rate_const", "t_func")}
    save(os.path.join("aux_checkpoint", "aux_checkpoint"), x=x.numpy(), t=t.numpy(), logits_live=live.numpy(), logits_ema=ema.numpy(),
         n_iter=np.array([state["n_iter"]]), lr=np.array([float(state["optimizer"].param_groups[0]["lr"])]),
         ema_num_updates=np.array([int(model.state_dict()["ema_num_updates"])]), cfg=np.array(repr({**fields, "S": S})))
    print("checkpoint", os.path.getsize(os.path.join(ck_dir, "model_3.pt")) // 1024, "KiB")


if __name__ == "__main__":
    which = sys.argv[1:] or ["maze", "synthetic", "metrics", "checkpoint"]
    for w in which:
        {"maze": gen_maze, "synthetic": gen_synthetic, "metrics": gen_metrics, "checkpoint": gen_checkpoint}[w]()
