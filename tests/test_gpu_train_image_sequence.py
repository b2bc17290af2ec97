"""GPU: the call sequence of the reference's training driver, TAUnSDDM/train_image.py:51-153, performed through the mirror
(`lib/...` of the package) exactly in its order and with its argument shapes:

    create_model -> get_optimizer -> state dict -> [train_resume: load_config / load_state / save_config] -> get_loss ->
    get_train_step -> get_sampler -> get_dataset -> DataLoader -> step(state, loss, minibatch.long()) -> save_state ->
    model.eval() -> sampler.sample(model, n) -> reshape(n, C, H, W) -> model.train() -> n_iter += 1

on a tiny IDX file written into tmp_path (no download), with a reduced U-Net so the test runs in seconds.  This is the
"train_image.py runs unchanged" claim of INTEGRATION.md executed: every registry lookup, constructor signature, return
type and on-disk artefact the script touches."""
import glob
import os
import struct

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _write_idx(path, arr):
    arr = np.ascontiguousarray(arr, dtype=np.uint8)
    with open(path, "wb") as f:
        f.write(struct.pack(">HBB", 0, 0x08, arr.ndim))
        f.write(struct.pack(">" + "I" * arr.ndim, *arr.shape))
        f.write(arr.tobytes())


def _tiny_cfg():
    from config.mnist_config.config_tauUnet_mnist import get_config
    cfg = get_config()
    from config._common import tau_unet
    cfg.device = "cuda"
    tau_unet(cfg, 32, [1, 2, 2], 1, 28, "logits")     # a reduced U-Net (channel counts stay multiples of 16: the HIP training plan's coverage)
    cfg.model.num_res_blocks = 1
    cfg.model.attn_resolutions = [48]              # as the shipped config (int(96 / 2)): attention in the mid block only
    cfg.data.batch_size = 8
    cfg.data.use_augm = False
    cfg.sampler.num_steps = 6
    cfg.sampler.sample_freq = 3
    cfg.saving.checkpoint_freq = 3
    cfg.training.n_iters = 6
    cfg.training.warmup = 2
    return cfg


def test_train_image_call_sequence(tmp_path):
    import lib.datasets.dataset_utils as dataset_utils
    import lib.datasets.mnist  # noqa: F401
    import lib.losses.losses  # noqa: F401
    import lib.losses.losses_utils as losses_utils
    import lib.models.model_utils as model_utils
    import lib.models.models  # noqa: F401
    import lib.optimizers.optimizers  # noqa: F401
    import lib.optimizers.optimizers_utils as optimizers_utils
    import lib.sampling.sampling  # noqa: F401
    import lib.sampling.sampling_utils as sampling_utils
    import lib.training.training  # noqa: F401
    import lib.training.training_utils as training_utils
    import lib.utils.bookkeeping as bookkeeping
    from ctdd import native

    # ---- a 24-image "MNIST" where torchvision would have put it
    raw = tmp_path / "data" / "MNIST" / "raw"
    raw.mkdir(parents=True)
    rng = np.random.default_rng(0)
    _write_idx(raw / "train-images-idx3-ubyte", rng.integers(0, 256, (24, 28, 28)))
    _write_idx(raw / "train-labels-idx1-ubyte", rng.integers(0, 10, (24,)))
    save_location = str(tmp_path / "SavedModels" / "MNIST")
    dataset_location = str(tmp_path / "data")

    def run(train_resume, resume_from=None):
        # ---- train_image.py:36-65
        if not train_resume:
            cfg = _tiny_cfg()
            bookkeeping.save_config(cfg, save_location)
        else:
            cfg = bookkeeping.load_config(resume_from["config"])
        device = torch.device(cfg.device)
        torch.manual_seed(0)
        model = model_utils.create_model(cfg, device)
        optimizer = optimizers_utils.get_optimizer(model.parameters(), cfg)
        state = {"model": model, "optimizer": optimizer, "n_iter": 0}
        if train_resume:
            state = bookkeeping.load_state(state, resume_from["checkpoint"], device)
            cfg.training.n_iters = 9
            cfg.sampler.sample_freq = 3
            cfg.saving.checkpoint_freq = 3
            cfg.sampler.num_steps = 6
            bookkeeping.save_config(cfg, save_location)
        # ---- 67-76
        loss = losses_utils.get_loss(cfg)
        training_step = training_utils.get_train_step(cfg)
        sampler = sampling_utils.get_sampler(cfg)
        dataset = dataset_utils.get_dataset(cfg, device, dataset_location)
        dataloader = torch.utils.data.DataLoader(dataset, batch_size=cfg.data.batch_size, shuffle=cfg.data.shuffle)
        n_params = sum(p.numel() for p in model.parameters())
        assert n_params > 0 and cfg.data.name == "DiscreteMNIST"
        # ---- 92-153
        n_samples = 4
        training_loss, sampled, exit_flag = [], [], False
        launches0 = dict(native.launch_counts()) if hasattr(native, "launch_counts") else None
        while True:
            for minibatch, label in dataloader:
                minibatch = minibatch.to(device)
                l = training_step.step(state, loss, minibatch.long())
                training_loss.append(l.item())
                if (state["n_iter"] + 1) % cfg.saving.checkpoint_freq == 0 or state["n_iter"] == cfg.training.n_iters - 1:
                    bookkeeping.save_state(state, save_location)
                if (state["n_iter"] + 1) % cfg.sampler.sample_freq == 0 or state["n_iter"] == cfg.training.n_iters - 1:
                    state["model"].eval()
                    samples, _ = sampler.sample(state["model"], n_samples)
                    samples = samples.reshape(n_samples, cfg.model.input_channels, cfg.data.image_size, cfg.data.image_size)
                    state["model"].train()
                    sampled.append(samples)
                state["n_iter"] += 1
                if state["n_iter"] > cfg.training.n_iters - 1:
                    exit_flag = True
                    break
            if exit_flag:
                break
        return cfg, state, training_loss, sampled, launches0

    cfg, state, losses, sampled, _ = run(False)
    assert state["n_iter"] == 6 and len(losses) == 6 and all(np.isfinite(losses)) and all(l < 1e8 for l in losses)
    assert len(sampled) == 2 and all(s.shape == (4, 1, 28, 28) and s.dtype.kind == "i" and 0 <= s.min() and s.max() <= 255 for s in sampled)
    assert state["model"].training                                   # eval() -> sample -> train() left the module in train mode
    ckpts = sorted(glob.glob(os.path.join(save_location, "*", "model_*.pt")))
    confs = sorted(glob.glob(os.path.join(save_location, "*", "config_001.yaml")))
    assert [os.path.basename(c) for c in ckpts] == ["model_2.pt", "model_5.pt"] and len(confs) == 1
    ck = torch.load(ckpts[-1], map_location="cpu", weights_only=True)
    assert set(ck) == {"model", "optimizer", "n_iter"} and ck["n_iter"] == 5
    assert {"ema_decay", "ema_num_updates", "ema_shadow_params"} <= set(ck["model"])            # models.py:760-766
    # the weights did move, and the EMA shadow follows them
    w_live = {k: v.detach().clone() for k, v in state["model"].state_dict().items() if torch.is_tensor(v) and v.dtype.is_floating_point}

    # ---- train_resume = True (train_image.py:44-47, 57-65): config from YAML, state from the .pt, three more iterations
    cfg2, state2, losses2, sampled2, _ = run(True, {"config": confs[0], "checkpoint": ckpts[-1]})
    assert cfg2.model.ch == 32 and list(cfg2.model.ch_mult) == [1, 2, 2] and cfg2.sampler.num_steps == 6
    assert state2["n_iter"] == 9 and len(losses2) == 4 and all(np.isfinite(losses2))            # resumes at n_iter 5: 5, 6, 7, 8
    assert len(sampled2) >= 1 and sampled2[-1].shape == (4, 1, 28, 28)
    # resumed from iteration 5 of the first run: same weights there (load_state restored model, EMA and Adam moments)
    probe = next(k for k in w_live if k.endswith("weight"))
    assert not torch.equal(state2["model"].state_dict()[probe].cpu(), ck["model"][probe])       # ... and trained on from them
    assert os.path.exists(os.path.join(save_location, os.path.basename(os.path.dirname(ckpts[-1])), "model_8.pt"))
