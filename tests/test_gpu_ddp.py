"""GPU: cfg.distributed training with two ranks on the one GPU of the test box -- gloo carries the gradient
all-reduce here (RCCL refuses two ranks on one device; on a multi-GPU node the same code runs with backend "nccl", one
rank per GPU).  Checks that the fused HIP Adam + EMA + clip step (K28) under DistributedDataParallel leaves both ranks
with identical weights and EMA shadows after three CT-ELBO steps of the MNIST tauLDR U-Net."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _run(rank, world, port, q, kind="unet"):
    import torch.distributed as dist
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "continuous-time-diffusion-models-for-discrete-data_amd")]
    import lib.models.models, lib.losses.losses, lib.training.training, lib.optimizers.optimizers  # noqa: F401,E401
    import lib.models.model_utils as mu
    import lib.losses.losses_utils as lu
    import lib.training.training_utils as tu
    import lib.optimizers.optimizers_utils as ou
    if kind == "hollow":
        from config.maze_config.config_hollow_maze import get_config
    else:
        from config.mnist_config.config_tauUnet_mnist import get_config
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    cfg = get_config()
    cfg.device, cfg.distributed = "cuda", True
    if kind == "hollow":
        cfg.model.num_layers = 2
    torch.manual_seed(0)
    model = mu.create_model(cfg, torch.device("cuda"), rank=rank)
    state = {"model": model, "optimizer": ou.get_optimizer(model.parameters(), cfg), "n_iter": 0}
    loss, step = lu.get_loss(cfg), tu.get_train_step(cfg)
    g = torch.Generator().manual_seed(100 + rank)                  # a different minibatch per rank
    for it in range(3):
        mb = (torch.randint(0, 3, (4, 1, 15, 15), generator=g) if kind == "hollow" else torch.randint(0, 256, (4, 1, 28, 28), generator=g)).cuda()
        torch.manual_seed(1000 * it + rank)
        last = step.step(state, loss, mb)
        state["n_iter"] += 1
    if kind == "hollow":
        assert model._trainer is not None, "the HIP training path did not run under DistributedDataParallel"
    else:
        assert getattr(model._engine, "_train_plans", None), "the HIP training plan did not run under DistributedDataParallel"
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).double().cpu()
    ema = torch.cat([p.detach().reshape(-1) for p in model.shadow_params]).double().cpu()
    q.put((rank, float(last), flat[::97].numpy(), float(flat.abs().sum()), float(ema.abs().sum())))
    dist.barrier()
    dist.destroy_process_group()


def _worker(rank, world, port, q, kind="unet"):
    try:
        _run(rank, world, port, q, kind)
    except BaseException:                                            # report instead of leaving the parent waiting
        import traceback
        q.put((rank, "ERROR", traceback.format_exc()))
        raise


@pytest.mark.parametrize("kind", ["unet", "hollow"])
def test_ddp_fused_step_two_ranks_one_gpu(kind):
    """kind "hollow": the maze hollow transformer (2 blocks per direction) on the HIP training Functions, ScoreElbo."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, kind)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        res = [q.get(timeout=240) for _ in range(2)]
    finally:
        for p in procs:
            p.join(timeout=30)
            if p.is_alive():
                p.kill()
    for r in res:
        assert r[1] != "ERROR", r[2]
    res.sort()
    assert all(np.isfinite(r[1]) for r in res)
    np.testing.assert_array_equal(res[0][2], res[1][2])              # same weights on both ranks
    assert res[0][3] == res[1][3] and res[0][4] == res[1][4]         # ... and the same EMA shadows
    assert res[0][1] != res[1][1]                                    # (the ranks did see different minibatches)
