"""Multi-GPU driver for sampling: samples are independent, so N is split across the
ranks of one node, each rank runs the whole sampler on its own GPU with its own Philox key
(`sampler.rank_stream = rank`), and there is NO collective inside the loop -- only an optional
final all_gather of the (n_r, D) integer results.  One process per GPU, torch.distributed over
RCCL ("nccl" backend) on the GPUs, gloo in the CPU tests.

Training data-parallelism keeps the reference's switch: `cfg.distributed = True` wraps the score
network in DistributedDataParallel (lib/models/models.py:_maybe_ddp): the minibatch is sharded over the ranks and the
gradients are all-reduced once per step; `clip_grad_norm_` then sees identical gradients on every rank.
  - hollow transformer: one autograd Function per block, 25 MB buckets -> the reduce of the late blocks' gradients overlaps
    the backward of the early ones;
  - U-Net: the hand-written backward is one Function whose weight gradients come from one table launch at its end, so the
    all-reduce is NOT overlapped with backward: it runs as one bucket right after it.  That is deliberate: splitting the
    weight-gradient table in two stages to expose half of the gradients early costs ~0.4 ms of the 8.6 ms MNIST step
    (atomic-bound M-split kernels), as much as the whole 56 MB all-reduce is expected to take over xGMI
    (0.1 ms direct reduce-scatter + all-gather, 0.6 ms on a ring) -- and no multi-GPU node has been available to
    this build to measure either.  `bench.py --gpus N --train` times the data-parallel step for that day.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_counts(n_total, world_size):
    """Per-rank sample counts: as even as possible, earlier ranks take the remainder."""
    base, rem = divmod(int(n_total), int(world_size))
    return [base + (1 if r < rem else 0) for r in range(world_size)]


def shard_offsets(n_total, world_size):
    c = shard_counts(n_total, world_size)
    return [int(x) for x in np.concatenate(([0], np.cumsum(c)[:-1]))]


def sample_sharded(sampler, model, n_total, gather=True, group=None):
    """Run `sampler.sample(model, n_r)` on every rank; returns the concatenated samples on every
    rank when `gather`, else this rank's shard.  Extra outputs of the sampler (change-rate lists,
    ...) are returned from the local shard only."""
    if not (dist.is_available() and dist.is_initialized()):
        return sampler.sample(model, n_total)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    counts = shard_counts(n_total, world)
    sampler.rank_stream = rank
    out = sampler.sample(model, counts[rank]) if counts[rank] > 0 else None
    local = out[0] if isinstance(out, tuple) else out
    extras = out[1:] if isinstance(out, tuple) else ()
    if not gather:
        return out
    D = int(sampler.D) if hasattr(sampler, "D") else (local.shape[1] if local is not None else 0)
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    width = max(counts)
    buf = torch.zeros((width, D), dtype=torch.int32, device=dev)
    if local is not None:
        buf[: counts[rank]] = torch.as_tensor(np.asarray(local), dtype=torch.int32).to(dev)
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf, group=group)
    full = torch.cat([p[:c] for p, c in zip(parts, counts)], dim=0).cpu().numpy().astype(int)
    return (full,) + tuple(extras) if extras else full
