"""Train step (reference lib/training/training.py:17-40): zero_grad -> loss -> NaN guard ->
backward -> clip -> warm-up LR -> Adam step -> EMA.  Accepts both argument orders found in the
reference's scripts (`step(state, loss, minibatch)` and the stale `step(state, minibatch, loss)`,
SURVEY 0.2) by dispatching on which argument has `calc_loss`."""
import inspect

import numpy as np
import torch

import lib.training.training_utils as training_utils


def call_calc_loss(loss, state, minibatch, label=None):
    """Losses come with (state, minibatch, label) or the old (minibatch, state) signature."""
    names = list(inspect.signature(loss.calc_loss).parameters)
    if names and names[0] == "minibatch":
        return loss.calc_loss(minibatch, state)
    return loss.calc_loss(state, minibatch, label)


@training_utils.register_train_step
class Standard:
    def __init__(self, cfg):
        self.do_ema = "ema_decay" in cfg.model
        self.clip_grad = cfg.training.clip_grad
        self.grad_norm = cfg.training.grad_norm
        self.warmup = cfg.training.warmup
        self.lr = cfg.optimizer.lr
        self.device = cfg.device

    def step(self, state, loss, minibatch, label=None):
        if hasattr(minibatch, "calc_loss"):                # old order: step(state, minibatch, loss)
            loss, minibatch = minibatch, loss
        state["optimizer"].zero_grad()
        l = call_calc_loss(loss, state, minibatch, label)
        if l.isnan().any() or l.isinf().any():
            print("Loss is nan or inf")
            return torch.tensor(1e9, device=self.device)
        l.backward()
        opt, model = state["optimizer"], state["model"]
        if self.warmup > 0:                                   # (the reference sets the LR after clipping; the two commute)
            for g in opt.param_groups:
                # (a python float: the reference stores numpy's float64 here, which ends up in the optimizer's state_dict
                #  and makes its checkpoints need an allow-listed numpy global to load safely -- bookkeeping.load_state)
                g["lr"] = float(self.lr * np.minimum(state["n_iter"] / self.warmup, 1.0))
        owner = model.module if hasattr(model, "module") else model      # DDP wrapper
        first = next(model.parameters())
        if first.is_cuda and hasattr(opt, "fused_step"):
            # K28: clip + Adam + EMA of every tensor in two libctdd launches (csrc/optim.hip)
            shadow, decay, trainable = None, -1.0, None
            if self.do_ema:
                decay = owner.next_ema_decay()
                shadow, trainable = owner.shadow_params, owner._trainable()
            opt.fused_step(self.grad_norm if self.clip_grad else 0.0, shadow, decay, trainable)
            owner._weights_version = getattr(owner, "_weights_version", 0) + 1    # raw-pointer writes: tell the inference engine
        else:                                                  # host-logic path for cfg.device == "cpu" (torch ops, as the reference)
            if self.clip_grad:
                torch.nn.utils.clip_grad_norm_(model.parameters(), self.grad_norm)
            opt.step()
            if self.do_ema:
                model.update_ema()
        return l.detach()
