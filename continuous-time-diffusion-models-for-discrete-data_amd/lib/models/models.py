"""Model objects = EMA (+) score-network wrapper (+) forward CTMC process, registered under the
reference's class names (lib/models/models.py:192-299, 495-525, 730-823, 832-1082).

A model is callable `model(x, t) -> (B, D, S)` fp32 logits and exposes `.transition/.rate/
.rate_mat/.transit_between`, `.device`, `.S`, `.update_ema()`, EMA-swapping `.train()/.eval()` and a
`state_dict()` carrying `ema_decay / ema_num_updates / ema_shadow_params`, as the reference does.
Only the wrappers named by the BASELINE configs are built."""
import warnings

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.parallel import DistributedDataParallel as DDP

import lib.models.model_utils as model_utils
from lib.models.forward_model import GaussianTargetRate, UniformRate, UniformVariantRate, BirthDeathForwardBase  # noqa: F401
from lib.networks import unet


def unwrap(net):
    """The module behind a DistributedDataParallel wrapper (cfg.distributed, models.py:104-107)."""
    return net.module if isinstance(net, DDP) else net


def _warn_once(model, key, msg):
    """A fallback to torch device ops is never silent: one warning per model and reason."""
    seen = model.__dict__.setdefault("_fallback_warned", set())
    if key not in seen:
        seen.add(key)
        warnings.warn(f"[ctdd] {model.__class__.__name__}: {msg}", RuntimeWarning, stacklevel=3)


class borrow_engine_output:
    """`with borrow_engine_output(model):` -- inside, model(x, t) returns the engine plan's own logits buffer
    (no copy); the next forward overwrites it.  The sampler loops consume logits within a step and use this."""

    def __init__(self, model, bf16_logits=False, uniform_time=False):
        """bf16_logits: the U-Net engine's output convolution writes (B, D, S) logits in bf16 (bf16 engine, `logits` head only;
        any other model ignores the request and returns fp32) -- what the S = 256 bf16 step kernel reads.
        uniform_time: a PROMISE of the caller that every forward inside passes one time value for the whole batch
        (`t * ones((N,))`, as every sampler does): the U-Net engine then runs its time path once for times[0]."""
        self.model = model
        self.bf16_logits = bool(bf16_logits)
        self.uniform_time = bool(uniform_time)

    def __enter__(self):
        self.prev = (getattr(self.model, "_borrow_engine_output", False), getattr(self.model, "_engine_logits_bf16", False),
                     getattr(self.model, "_engine_uniform_time", False))
        self.model._borrow_engine_output = True
        self.model._engine_logits_bf16 = self.bf16_logits
        self.model._engine_uniform_time = self.uniform_time
        return self.model

    def __exit__(self, *exc):
        self.model._borrow_engine_output, self.model._engine_logits_bf16, self.model._engine_uniform_time = self.prev
        return False


def _maybe_ddp(net, cfg, rank):
    """cfg.distributed: minibatch-sharded data parallelism, one gradient all-reduce per step (RCCL on GPUs)."""
    if not cfg.distributed:
        return net
    dev = next(net.parameters()).device
    on_gpu = dev.type == "cuda"
    # find_unused_parameters: the hollow network carries two sub-modules its forward never uses (`embedding`, `temb_net`,
    # hollow_networks.py:690-712); without the flag their bucket never reduces and the second step raises
    # (the reference passes device_ids=[rank]; the device the parameters live on is the same thing on one node with one
    # process per GPU, and stays right when ranks and device indices differ -- several nodes, or ranks sharing a GPU)
    hollow = net.__class__.__name__ == "BidirectionalTransformer2"
    unused = hollow or bool(getattr(cfg, "ddp_find_unused_parameters", False))
    # Bucket size.  The hollow transformer's backward is one autograd Function per block: 25 MB buckets reduce while the
    # earlier blocks still run (overlap).  The U-Net's hand-written backward is ONE Function whose weight gradients come out
    # of a single table launch at its end, so every hook fires at once and nothing is left to overlap with:
    # there the whole gradient goes as ONE bucket -- one large all-reduce instead of three back-to-back ones (xGMI links are
    # point-to-point; fewer, larger collectives amortise their start-up).  cfg.ddp_bucket_cap_mb overrides.
    cap = getattr(cfg, "ddp_bucket_cap_mb", None)
    if cap is None:
        nbytes = sum(p.numel() * p.element_size() for p in net.parameters() if p.requires_grad)
        cap = 25 if hollow else max(25, int(nbytes / 2 ** 20) + 8)
    return DDP(net, device_ids=[dev.index if dev.index is not None else torch.cuda.current_device()] if on_gpu else None,
               find_unused_parameters=unused, bucket_cap_mb=int(cap))


def logistic_logits(mu, log_scale, S, fix_logistic, eps=1e-6):
    """Truncated-logistic head (D3PM, arXiv:2107.03006 app. A; models.py:249-283): the S bin
    log-probabilities of a logistic(mu, exp(log_scale-2)) over equal bins of [-1, 1].
    mu, log_scale: (..., 1) broadcast against S bins -> (..., S)."""
    inv_scale = torch.exp(-(log_scale - 2))
    width = 2.0 / S
    centres = torch.linspace(-1.0 + width / 2, 1.0 - width / 2, S, device=mu.device)
    left = (centres - width / 2 - mu) * inv_scale
    right = (centres + width / 2 - mu) * inv_scale
    lcdf_l, lcdf_r = F.logsigmoid(left), F.logsigmoid(right)
    lme = lambda a, b: a + torch.log1p(-torch.exp(b - a) + eps)       # log(exp(a) - exp(b)), b < a
    logits = lme(lcdf_r, lcdf_l)
    if fix_logistic:
        logits = torch.min(logits, lme(-left + lcdf_l, -right + lcdf_r))
    return logits


class ImageX0PredBasePaul(nn.Module):
    """x0-prediction wrapper around the U-Net (models.py:192-299)."""

    def __init__(self, cfg, device, rank=None):
        super().__init__()
        self.cfg = cfg
        m = cfg.model
        self.fix_logistic = m.fix_logistic
        self.data_shape = cfg.data.shape
        self.S = cfg.data.S
        self.padding = m.padding
        net = unet.UNet(in_channel=m.input_channels, out_channel=m.input_channels, channel=m.ch,
                        channel_multiplier=m.ch_mult, n_res_blocks=m.num_res_blocks,
                        attn_resolutions=m.attn_resolutions, num_heads=m.num_heads, dropout=m.dropout,
                        model_output=m.model_output, num_classes=cfg.data.S, x_min_max=m.data_min_max,
                        img_size=cfg.data.image_size + (1 if self.padding else 0)).to(device)
        self.net = _maybe_ddp(net, cfg, rank)
        self._engine = None

    def forward(self, x, times):
        if x.dim() == 2:
            B, D = x.shape
            C, H, W = self.data_shape
            x = x.view(B, C, H, W)
        else:
            B, C, H, W = x.shape
            D = C * H * W
        if self._use_engine(x):
            return self._engine_forward(x, times)
        x = x.float()
        if self.padding:
            x = F.pad(x, (0, 1, 0, 1), mode="replicate")
        out = self.net(x, times)
        if self.cfg.model.model_output == "logits":
            logits = out                                              # (B,C,H,W,S)
        else:
            mu, log_scale = out
            logits = logistic_logits(mu.unsqueeze(-1), log_scale.unsqueeze(-1), self.S, self.fix_logistic)
        if self.padding:
            logits = logits[:, :, :-1, :-1, :]
        return logits.reshape(B, D, self.S)

    # -- hand-written HIP engine (ctdd/unet_engine.py): inference plan under no_grad/eval, training plan
    #    (forward that saves what backward needs + hand-written backward) when gradients are on
    def _use_engine(self, x):
        if getattr(self.cfg.model, "engine", "hip") != "hip":
            return False                                   # explicit opt-out (cfg.model.engine = "torch")
        if not x.is_cuda:
            return False                                   # cfg.device == "cpu": the reference's host path
        from ctdd import unet_engine
        if not unet_engine.supports(self):
            _warn_once(self, "unet", "U-Net variant outside the HIP engine's coverage (padding / model_output); running torch device ops")
            return False
        if x.dtype not in (torch.int64, torch.int32):
            _warn_once(self, "unet-dtype", f"HIP engine takes integer states, got {x.dtype}; running torch device ops")
            return False
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.net.parameters()):
            ok = unet_engine.training_supported(self)
            if not ok:
                _warn_once(self, "unet-train", "U-Net outside the HIP training plan's coverage (channel counts must be multiples of 16); "
                                               "training runs on torch device ops")
            return ok
        if self.training and float(getattr(self.cfg.model, "dropout", 0.0)) > 0.0:
            # train mode without gradients (torch.no_grad() around a train-mode forward): the reference applies dropout whenever
            # the module is in train mode (unet.py:100-140); the inference plan has none, so this case runs the module itself
            _warn_once(self, "unet-train-nograd", "train-mode forward without gradients: dropout is active, running torch device ops "
                                                  "(call model.eval() for the HIP inference plan)")
            return False
        return True

    _engine_int32_states = True          # forward() takes the samplers' int32 states without a widening copy (the engine's plans do)

    def engine_time_table(self, times):
        """Sampler hook: (T,) grid times -> (T, Ntot) time-projection rows of the HIP inference plan (UNetEngine.time_table), or None
        when forwards on this device do not go through that plan.  While `_engine_time_row` holds row i, a forward at time
        times[i] skips the time path (set and cleared by the sampler loop around its network calls)."""
        probe = torch.zeros((1, 1), dtype=torch.int64, device=times.device)
        if torch.is_grad_enabled() or not self._use_engine(probe):
            return None
        from ctdd import unet_engine
        if self._engine is None:
            self._engine = unet_engine.UNetEngine(self)
        return self._engine.time_table(times)

    def _engine_forward(self, x, times):
        from ctdd import unet_engine
        if self._engine is None:
            self._engine = unet_engine.UNetEngine(self)
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.net.parameters()):
            if isinstance(self.net, DDP):
                # through DDP.forward (reducer bookkeeping for this iteration); the wrapped U-Net hands over to the training plan,
                # whose autograd Function returns every parameter gradient -> DDP's hooks bucket and all-reduce them (RCCL)
                inner = unwrap(self.net)
                inner._engine_hook = self._engine.train_forward
                try:
                    return self.net(x, times)
                finally:
                    inner._engine_hook = None
            return self._engine.train_forward(x, times)
        out = self._engine(x, times, logits_bf16=bool(getattr(self, "_engine_logits_bf16", False)),
                           uniform_time=bool(getattr(self, "_engine_uniform_time", False)), slot=getattr(self, "_engine_slot", None),
                           time_row=getattr(self, "_engine_time_row", None))
        # the plan owns its output buffer: hand out a copy unless the caller (a sampler loop that consumes the
        # logits before the next forward) asked to borrow it -- two live results must not alias
        return out if getattr(self, "_borrow_engine_output", False) else out.clone()


class HollowTransformer(nn.Module):
    """SDDM hollow transformer wrapper (models.py:495-525)."""

    def __init__(self, cfg, device, rank=None):
        super().__init__()
        from lib.networks import hollow_networks
        if cfg.model.nets == "bidir_transformer2":
            net = hollow_networks.BidirectionalTransformer2(cfg, readout_dim=None).to(device)
        else:
            raise ValueError(f"only nets='bidir_transformer2' is built (got {cfg.model.nets})")
        self.net = _maybe_ddp(net, cfg, rank)
        self.cfg = cfg
        self._engine, self._trainer = None, None

    def forward(self, x, times):
        mode = self._use_engine(x)
        if mode == "infer":
            from ctdd import hollow_engine
            if self._engine is None:
                # cfg.model.engine_precision: "bf16x3" (default; hi + lo bf16 operand pairs, three matrix-core products per
                # contraction: ~1e-5 of the logit range from the fp32 module, 2.3x its speed), "fp32" (exact-fp32 matrix
                # instructions, ~2e-6), "bf16" (single bf16 operands: ~5e-3, 3.4x)
                self._engine = hollow_engine.HollowEngine(self, precision=getattr(self.cfg.model, "engine_precision", None))
            out = self._engine(x, times)
            return out if getattr(self, "_borrow_engine_output", False) else out.clone()
        if mode == "train":
            from ctdd import hollow_train
            if self._trainer is None:
                # cfg.model.engine_train_precision: "bf16" (default: bf16 GEMM operands, fp32 accumulation / attention /
                # LayerNorm / residual streams) or "fp32" (exact-fp32 matrix instructions: gradients to ~1e-5 of autograd's)
                self._trainer = hollow_train.HollowTrainer(self)
            inner = unwrap(self.net)
            inner._engine_hook = self._trainer          # through DistributedDataParallel.forward when wrapped
            try:
                return self.net(x, times)
            finally:
                inner._engine_hook = None
        return self.net(x, times)

    # -- hand-written HIP engines: inference plan (ctdd/hollow_engine.py), training Functions (ctdd/hollow_train.py)
    def _use_engine(self, x):
        if getattr(self.cfg.model, "engine", "hip") != "hip" or not x.is_cuda:
            return None
        from ctdd import hollow_engine
        if x.dtype not in (torch.int64, torch.int32):
            _warn_once(self, "hollow-dtype", f"HIP engine takes integer states, got {x.dtype}; running torch device ops")
            return None
        if not hollow_engine.supports(self):
            _warn_once(self, "hollow", "hollow-transformer variant outside the HIP engine's coverage; running torch device ops")
            return None
        if torch.is_grad_enabled() or self.training:
            from ctdd import hollow_train
            if not hollow_train.training_supported(self):
                if getattr(self.cfg.model, "engine_train", "hip") == "hip":
                    _warn_once(self, "hollow-train", "hollow-transformer shape outside the HIP training kernels' coverage; running torch device ops")
                return None
            return "train"
        return "infer"


class EMA:
    """Exponential moving average of the trainable parameters with train/eval weight swapping
    (models.py:730-823).  Mixed in FIRST so its state_dict/load_state_dict/train win the MRO."""

    def __init__(self, cfg):
        self.decay = cfg.model.ema_decay
        self.device = cfg.device
        if self.decay < 0.0 or self.decay > 1.0:
            raise ValueError("Decay must be between 0 and 1")
        self.shadow_params, self.collected_params, self.num_updates = [], [], 0

    def _trainable(self):
        return [p for p in self.parameters() if p.requires_grad]

    def init_ema(self):
        self.shadow_params = [p.clone().detach() for p in self._trainable()]

    def next_ema_decay(self):
        """Count one update and return its decay (models.py:745-750)."""
        if len(self.shadow_params) == 0:
            raise ValueError("Shadow params not initialized before first ema update!")
        self.num_updates += 1
        return min(self.decay, (1 + self.num_updates) / (10 + self.num_updates))

    def update_ema(self):
        decay = self.next_ema_decay()
        with torch.no_grad():
            params = self._trainable()
            # shadow -= (1-decay) * (shadow - param), one fused multi-tensor launch
            torch._foreach_lerp_(self.shadow_params, [p.detach() for p in params], 1.0 - decay)

    def state_dict(self):
        sd = nn.Module.state_dict(self)
        sd["ema_decay"] = self.decay
        sd["ema_num_updates"] = self.num_updates
        sd["ema_shadow_params"] = self.shadow_params
        return sd

    def load_state_dict(self, state_dict):
        missing, unexpected = nn.Module.load_state_dict(self, state_dict, strict=False)
        if len(missing) > 0:
            raise ValueError(f"Missing keys: {missing}")
        if sorted(unexpected) != ["ema_decay", "ema_num_updates", "ema_shadow_params"]:
            raise ValueError(f"Unexpected keys: {unexpected}")
        self.decay = state_dict["ema_decay"]
        self.num_updates = state_dict["ema_num_updates"]
        self.shadow_params = state_dict["ema_shadow_params"]

    def move_shadow_params_to_model_params(self):
        for s, p in zip(self.shadow_params, self._trainable()):
            p.data.copy_(s.data)

    def move_model_params_to_collected_params(self):
        self.collected_params = [p.clone() for p in self.parameters()]

    def move_collected_params_to_model_params(self):
        for c, p in zip(self.collected_params, self.parameters()):
            p.data.copy_(c.data)

    def train(self, mode=True):
        if self.training == mode:
            raise ValueError("Dont call model.train() with the same mode twice! Otherwise EMA parameters may "
                             f"overwrite original parameters (current {self.training}, requested {mode})")
        nn.Module.train(self, mode)
        if mode:
            if len(self.collected_params) > 0:
                self.move_collected_params_to_model_params()
        else:
            self.move_model_params_to_collected_params()
            self.move_shadow_params_to_model_params()
        self._weights_version = getattr(self, "_weights_version", 0) + 1
        return self


def _compose(name, net_cls, rate_cls, doc):
    """Build `class name(EMA, net_cls, rate_cls)` with the reference's init order and register it."""

    def __init__(self, cfg, device, rank=None):
        EMA.__init__(self, cfg)
        net_cls.__init__(self, cfg, device, rank)
        rate_cls.__init__(self, cfg, device)
        self.init_ema()

    cls = type(name, (EMA, net_cls, rate_cls), {"__init__": __init__, "__doc__": doc})
    return model_utils.register_model(cls)


# U-Net models: MNIST / CIFAR-10 tauLDR (models.py:942-951), uniform variants (975-992, 851-859)
GaussianTargetRateImageX0PredEMAPaul = _compose(
    "GaussianTargetRateImageX0PredEMAPaul", ImageX0PredBasePaul, GaussianTargetRate, "tauLDR U-Net, Gaussian-target CTMC")
UniformRateUnetEMA = _compose("UniformRateUnetEMA", ImageX0PredBasePaul, UniformRate, "U-Net, uniform CTMC")
UniVarUnetEMA = _compose("UniVarUnetEMA", ImageX0PredBasePaul, UniformVariantRate, "U-Net, time-warped uniform CTMC")
UniformRateImageX0PredEMA = _compose("UniformRateImageX0PredEMA", ImageX0PredBasePaul, UniformRate, "U-Net, uniform CTMC")
# hollow-transformer models: SDDM (models.py:862-869, 954-961, 905-912)
GaussianHollowEMA = _compose("GaussianHollowEMA", HollowTransformer, GaussianTargetRate, "hollow transformer, Gaussian-target CTMC")
UniVarHollowEMA = _compose("UniVarHollowEMA", HollowTransformer, UniformVariantRate, "hollow transformer, time-warped uniform CTMC")
UniformHollowEMA = _compose("UniformHollowEMA", HollowTransformer, UniformRate, "hollow transformer, uniform CTMC")
