"""Samplers behind the reference's registry names (lib/sampling/sampling.py): TauL 82-234,
LBJF 238-356, MidPointTauL 360-526, PCTauL 530-646.

Same constructor `(cfg)`, same `sample(model, N)` return shapes; the per-step work is one fused
libctdd launch (reverse rates -> jump draw -> state update) on int32 device state.  Differences
that are deliberate and documented in DESIGN.md:
  * one q_{t|0} table per step instead of N identical copies; all steps' tables are
    built in one launch and stay resident in HBM;
  * no host sync inside the loop: the per-step `changed` counters are read back once at the end;
  * Poisson draws come from Philox (seeded from torch's global generator) -- distributional,
    not stream, parity with torch.poisson.
"""
import numpy as np
import torch

import lib.sampling.sampling_utils as sampling_utils
from ctdd import native
from lib.models.model_utils import get_logprob_with_logits  # noqa: F401  (re-exported like the reference)
from lib.models.models import borrow_engine_output

_CTELBO_LOSSES = ("CTElbo", "NLL", "CTElboLambda")


def get_initial_samples(N, D, device, S, initial_dist, initial_dist_std=None, seed=None):
    """sampling.py:14-28.  Returns int64 (N,D) on `device` (uniform randint, or the discretised
    Gaussian centred at S//2 drawn by inverse CDF on the device)."""
    if seed is None:
        seed = _fresh_seed()
    if initial_dist == "uniform":
        cdf = None
    elif initial_dist == "gaussian":
        k = np.arange(1, S + 1)
        pmf = np.exp(-((k - S // 2) ** 2) / (2 * initial_dist_std**2))
        cdf = torch.from_numpy(np.cumsum(pmf / np.sum(pmf))).float().to(device)
    else:
        raise NotImplementedError("Unrecognized initial dist " + initial_dist)
    return native.initial_samples(N, D, S, torch.device(device), seed, 0xFFFF0000, cdf).long()


def _fresh_seed():
    """One 62-bit Philox key per call, taken from torch's global CPU generator so that
    torch.manual_seed() makes sampling reproducible."""
    return int(torch.randint(0, 2**62, (1,), dtype=torch.int64).item())


def _branch(cfg):
    """Which arm of get_reverse_rates applies (sampling.py:32 / 61: the elif is always truthy)."""
    return native.BRANCH_CTELBO if cfg.loss.name in _CTELBO_LOSSES else native.BRANCH_CRM


def get_reverse_rates(model, logits, x, t_ones, cfg, N, D, S):
    """sampling.py:31-78: (reverse_rates, ratio), both (N,D,S); own state not zeroed."""
    branch = _branch(cfg)
    lt = getattr(cfg.loss, "logit_type", "direct") if branch == native.BRANCH_CRM else "direct"
    need_q = branch == native.BRANCH_CTELBO or lt != "direct"
    pr = model.process
    qt0, _, rate, _ = pr.tables(t_ones, want_qt0=need_q, want_rate=True)
    tidx = torch.arange(N, dtype=torch.int32, device=logits.device)
    return native.reverse_rates(branch, lt, logits.float().contiguous(), x.to(torch.int32).contiguous(), qt0, rate,
                                cfg.sampler.eps_ratio, tidx)


class _GridSampler:
    """Shared plumbing: config, device state, resident per-step tables, final denoise."""

    def __init__(self, cfg):
        self.cfg = cfg
        self.D = cfg.model.concat_dim
        self.S = cfg.data.S
        s = cfg.sampler
        self.num_steps, self.min_t, self.initial_dist = s.num_steps, s.min_t, s.initial_dist
        self.eps_ratio = s.eps_ratio
        self.loss_name = cfg.loss.name
        self.branch = _branch(cfg)
        self.logit_type = getattr(cfg.loss, "logit_type", "direct") if self.branch == native.BRANCH_CRM else "direct"
        self.seed = None              # set to an int for a fixed Philox key
        self.rank_stream = 0          # added to the key by the multi-GPU driver (ctdd/distributed.py)

    # -- pieces
    def _key(self):
        return (self.seed if self.seed is not None else _fresh_seed()) + self.rank_stream

    def _needs_qt0(self):
        return self.branch == native.BRANCH_CTELBO or self.logit_type != "direct"

    def _tables(self, model, times):
        """All steps' q_{t|0} in one launch, resident in HBM; beta(t) as host floats.
        `times` is float64 numpy; it reaches the schedules as float32 like `t*torch.ones(N)`."""
        t32 = torch.from_numpy(np.asarray(times, dtype=np.float64)).to(torch.float32)
        pr = model.process
        qt0 = pr.tables(t32, want_qt0=True)[0] if self._needs_qt0() else None
        return t32, qt0, pr.beta(t32).tolist()

    def _initial(self, model, N, key, std):
        return get_initial_samples(N, self.D, model.device, self.S, self.initial_dist, std, seed=key).to(torch.int32)

    def _final_argmax(self, model, x, N):
        t = torch.full((N,), float(np.float32(self.min_t)), device=x.device)
        return native.argmax(model(x.long(), t).float().contiguous())

    def _fast_tables(self, model, qt0):
        """S = 256, CT-ELBO branch or CRM branch with reverse_prob logits: derived tables for the MFMA kernel (csrc/steps_s256.hip)."""
        if self.S != 256 or qt0 is None or not getattr(self.cfg.sampler, "fast_s256", True):
            return None
        bf16 = self._step_bf16(model)
        if self.branch == native.BRANCH_CTELBO:
            return native.S256Tables(qt0, model.process.base_rate, self.eps_ratio, bf16=bf16)
        if self.logit_type == "reverse_prob":                    # CRM branch: the same contraction with a unit left scaling
            return native.S256Tables(qt0, model.process.base_rate, 0.0, crm=True, bf16=bf16)
        return None

    def _borrow(self, model):
        """Context for the sampler loops: the engine's own logits buffer is borrowed (no copy per step), and when the S = 256
        step runs its single-product bf16 mode the U-Net engine's output convolution writes the logits in bf16
        (cfg.sampler.logits_bf16, default True: half the bytes of the step's one large write and read; the rounding, 2^-8 of a
        logit, is below the bf16 network's own error)."""
        bf = (self.S == 256 and self._needs_qt0() and getattr(self.cfg.sampler, "fast_s256", True)
              and (self.branch == native.BRANCH_CTELBO or self.logit_type == "reverse_prob")
              and self._step_bf16(model) and bool(getattr(self.cfg.sampler, "logits_bf16", True)))
        return borrow_engine_output(model, bf16_logits=bf, uniform_time=True)      # (every sampler passes t * ones((N,)))

    @staticmethod
    def _net_logits(model, x, t, fast=None):
        """model(x, t) as the step kernels take it: bf16 straight through to the bf16 step (fast.bf16), fp32 otherwise."""
        # (the step kernels hand back int32 states; the U-Net engine's plans take them as they are -- no widening pass per step)
        out = model(x if (x.dtype == torch.int32 and getattr(model, "_engine_int32_states", False)) else x.long(), t)
        if out.dtype == torch.bfloat16 and fast is not None and fast.bf16:
            return out.contiguous()
        return out.float().contiguous()

    def _step_bf16(self, model):
        """cfg.sampler.step_precision: "bf16" = one bf16 product for the S x S ratio contraction (relative rate error <= 3 * 2^-8),
        "fp32" = three split-bf16 products (<= 3e-5, the parity mode), "auto" (default) = bf16 exactly when the score network
        itself runs single bf16 operands (cfg.model.engine == "hip" with engine_precision "bf16": its logits carry ~1e-2)."""
        mode = getattr(self.cfg.sampler, "step_precision", "auto")
        if mode == "auto":
            m = self.cfg.model
            default = "bf16" if hasattr(model, "data_shape") else "bf16x3"     # U-Net engine / hollow engine defaults
            return getattr(m, "engine", "hip") == "hip" and getattr(m, "engine_precision", default) == "bf16"
        if mode not in ("bf16", "fp32"):
            raise ValueError(f"sampler.step_precision must be 'auto', 'bf16' or 'fp32', got {mode!r}")
        return mode == "bf16"

    def _leap(self, model, logits, x, q_i, fast, i, beta, h, flags, key, offset, x_base=None, changed=None):
        """One fused reverse-rate / jump / update launch (MFMA path when prepared, else generic)."""
        if fast is not None:
            return native.tauleap_step_s256(logits, x, fast, i, beta, h, flags, key, offset, x_base=x_base,
                                            changed=changed)
        return native.tauleap_step(self.branch, self.logit_type, logits, x, q_i, model.process.base_rate, beta,
                                   self.eps_ratio, h, flags, key, offset, x_base=x_base, changed=changed)

    def _lbjf(self, model, logits, x, q_i, fast, i, beta, h, flags, key, offset, changed=None):
        """One Euler / LBJF step; at S = 256 the reverse rates come from the matrix-core kernel, the posterior and the
        categorical draw run on them."""
        if fast is not None:
            _, rates = native.tauleap_step_s256(logits, x, fast, i, beta, h, flags & native.STEP_CORRECTOR, key, offset,
                                                want_rates=True, want_x=False)
            return native.lbjf_from_rates(rates, x, h, None, key, offset, changed=changed)
        return native.lbjf_step(self.branch, self.logit_type, logits, x, q_i, model.process.base_rate, beta, self.eps_ratio,
                                h, flags, None, key, offset, changed=changed)

    @staticmethod
    def _t_ones(t32, i, N, device):
        return torch.full((N,), float(t32[i]), device=device, dtype=torch.float32)


@sampling_utils.register_sampler
class TauL(_GridSampler):
    """Tau-leaping with optional corrector steps (sampling.py:82-234)."""

    def __init__(self, cfg):
        super().__init__(cfg)
        s = cfg.sampler
        self.max_t = cfg.training.max_t
        self.corrector_entry_time = s.corrector_entry_time
        self.num_corrector_steps = s.num_corrector_steps
        self.is_ordinal = s.is_ordinal

    # The loop is split into begin / advance / finish so that a driver (bench.py, the multi-GPU
    # sharder) can time or interleave exact sampler steps; sample() is their composition.
    def _pipeline_parts(self, model, N):
        """Independent sub-batches the loop drives on parallel streams (cfg.sampler.pipeline_sub_batches; default: the U-Net
        engine's `engine_streams`, 2).  Samples never interact, so each sub-batch runs its own chain
        forward -> fused step -> forward -> ... with NO join per step: one chain's tau-leap launch (vector ALU / memory) runs
        under the other's convolutions (matrix cores) instead of on the critical path of both."""
        p = getattr(self.cfg.sampler, "pipeline_sub_batches", None)
        unet = hasattr(model, "_engine_forward") and getattr(self.cfg.model, "engine", "hip") == "hip" and str(model.device) != "cpu"
        if p is None:
            p = int(getattr(self.cfg.model, "engine_streams", 2)) if unet else 1
        p = int(p)
        return p if (unet and p > 1 and N % p == 0 and N // p >= 32) else 1

    def begin(self, model, N, pipeline=None):
        dev = torch.device(model.device)
        st = type("TauLState", (), {})()
        st.model, st.N, st.dev, st.key = model, N, dev, self._key()
        st.x = self._initial(model, N, st.key, self.cfg.model.Q_sigma)
        st.ts = np.concatenate((np.linspace(self.max_t, self.min_t, self.num_steps), np.array([0])))
        st.t32, st.qt0, st.betas = self._tables(model, st.ts[:-1])
        st.fast = self._fast_tables(model, st.qt0)
        # the grid's times are known here: the U-Net engine computes every step's time-projection row in one launch and the
        # per-step plans run without their time path (cfg.sampler.time_table, default True; None: the model has no such plan)
        tt = getattr(model, "engine_time_table", None) if getattr(self.cfg.sampler, "time_table", True) else None
        st.t_dev = st.t32.to(dev)
        st.temb = tt(st.t_dev) if tt is not None else None
        st.changed = torch.zeros(self.num_steps, dtype=torch.int32, device=dev)
        st.flags = native.STEP_ORDINAL if self.is_ordinal else 0
        st.sub = 1 + max(int(self.num_corrector_steps), 0)
        st.parts = self._pipeline_parts(model, N) if pipeline is None else (int(pipeline) if pipeline else 1)
        if st.parts > 1:
            P = st.parts
            n = N // P
            st.xs = [st.x[j * n:(j + 1) * n].clone() for j in range(P)]
            st.x = None                                     # (joined again by finish / state_x)
            st.changed_p = torch.zeros((P, self.num_steps), dtype=torch.int32, device=dev)
            main = torch.cuda.current_stream(dev)
            st.streams = [main] + [torch.cuda.Stream(device=dev) for _ in range(P - 1)]
            for s_ in st.streams[1:]:
                s_.wait_stream(main)                        # tables, initial state and plans' weights are ready
        return st

    def state_x(self, st):
        """The current state (N, D) of all samples (joins the sub-batch streams when the loop is pipelined)."""
        if st.parts == 1:
            return st.x
        main = st.streams[0]
        for s_ in st.streams[1:]:
            main.wait_stream(s_)
        return torch.cat(st.xs, 0)

    def advance(self, st, i):
        """Step i of the grid: network forward, fused reverse-rate/jump/update launch, correctors."""
        model = st.model
        with self._borrow(model):
            try:
                if st.temb is not None:
                    model._engine_time_row = st.temb[i]
                if st.parts == 1:
                    st.x = self._advance_one(st, i, st.x, st.N, st.key, st.changed[i:i + 1])
                    return
                for j in range(st.parts):
                    with torch.cuda.stream(st.streams[j]):
                        model._engine_slot = j
                        # a Philox key per sub-batch (the rows of a launch are numbered from 0)
                        st.xs[j] = self._advance_one(st, i, st.xs[j], st.N // st.parts, st.key + 7919 * j, st.changed_p[j, i:i + 1])
            finally:
                model._engine_slot = None
                if st.temb is not None:
                    model._engine_time_row = None

    def _advance_one(self, st, i, x, N, key, changed):
        model = st.model
        t = st.ts[i]
        h = float(np.float32(st.ts[i] - st.ts[i + 1]))
        # (with the time table the plans never read the times: a stride-0 view of the grid value instead of a fill launch per step)
        t_ones = st.t_dev[i].expand(N) if st.temb is not None else self._t_ones(st.t32, i, N, st.dev)
        q_i = st.qt0[i] if st.qt0 is not None else None
        logits = self._net_logits(model, x, t_ones, st.fast)
        x = self._leap(model, logits, x, q_i, st.fast, i, st.betas[i], h, st.flags, key, i * st.sub, changed=changed)
        if t <= self.corrector_entry_time:
            for c in range(self.num_corrector_steps):
                logits = self._net_logits(model, x, t_ones, st.fast)
                x = self._leap(model, logits, x, q_i, st.fast, i, st.betas[i], h, st.flags | native.STEP_CORRECTOR, key, i * st.sub + 1 + c)
        return x

    def finish(self, st):
        x = self.state_x(st)
        changed = st.changed if st.parts == 1 else st.changed_p.sum(0)
        if self.loss_name in ("CTElbo", "NLL"):
            x = self._final_argmax(st.model, x, st.N)
        return x.cpu().numpy().astype(int), (changed.cpu().numpy() / st.N).tolist()

    def sample(self, model, N):
        with torch.no_grad(), self._borrow(model):
            st = self.begin(model, N)
            for i in range(self.num_steps):
                self.advance(st, i)
            return self.finish(st)


@sampling_utils.register_sampler
class LBJF(_GridSampler):
    """Euler / 'LBJF' sampler: one categorical draw per dimension per step (sampling.py:238-356)."""

    def __init__(self, cfg):
        super().__init__(cfg)
        s = cfg.sampler
        self.max_t = cfg.training.max_t
        self.corrector_entry_time = s.corrector_entry_time
        self.num_corrector_steps = s.num_corrector_steps

    def sample(self, model, N):
        dev = torch.device(model.device)
        key = self._key()
        with torch.no_grad(), self._borrow(model):
            x = self._initial(model, N, key, self.cfg.model.Q_sigma)
            ts = np.concatenate((np.linspace(self.max_t, self.min_t, self.num_steps), np.array([0])))
            t32, qt0, betas = self._tables(model, ts[:-1])
            changed = torch.zeros(self.num_steps, dtype=torch.int32, device=dev)
            fast = self._fast_tables(model, qt0)
            sub = 1 + max(int(self.num_corrector_steps), 0)
            for i, t in enumerate(ts[:-1]):
                h = float(np.float32(ts[i] - ts[i + 1]))
                t_ones = self._t_ones(t32, i, N, dev)
                q_i = qt0[i] if qt0 is not None else None
                logits = self._net_logits(model, x, t_ones, fast)
                x = self._lbjf(model, logits, x, q_i, fast, i, betas[i], h, 0, key, i * sub, changed=changed[i:i + 1])
                if t <= self.corrector_entry_time:
                    for c in range(self.num_corrector_steps):
                        logits = self._net_logits(model, x, t_ones, fast)
                        x = self._lbjf(model, logits, x, q_i, fast, i, betas[i], h, native.STEP_CORRECTOR, key, i * sub + 1 + c)
            if self.loss_name == "CTElbo":
                x = self._final_argmax(model, x, N)
            return x.cpu().numpy().astype(int), (changed.cpu().numpy() / N).tolist()


@sampling_utils.register_sampler
class MidPointTauL(_GridSampler):
    """Midpoint tau-leaping: deterministic half-step drift, then a Poisson step with rates taken
    at (x', t-h/2) (sampling.py:360-526).  The reference indexes a `state_change[s][x] = s - x`
    table (loaded from a file that is not in the repo for MNIST); the kernels compute
    s - x directly, for any S."""

    def __init__(self, cfg):
        super().__init__(cfg)
        self.max_t = cfg.training.max_t
        self.is_ordinal = cfg.sampler.is_ordinal
        self.device = cfg.device

    def sample(self, model, N):
        dev = torch.device(model.device)
        key = self._key()
        with torch.no_grad(), self._borrow(model):
            x = self._initial(model, N, key, self.cfg.model.Q_sigma)
            h = (self.max_t - self.min_t) / self.num_steps
            full, half, t = [], [], self.max_t
            while t - 0.5 * h > self.min_t:
                full.append(t)
                t = t - h
            # t_05 = float32(t)*ones - 0.5*h evaluated in float32, as in the reference
            t32 = torch.tensor(full, dtype=torch.float64).to(torch.float32)
            t32_half = t32 - 0.5 * h
            nst = len(full)
            pr = model.process
            need_q = self._needs_qt0()
            q_full = pr.tables(t32, want_qt0=True)[0] if need_q else None
            q_half = pr.tables(t32_half, want_qt0=True)[0] if need_q else None
            b_full, b_half = pr.beta(t32).tolist(), pr.beta(t32_half).tolist()
            fast_half = self._fast_tables(model, q_half)
            fast_full = self._fast_tables(model, q_full)
            # per step: [final changes, dims with >= 1 jump event, dims with > 1, first-stage changes, 1to2 changes]
            cnt = torch.zeros(nst, 5, dtype=torch.int32, device=dev)
            flags = (native.STEP_ORDINAL if self.is_ordinal else 0) | native.STEP_COUNT_RAW | native.STEP_COUNT_JUMPS
            hf = float(np.float32(h))
            for i in range(nst):
                t_ones = torch.full((N,), float(t32[i]), device=dev)
                t_05 = torch.full((N,), float(t32_half[i]), device=dev)
                logits = self._net_logits(model, x, t_ones, fast_full)
                if fast_full is not None:           # S = 256: rates from the matrix-core kernel, drift on them
                    _, rates = native.tauleap_step_s256(logits, x, fast_full, i, b_full[i], hf, 0, key, 0, want_rates=True,
                                                        want_x=False)
                    x_prime = native.midpoint_from_rates(rates, x, h)
                else:
                    x_prime = native.midpoint_predict(self.branch, self.logit_type, logits, x,
                                                      q_full[i] if need_q else None, pr.base_rate, b_full[i],
                                                      self.eps_ratio, h)
                logits_p = self._net_logits(model, x_prime, t_05, fast_half)
                x_new = self._leap(model, logits_p, x, q_half[i] if need_q else None, fast_half, i, b_half[i], hf,
                                   flags, key, i, x_base=x_prime, changed=cnt[i, 0:3])
                cnt[i, 3] = (x != x_prime).sum()
                cnt[i, 4] = (x_prime != x_new).sum()
                x = x_new
            if self.loss_name == "CTElbo":
                x = self._final_argmax(model, x, N)
            raw = cnt.cpu().numpy().astype(np.float64)
            c = raw / (N * self.D)
            # (samples, change_jump, change_dim, change_dim_first, change_1to2); change_jump = share of the jumping
            # dimensions that drew more than one event, appended only when is_ordinal (sampling.py:489-495; 0/0 -> nan there too)
            with np.errstate(divide="ignore", invalid="ignore"):
                change_jump = (raw[:, 2] / raw[:, 1]).tolist() if self.is_ordinal else []
            return x.cpu().numpy().astype(int), change_jump, c[:, 0].tolist(), c[:, 3].tolist(), c[:, 4].tolist()


@sampling_utils.register_sampler
class PCTauL(_GridSampler):
    """Original tauLDR predictor-corrector (sampling.py:530-646): CT-ELBO rates, ordinal update,
    t grid from 1.0, initial std 200, bare ndarray return."""

    def __init__(self, cfg):
        super().__init__(cfg)
        self.branch, self.logit_type = native.BRANCH_CTELBO, "direct"

    def sample(self, model, N):
        s = self.cfg.sampler
        dev = torch.device(model.device)
        key = self._key()
        with torch.no_grad(), self._borrow(model):
            x = self._initial(model, N, key, 200)
            h0 = 1.0 / s.num_steps
            ts = np.linspace(1.0, s.min_t + h0, s.num_steps)
            pr = model.process
            t32, qt0, betas = self._tables(model, ts)
            fast = self._fast_tables(model, qt0)
            sub = 1 + max(int(s.num_corrector_steps), 0)
            for i, t in enumerate(ts[:-1]):
                h = ts[i] - ts[i + 1]
                logits = self._net_logits(model, x, self._t_ones(t32, i, N, dev), fast)
                x = self._leap(model, logits, x, qt0[i], fast, i, betas[i], float(np.float32(h)), native.STEP_ORDINAL,
                               key, i * sub)
                if t <= s.corrector_entry_time:
                    tc = torch.tensor([t - h], dtype=torch.float64).to(torch.float32)
                    qc = pr.tables(tc, want_qt0=True)[0][0]
                    bc = float(pr.beta(tc)[0])
                    t_c = torch.full((N,), float(tc[0]), device=dev)
                    fast_c = self._fast_tables(model, qc.unsqueeze(0)) if fast is not None else None
                    for c in range(s.num_corrector_steps):
                        logits = self._net_logits(model, x, t_c, fast_c)
                        x = self._leap(model, logits, x, qc, fast_c, 0, bc, float(np.float32(s.corrector_step_size_multiplier * h)),
                                       native.STEP_ORDINAL | native.STEP_CORRECTOR, key, i * sub + 1 + c)
            x = self._final_argmax(model, x, N)
            return x.cpu().numpy().astype(int)


@sampling_utils.register_sampler
class ExactSampling(_GridSampler):
    """Exact one-step posterior sampling for SDDM models (sampling.py:975-1061):
    x_{t-h}^d ~ sum_{x0} p_theta(x0 | x_t^{\\d}) q_{t-h|0}(.|x0) q_{t|t-h}(x_t^d | .)   per dimension.
    The (N,D,S,S) log-sum-exp of the reference is the matrix product
    softmax(logits) @ q_{t-h|0} times the column x_t of q_{t|t-h}; contraction, normalisation and the categorical
    draw (exponential race, the K7 rule) are one HIP launch per step: `ctdd_exact_step`."""

    def __init__(self, cfg):
        super().__init__(cfg)
        self.max_t = cfg.training.max_t

    def sample(self, model, N):
        dev = torch.device(model.device)
        key = self._key()
        with torch.no_grad(), borrow_engine_output(model, uniform_time=True):
            x = self._initial(model, N, key, self.cfg.model.Q_sigma).long()
            ts = np.concatenate((np.linspace(self.max_t, self.min_t, self.num_steps), np.array([0])))
            pr = model.process
            t_hi = torch.from_numpy(ts[:-1]).to(torch.float32)
            t_lo = torch.from_numpy(ts[:-1] - (ts[:-1] - ts[1:])).to(torch.float32)
            q_lo = pr.tables(t_lo, want_qt0=True)[0]                      # q_{t-h|0}  (steps,S,S)
            q_step = pr.transit_between(t_lo, t_hi)                       # q_{t|t-h}  (steps,S,S)
            q_lo, q_step = q_lo.contiguous(), q_step.contiguous()
            x = x.to(torch.int32).contiguous()
            moved = torch.zeros(self.num_steps, dtype=torch.int32, device=dev)
            for i in range(self.num_steps):
                t_ones = self._t_ones(t_hi, i, N, dev)
                logits = model(x.long(), t_ones).float().contiguous()
                # softmax, the S x S contraction with q_{t-h|0}, the column x_t of q_{t|t-h}, normalisation and the categorical
                # draw (exponential race on Philox(key, i, row, s)) in ONE launch: ctdd_exact_step
                x = native.exact_step(logits, x, q_lo[i], q_step[i], None, key, i, changed=moved[i:i + 1])
            change = moved.float() / float(N * self.D)
            return x.cpu().numpy().astype(int), change.cpu().tolist()


def lbjf_corrector_step(cfg, model, xt, t, h, N, device, xt_target=None, seed=None, E=None, want_probs=False):
    """One Euler corrector step with SDDM ratios (sampling.py:1064-1085):
    posterior = h * (exp(ll_all - ll_xt) + 1) * R_t[x_t, :] off the diagonal, clip(1 - sum, 0) on it, normalised, then one
    categorical draw per dimension.  That is the Euler posterior of the corrector rates R^ + R_t[x_t, :], i.e. ONE
    `ctdd_lbjf_step` launch with CTDD_STEP_CORRECTOR on the CRM branch (K5 + K7 fused).
    The reference body only runs when D == S (it multiplies the (N,D,S) ratios by the (N,S,S) `model.rate(t)` and so
    takes the rate row of the DIMENSION index, and it passes log-probabilities to Categorical as `probs`); nothing calls
    it.  This is the formula its docstring and the LBJF corrector (sampling.py:296-341) state: rate row of the STATE
    x_t, Categorical(logits=log posterior).  Parity: oracle restatement `oracle.samplers.lbjf_corrector_posterior`
    (reference fixture impossible: parity unpinned).  `t` may be a tensor only if all its entries are equal."""
    if torch.is_tensor(t):
        tv = t.reshape(-1)
        if tv.numel() > 1 and not bool((tv == tv[0]).all()):
            raise ValueError("lbjf_corrector_step: every sample shares one time (the reference multiplies a scalar t by ones((N,)))")
        t = float(tv[0])
    t32 = torch.tensor([t], dtype=torch.float64).to(torch.float32)
    t_ones = torch.full((N,), float(t32[0]), device=device, dtype=torch.float32)
    with torch.no_grad():
        logits = model(xt.long(), t_ones).float().contiguous()
    lt = getattr(cfg.loss, "logit_type", "direct")
    pr = model.process
    q = pr.tables(t32, want_qt0=True)[0][0] if lt != "direct" else None
    beta = float(pr.beta(t32)[0])
    key = seed if seed is not None else _fresh_seed()
    xi = xt.to(torch.int32).contiguous()
    if xt_target is None or xt_target is xt or torch.equal(xt_target, xt):
        out = native.lbjf_step(native.BRANCH_CRM, lt, logits, xi, q, pr.base_rate, beta, cfg.sampler.eps_ratio, float(h),
                               native.STEP_CORRECTOR, E, key, 0, want_probs=want_probs)
    else:
        # xt_target != xt (sampling.py:1064-1067, 1076-1080): ratios and forward-rate row of x_t, the own-state mask and the
        # diagonal at x_target.  K5 gives R^ = ratio * R_t[x_t, :]; add the corrector's forward row, zero the target's entry,
        # and the Euler posterior + categorical draw of K7 run on those rates with x_target as the state.  The rate row's own
        # entry R_t[x_t, x_t] = -sum of the row is not a jump rate: it is zeroed too (as written, the reference leaves that
        # negative number in the "posterior" whenever x_target != x_t and its log is NaN).
        rate_t = pr.rate(t32)                                               # (1, S, S) = beta(t) R
        rr, _ = native.reverse_rates(native.BRANCH_CRM, lt, logits, xi, q.unsqueeze(0) if q is not None else None, rate_t,
                                     cfg.sampler.eps_ratio, want_ratio=False)
        rr = rr + rate_t[0][xt.long()]
        tgt = xt_target.to(torch.int32).contiguous()
        rr.scatter_(-1, xt.long().unsqueeze(-1), 0.0)
        rr.scatter_(-1, tgt.long().unsqueeze(-1), 0.0)
        out = native.lbjf_from_rates(rr.contiguous(), tgt, float(h), E, key, 0, want_probs=want_probs)
    return (out[0].long(), out[1]) if want_probs else out.long()


# Names that shipped configs still use but the reference never registers: resolve
# them to the sampler the authors' scripts substitute by hand.
for _alias, _cls in (("TauLeaping", TauL), ("ElboTauL", TauL), ("CRMTauL", TauL), ("LBJFSampling", LBJF),
                     ("CRMLBJF", LBJF), ("ElboLBJF", LBJF), ("CRMebmLBJF", LBJF), ("PCTauLeaping", PCTauL)):
    sampling_utils.register_alias(_alias, _cls)
