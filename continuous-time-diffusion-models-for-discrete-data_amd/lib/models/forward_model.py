"""Forward CTMC processes behind the reference's class names (lib/models/forward_model.py).

Same constructor signature `(cfg, device)` and the same public surface -- `transition(t)`,
`rate(t)`, `rate_mat(y, t)`, `transit_between(t1, t2)`, attributes `S`, `device`, `base_rate` /
`rate_matrix`, `eigvals`, `eigvecs`, `inv_eigvecs` -- but the tables are produced on the GPU by
libctdd's K1 kernel (ctdd/process.py); there is no CPU path."""
from ctdd.process import DeviceForwardProcess


class _ProcessMixin:
    """Shared delegation; `self.process` is the device-resident engine object."""

    def _init_process(self, kind, cfg, device, **params):
        self.S = cfg.data.S
        self.device = device
        self.process = DeviceForwardProcess(kind, self.S, device, **params)
        pr = self.process
        self.eigvals, self.eigvecs, self.inv_eigvecs = pr.eigvals, pr.eigvecs, pr.right

    def rate(self, t):
        return self.process.rate(t)

    def rate_mat(self, y, t):
        return self.process.rate_mat(y, t)

    def transition(self, t):
        return self.process.transition(t)

    def transit_between(self, t1, t2):
        return self.process.transit_between(t1, t2)

    def _rate_scalar(self, t):
        return self.process.beta(t)

    def _integral_rate_scalar(self, t):
        return self.process.integral(t)


class GaussianTargetRate(_ProcessMixin):
    """forward_model.py:207-306."""

    def __init__(self, cfg, device):
        m = cfg.model
        self.rate_sigma, self.Q_sigma, self.time_exp, self.time_base = m.rate_sigma, m.Q_sigma, m.time_exp, m.time_base
        self._init_process("gaussian", cfg, device, rate_sigma=m.rate_sigma, Q_sigma=m.Q_sigma,
                           time_exp=m.time_exp, time_base=m.time_base)
        self.base_rate = self.process.base_rate


class UniformRate(_ProcessMixin):
    """forward_model.py:78-129."""

    def __init__(self, cfg, device):
        self.rate_const = cfg.model.rate_const
        self._init_process("uniform", cfg, device, rate_const=self.rate_const)
        self.rate_matrix = self.process.base_rate


class UniformVariantRate(_ProcessMixin):
    """forward_model.py:132-204 (time-warped uniform: t_func in {log_sqr, sqrt_cos, log})."""

    def __init__(self, config, device):
        self.config = config
        self.rate_const = config.model.rate_const
        self.t_func = config.model.t_func
        kw = dict(rate_const=self.rate_const, t_func=self.t_func)
        if self.t_func == "log":
            self.time_base, self.time_exp = config.model.time_base, config.model.time_exp
            kw.update(time_base=self.time_base, time_exp=self.time_exp)
        elif self.t_func not in ("log_sqr", "sqrt_cos"):
            raise ValueError("Unknown t_func %s" % self.t_func)
        self._init_process("univar", config, device, **kw)
        self.rate_matrix = self.process.base_rate


class BirthDeathForwardBase(_ProcessMixin):
    """forward_model.py:9-75."""

    def __init__(self, cfg, device):
        self.sigma_min, self.sigma_max = cfg.model.sigma_min, cfg.model.sigma_max
        self._init_process("birthdeath", cfg, device, sigma_min=self.sigma_min, sigma_max=self.sigma_max)
        self.base_rate = self.process.base_rate
        self.base_eigvals, self.base_eigvecs = self.process.eigvals, self.process.eigvecs
