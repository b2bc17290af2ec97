"""Registry of sampler classes, keyed by class name (reference lib/sampling/sampling_utils.py)."""
_SAMPLERS = {}


def register_sampler(cls):
    name = cls.__name__
    if name in _SAMPLERS:
        raise ValueError(f"{name} is already registered!")
    _SAMPLERS[name] = cls
    return cls

def get_sampler(cfg):
    return _SAMPLERS[cfg.sampler.name](cfg)


def register_alias(alias, cls):
    """Stale sampler names still found in shipped configs resolve to a registered class."""
    _SAMPLERS.setdefault(alias, cls)

