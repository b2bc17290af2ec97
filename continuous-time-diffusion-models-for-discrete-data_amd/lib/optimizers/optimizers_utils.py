"""Registry of optimizer classes, keyed by class name (reference lib/optimizers/optimizers_utils.py)."""
_OPTIMIZERS = {}


def register_optimizer(cls):
    name = cls.__name__
    if name in _OPTIMIZERS:
        raise ValueError(f"{name} is already registered!")
    _OPTIMIZERS[name] = cls
    return cls

def get_optimizer(params, cfg):
    return _OPTIMIZERS[cfg.optimizer.name](params, cfg)

