"""Registry of train-step classes, keyed by class name (reference lib/training/training_utils.py)."""
_TRAINSTEPS = {}


def register_train_step(cls):
    name = cls.__name__
    if name in _TRAINSTEPS:
        raise ValueError(f"{name} is already registered!")
    _TRAINSTEPS[name] = cls
    return cls

def get_train_step(cfg):
    return _TRAINSTEPS[cfg.training.train_step_name](cfg)

