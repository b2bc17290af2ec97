"""Registry of dataset classes, keyed by class name (reference lib/datasets/dataset_utils.py)."""
_DATASETS = {}


def register_dataset(cls):
    name = cls.__name__
    if name in _DATASETS:
        raise ValueError(f"{name} is already registered!")
    _DATASETS[name] = cls
    return cls

def get_dataset(cfg, device, root=None):
    return _DATASETS[cfg.data.name](cfg, device, root)

