"""Build-owned test stub (torchvision absent). No datasets are ever touched by the oracle."""
from . import transforms, datasets, utils  # noqa: F401
