"""Checkpoint / config IO in the reference's on-disk formats (lib/utils/bookkeeping.py:343-394):
`<dir>/<YYYY-MM-DD>/model_<n_iter>.pt` = torch.save({"model": state_dict incl. the three EMA
keys, "optimizer", "n_iter"}) and `<dir>/<YYYY-MM-DD>/config_001.yaml`, so artefacts written by the
reference load here and vice versa.  YAML goes through ruamel when present, PyYAML otherwise (the
files are plain mappings of scalars and lists).  The legacy tauLDR experiment-folder helpers and
TensorBoard writers of the reference (17-340) are unused by its scripts and not provided."""
import os
from datetime import datetime

import torch

from ctdd.config_dict import ConfigDict

try:  # pragma: no cover - not installed in this image
    import ruamel.yaml as _ruamel
except ImportError:
    _ruamel = None
import yaml as _pyyaml


def _today():
    return datetime.now().strftime("%Y-%m-%d")


def save_state(state: dict, save_dir) -> None:
    path = os.path.join(save_dir, _today())
    os.makedirs(path, exist_ok=True)
    ckpt = {"model": state["model"].state_dict(), "optimizer": state["optimizer"].state_dict(), "n_iter": state["n_iter"]}
    torch.save(ckpt, os.path.join(path, f"model_{state['n_iter']}.pt"))


def _numpy_scalar_globals():
    """The reference's warm-up writes a numpy float64 learning rate into the optimizer's param_groups
    (training.py:31-33), so its checkpoints pickle a numpy scalar.  Reconstructing a numpy scalar executes nothing from the
    file: allow-list exactly those constructors (numpy 1.x and 2.x module paths) for the weights-only loader."""
    import numpy as np
    out = [np.dtype]
    for modname in ("numpy._core.multiarray", "numpy.core.multiarray"):
        try:
            mod = __import__(modname, fromlist=["scalar"])
            out.append(mod.scalar)
        except (ImportError, AttributeError):
            pass
    for name in ("Float64DType", "Float32DType", "Int64DType", "Int32DType", "BoolDType"):
        if hasattr(np, "dtypes") and hasattr(np.dtypes, name):
            out.append(getattr(np.dtypes, name))
    return out


def load_state(state: dict, checkpoint_path: str, mapping=torch.device("cuda")) -> dict:
    # our own / the reference's checkpoints hold tensors, python and numpy scalars and a list of tensors
    # (ema_shadow_params): loadable with weights_only=True (nothing from the file is executed)
    with torch.serialization.safe_globals(_numpy_scalar_globals()):
        ckpt = torch.load(checkpoint_path, map_location=mapping, weights_only=True)
    state["model"].load_state_dict(ckpt["model"])
    state["optimizer"].load_state_dict(ckpt["optimizer"])
    state["n_iter"] = ckpt["n_iter"]
    return state


def save_config(config, config_dir: str) -> None:
    path = os.path.join(config_dir, _today())
    os.makedirs(path, exist_ok=True)
    data = config.to_dict() if hasattr(config, "to_dict") else dict(config)
    with open(os.path.join(path, "config_001.yaml"), "w") as f:
        if _ruamel is not None:
            _ruamel.YAML().dump(data, f)
        else:
            _pyyaml.safe_dump(data, f, default_flow_style=False, sort_keys=False)


def load_config(config_dir: str):
    with open(config_dir, "r") as f:
        data = _ruamel.YAML().load(f) if _ruamel is not None else _pyyaml.safe_load(f)
    return ConfigDict(data)
