"""Attribute-access config container with the slice of ml_collections.ConfigDict's behaviour the
reference relies on (`cfg.a.b`, `"key" in cfg.model`, item access, `.to_dict()`, in-place
mutation by the scripts).  ml_collections itself is used when importable."""
try:  # pragma: no cover - not installed in this image
    from ml_collections import ConfigDict  # type: ignore
except ImportError:

    class ConfigDict(dict):
        def __init__(self, initial=None, **kw):
            super().__init__()
            for k, v in dict(initial or {}, **kw).items():
                self[k] = v

        def __setitem__(self, k, v):
            if isinstance(v, dict) and not isinstance(v, ConfigDict):
                v = ConfigDict(v)
            super().__setitem__(k, v)

        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError as e:
                raise AttributeError(k) from e

        def __setattr__(self, k, v):
            self[k] = v

        def __delattr__(self, k):
            del self[k]

        def to_dict(self):
            return {k: (v.to_dict() if isinstance(v, ConfigDict) else v) for k, v in self.items()}

        def copy_and_resolve_references(self):
            return ConfigDict(self.to_dict())
