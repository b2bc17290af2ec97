"""Build-owned test stub: minimal attr-dict `ConfigDict` (ml_collections is absent in this image)."""


class ConfigDict(dict):
    def __init__(self, initial=None, **kw):
        super().__init__()
        if initial:
            for k, v in dict(initial).items():
                self[k] = ConfigDict(v) if isinstance(v, dict) and not isinstance(v, ConfigDict) else v
        for k, v in kw.items():
            self[k] = v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def to_dict(self):
        return {k: (v.to_dict() if isinstance(v, ConfigDict) else v) for k, v in self.items()}
