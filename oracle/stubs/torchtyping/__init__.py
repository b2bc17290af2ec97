"""Build-owned test stub: annotation-only stand-in for `torchtyping` (absent in this image).
Contains no reference code. Used only by oracle/gen_golden.py to import the reference here."""


class _TT:
    def __getitem__(self, item):
        return object

    def __call__(self, *a, **k):
        return object


TensorType = _TT()


def patch_typeguard():
    return None
