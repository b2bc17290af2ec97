from .. import ConfigDict  # noqa: F401
