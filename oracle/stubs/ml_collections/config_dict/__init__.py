from .. import ConfigDict  # noqa: F401
from . import config_dict  # noqa: F401
