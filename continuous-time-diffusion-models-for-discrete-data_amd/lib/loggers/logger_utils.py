"""Registry of logger classes, keyed by class name (reference lib/loggers/logger_utils.py)."""
_LOGGERS = {}


def register_logger(cls):
    name = cls.__name__
    if name in _LOGGERS:
        raise ValueError(f"{name} is already registered!")
    _LOGGERS[name] = cls
    return cls

def get_logger(name):
    return _LOGGERS[name]

