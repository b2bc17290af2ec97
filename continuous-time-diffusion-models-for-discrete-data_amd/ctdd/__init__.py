"""ctdd -- host side of the MI355X engine: ctypes binding of libctdd.so (include/ctdd.h) and the
device-resident forward-process / sampler drivers used by the `lib.*` registry mirror."""
from .native import lib_path, load, CtddError  # noqa: F401
