"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/ctdd.h
declares (no compute calls without a GPU)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _built_lib():
    from ctdd import native
    if not os.path.exists(native.lib_path()):
        import __graft_entry__ as ge
        ge.build()
    return native


def test_header_symbols_exported():
    native = _built_lib()
    hdr = "".join(open(os.path.join(ROOT, "include", h)).read() for h in sorted(os.listdir(os.path.join(ROOT, "include"))))
    declared = set(re.findall(r"^\s*(?:int|int64_t|const char\*)\s+(ctdd_\w+)\s*\(", hdr, flags=re.M))
    assert len(declared) >= 15
    lib = ctypes.CDLL(native.lib_path())
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/ctdd.h but not exported"
    assert declared == set(native.EXPORTS), "ctypes binding and header disagree"
    assert native.load().ctdd_abi_version() == 1


def test_product_path_refuses_cpu_tensors():
    import pytest
    import torch
    native = _built_lib()
    with pytest.raises(native.CtddError):
        native.argmax(torch.zeros(1, 2, 3))
    with pytest.raises(native.CtddError):
        native.noise_categorical(torch.zeros(1, 3, 3), torch.zeros(1, 2, dtype=torch.int32))
