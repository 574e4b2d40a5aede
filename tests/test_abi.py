"""CPU: the C-ABI library loads and exports every symbol include/s2t_mi355.h declares."""
import ctypes
import os

from speech2text_amd import _native


def test_header_symbols_exported():
    protos = _native.parse_header()
    assert len(protos) >= 10
    assert os.path.exists(_native.LIB_PATH), "build first: python -m speech2text_amd.csrc.build"
    lib = ctypes.CDLL(_native.LIB_PATH)
    for name in protos:
        assert hasattr(lib, name), f"{name} declared in the header but not exported"


def test_ctc_workspace_size_is_pure_host_function():
    lib = _native.lib()
    assert lib.s2t_ctc_workspace_floats(2, 10, 3) == 2 * 10 + 3 * 2 * 10 * 7 + 2   # lse + lp,alpha,beta + nll
