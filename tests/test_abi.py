"""The C-ABI library loads without a GPU and exports every symbol include/gnnmp.h declares (no compute calls here)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "gnnmp.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gnnmp_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_expected_entry_points():
    from gnnmp import _lib
    assert header_symbols() == sorted(_lib.SYMBOLS)


def test_library_loads_and_exports_every_declared_symbol():
    import __graft_entry__ as ge
    from gnnmp import _lib
    if not os.path.exists(_lib.LIB_PATH):
        ge.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for sym in header_symbols():
        assert hasattr(lib, sym), f"libgnnmp.so does not export {sym}"
    lib.gnnmp_version.restype = ctypes.c_int
    assert lib.gnnmp_version() == 100
    lib.gnnmp_last_error.restype = ctypes.c_char_p
    assert isinstance(lib.gnnmp_last_error(), bytes)


def test_argument_validation_needs_no_gpu():
    """status codes for bad arguments are produced before any HIP call"""
    from gnnmp import _lib
    lib = _lib.load()
    h = ctypes.c_void_p()
    assert lib.gnnmp_plan_create(ctypes.byref(h), None, None, 3, 1, 4, 4, 0, 0, 0, None) == _lib.EINVAL
    assert b"idx_bytes" in lib.gnnmp_last_error()
    assert lib.gnnmp_plan_create(ctypes.byref(h), None, None, 8, 2, 4, 4, 0, 0, 0, None) == _lib.EINVAL
    assert lib.gnnmp_plan_create(ctypes.byref(h), None, None, 8, 1, 4, 5, 0, 1, 0, None) == _lib.EINVAL
    assert lib.gnnmp_plan_create(ctypes.byref(h), None, None, 8, 1, 4, 4, 2**31, 0, 0, None) in (_lib.EINVAL, _lib.EUNSUPPORTED)
    assert lib.gnnmp_propagate_f32(None, 0, 0, None, None, None, None, None, 4, None) == _lib.EINVAL
    assert lib.gnnmp_gather_f32(None, None, 5, 1, 1, None, 4, None) == _lib.EINVAL
    assert lib.gnnmp_dense_f32(None, None, 0, 0, None, None, 0, 0, 0, None, 0, None, 1, 1, None) == _lib.EINVAL
    assert lib.gnnmp_plan_destroy(None) == _lib.OK


def test_product_path_refuses_to_run_without_gpu():
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import gnnmp
    with pytest.raises(RuntimeError):
        gnnmp.GNNGraph([1, 2], [2, 1])


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under graphneuralnetworks.jl_amd/ may reference it"""
    pkg = os.path.join(ROOT, "graphneuralnetworks.jl_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".jl")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                for pat in ("import oracle", "from oracle", "libgnn_oracle", "oracle/", "oracle."):
                    assert pat not in txt, f"{f} references the oracle ({pat!r})"
