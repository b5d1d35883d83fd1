"""CPU: the product library (hipcc, gfx950) loads and exports every symbol that include/*.h
declares -- no compute calls (there is no GPU in the build container)."""
import ctypes
import os
import re

import pytest

from strumpack_amd import _loader, capi, hssk
from strumpack_amd import kernel as kernel_mod

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = set(re.findall(r"\b((?:hssk|SP_d_struct|SPX_d_struct|STRUMPACK|SPX)_\w+)\s*\(", txt))
    return sorted(n for n in names if not n.startswith("SPX_DECLARE"))   # (declaration macro, not a function)


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_loader.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return ctypes.CDLL(_loader.lib_path())


def test_hssk_symbols(lib):
    names = declared("hssk.h")
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/hssk.h but not exported"
    assert set(hssk.HSSK_SYMBOLS) <= set(names)


def test_structured_c_api_symbols(lib):
    names = declared(os.path.join("structured", "StructuredMatrix.h"))
    for n in names:
        assert hasattr(lib, n), f"{n} declared but not exported"
    assert set(capi.SP_SYMBOLS) <= set(names)


def test_single_precision_and_complex_c_api_symbols(lib):
    """SP_s_ / SP_c_ / SP_z_struct_*: the names of the reference's C header (structured/StructuredMatrix.h:103-602),
    declared through one macro in include/structured/StructuredMatrix.h"""
    txt = open(os.path.join(ROOT, "include", "structured", "StructuredMatrix.h")).read()
    stems = re.findall(r"SP_##P##_struct_(\w+)\(", txt)
    assert set(stems) >= {"default_options", "destroy", "rows", "cols", "memory", "nonzeros", "rank", "from_dense",
                          "from_elements", "mult", "factor", "solve", "shift"}
    for p in "scz":
        assert "SPX_DECLARE_C_API(%s," % p in txt
        for st in set(stems):
            assert hasattr(lib, "SP_%s_struct_%s" % (p, st)), "SP_%s_struct_%s not exported" % (p, st)


def test_kernel_c_api_symbols(lib):
    names = declared(os.path.join("kernel", "Kernel.h"))
    assert "STRUMPACK_kernel_fit_HSS_double" in names
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/kernel/Kernel.h but not exported"
    assert set(kernel_mod.KERNEL_SYMBOLS) <= set(names)


def test_no_cpu_fallback(lib):
    """Creating a context without a HIP device must fail loudly (no silent CPU path)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    ctx = ctypes.c_void_p()
    lib.hssk_ctx_create.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int]
    lib.hssk_last_error.restype = ctypes.c_char_p
    assert lib.hssk_ctx_create(ctypes.byref(ctx), 0) != 0
    assert b"no HIP device" in lib.hssk_last_error() or b"HIP error" in lib.hssk_last_error()
