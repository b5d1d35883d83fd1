"""Locates the product library (hipcc-built for gfx950).  There is NO fallback: if the library is
missing or no HIP device is visible, callers get an exception."""
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libstrumpack_amd.so")


def lib_path():
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` (hipcc --offload-arch=gfx950).  strumpack_amd has no CPU fallback.")
    return LIB_PATH
