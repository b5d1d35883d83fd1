"""strumpack_amd -- Python plumbing around the native MI355X HSS engine (ctypes bindings).

PyTorch-ROCm bundles its own HIP runtime with the same SONAME as /opt/rocm's: whichever copy is
loaded first serves the whole process.  When torch is installed we therefore import it BEFORE the
native library is dlopen'ed, so that device memory, streams and RCCL (torch.distributed) all live in
one runtime.  C / C++ / Fortran users of libstrumpack_amd.so are not affected.
"""
try:  # noqa: SIM105
    import torch  # noqa: F401
except Exception:  # pragma: no cover - torch is optional for the C-ABI itself
    pass
