"""Stage clocks of getrf_wg_kernel (library built with -DLUW_TIMING as strumpack_amd/lib/libstrumpack_amd_timing.so; the
kernel prints them)."""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import numpy as np
from strumpack_amd import hssk as K
hk = K.Hssk(os.path.join(root, "strumpack_amd", "lib", "libstrumpack_amd_timing.so"))
rng = np.random.default_rng(3)
for n in (256, 256, 472):
    A = rng.standard_normal((n, n))
    dA = hk.array(np.asfortranarray(A)); dpiv = hk.empty((n,), np.int32); dinfo = hk.empty((1,), np.int32)
    hk.batch("hssk_getrf_vbatched", [K.LuDesc(dA.ptr, n, n, dpiv.ptr, dinfo.ptr)]); hk.sync()
