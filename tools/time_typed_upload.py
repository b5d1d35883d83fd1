#!/usr/bin/env python3
"""Construction time of the same host-resident matrix given as double and as float (SP_d_ / SP_s_struct_from_dense): the
float operand crosses the link in its own format.  Prints one JSON line."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from strumpack_amd import _loader, capi  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
L = capi.load(_loader.lib_path())
i = np.arange(n, dtype=np.float32)
A32 = np.asfortranarray(1.0 / (1.0 + np.abs(i[:, None] - i[None, :])), dtype=np.float32)
A64 = np.asfortranarray(A32.astype(np.float64))
vp = C.c_void_p
out = {"n": n}
for p, A in (("d", A64), ("s", A32)):
    fn = lambda name: getattr(L, "SP_%s_struct_%s" % (p, name))
    o = capi.CSPOptions()
    fn("default_options")(C.byref(o))
    o.type, o.rel_tol, o.abs_tol, o.leaf_size, o.verbose = 0, 1e-4, 1e-8, 256, 0
    fn("from_dense").argtypes = [C.POINTER(vp), C.c_int, C.c_int, vp, C.c_int, C.POINTER(capi.CSPOptions)]
    fn("destroy").argtypes = [C.POINTER(vp)]
    ts = []
    for _ in range(4):
        h = vp()
        t0 = time.perf_counter()
        assert fn("from_dense")(C.byref(h), n, n, A.ctypes.data, n, C.byref(o)) == 0
        ts.append(time.perf_counter() - t0)
        fn("destroy")(C.byref(h))
    out[p] = {"ms": [round(1e3 * t, 2) for t in ts], "operand_GB": A.nbytes / 1e9, "best_GBps": A.nbytes / 1e9 / min(ts)}
print(json.dumps(out))
