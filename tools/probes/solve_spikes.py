"""Latency distribution of repeated device solves / applies on one factored matrix (N = 1e5)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from strumpack_amd import _loader, capi, hssk as K, dist as sdist
L = capi.load(_loader.lib_path()); hk = K.Hssk(_loader.lib_path())
n = 100000
dA = hk.empty((n, n)); hk.check(hk.lib.hssk_fill_toeplitz(hk.ctx, dA.ptr, n, n, b"T"))
dB = hk.empty((n, 1)); dY = hk.empty((n, 1))
hk.check(hk.lib.hssk_randn(hk.ctx, dB.ptr, n, 1, n, 0, 1, 7)); hk.sync()
o = capi.StructuredMatrix.options(L, rel_tol=1e-4, abs_tol=1e-8, leaf_size=256, max_rank=50000)
h = capi.StructuredMatrix.hss_options(L, random_engine="philox")
H = sdist.from_dense_device(L, dA.ptr, n, n, o, h, None); H.factor()
for name, fn in (("solve", lambda: H.solve_device(dB.ptr, 1)), ("apply", lambda: H.mult_device(dB.ptr, dY.ptr, 1))):
    ts = []
    for i in range(60):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    ts = np.array(ts)
    print(name, "median %.3f  min %.3f  max %.3f  #>5ms %d  at %s" % (np.median(ts), ts.min(), ts.max(), (ts > 5).sum(), np.nonzero(ts > 5)[0].tolist()), flush=True)
