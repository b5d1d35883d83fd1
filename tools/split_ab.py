"""A/B of the K-split of the sketch GEMM main group inside one process (same box, same thermal state)."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from strumpack_amd import _loader  # noqa: E402
from strumpack_amd import hssk as K  # noqa: E402
import time

n, d = 100000, 192
hk = K.Hssk(_loader.lib_path())
dA = hk.empty((n, n))
hk.check(hk.lib.hssk_fill_toeplitz(hk.ctx, dA.ptr, n, n, b"T"))
dR = hk.empty((d, n))
hk.check(hk.lib.hssk_randn(hk.ctx, dR.ptr, d, n, d, 0, n, 1))
dS = hk.empty((d, n))
splits = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1,7,1,7,3,12,1,7").split(",")]
for sp in splits:
    os.environ["HSSK_DGEMM_SPLIT"] = str(sp)
    res = []
    for tb in (1, 0):
        hk.sync()
        t0 = time.perf_counter()
        hk.check(hk.lib.hssk_dgemm(hk.ctx, tb, d, n, n, 1.0, dR.ptr, d, dA.ptr, n, 0.0, dS.ptr, d))
        hk.sync()
        res.append((hk.lib.hssk_last_dgemm_ms(hk.ctx), (time.perf_counter() - t0) * 1e3))
    print("split %2d  T main %.2f (call %.2f)  N main %.2f (call %.2f)  sum of calls %.2f ms" % (sp, res[0][0], res[0][1], res[1][0], res[1][1], res[0][1] + res[1][1]), flush=True)
