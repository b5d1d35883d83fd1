"""Times partial_factor / Schur_update / Schur_product_direct (device operands) at the bench size.
usage (GPU box): python tools/schur_timing.py [n] [c]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from strumpack_amd import _loader, capi  # noqa: E402
from strumpack_amd import hssk as K  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
c = int(sys.argv[2]) if len(sys.argv) > 2 else 64
L = capi.load(_loader.lib_path())
hk = K.Hssk(_loader.lib_path())
dA = hk.empty((n, n))
hk.check(hk.lib.hssk_fill_toeplitz(hk.ctx, dA.ptr, n, n, b"T"))
hk.sync()
o = capi.StructuredMatrix.options(L, rel_tol=1e-4, abs_tol=1e-8, leaf_size=256, max_rank=50000)
h = capi.StructuredMatrix.hss_options(L, random_engine="philox")
H = capi.StructuredMatrix.from_dense_device(L, dA.ptr, n, n, o, h)
dA.free()


def timed(f, reps=5):
    ts = []
    for _ in range(reps):
        t = time.perf_counter()
        f()
        ts.append(time.perf_counter() - t)
    return 1e3 * min(ts)


print("partial_factor   %.2f ms" % timed(H.partial_factor))
d = H.schur_dims()
print("dims", d)
t = time.perf_counter()
Th, DU, Ph, Vh = H.schur_update()
print("Schur_update (first, with D2H of the factors) %.2f ms" % (1e3 * (time.perf_counter() - t)))
n1 = d["n1"]
dR = hk.empty((n1, c))
hk.check(hk.lib.hssk_randn(hk.ctx, dR.ptr, n1, c, n1, 0, c, 11))
dSr, dSc = hk.empty((n1, c)), hk.empty((n1, c))
hk.sync()


def prod():
    assert L.SPX_d_struct_schur_product_direct(H.h, c, dR.ptr, n1, dSr.ptr, n1, dSc.ptr, n1, 1) == 0


print("Schur_product_direct (c = %d, device operands) %.2f ms" % (c, timed(prod)))
# consistency on the device result: direct product vs the low-rank form applied on the host
R = dR.get()
Sr = dSr.get()
H11R = H.mult_child(1, R)
ref = H11R - Th @ (Vh.T @ (Ph.T @ R))
print("||Sr - (H11 R - Theta Vhat^T Phi^T R)|| / ||Sr|| = %.2e" % (np.linalg.norm(Sr - ref) / np.linalg.norm(ref)))
