"""Times the SJLT sketch kernels alone (hssk_sjlt_sketch, both products) at the bench size and a few tile choices.
usage (GPU box): python tools/sjlt_only.py [n] [dn] [nnz]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from strumpack_amd import _loader  # noqa: E402
from strumpack_amd import hssk as K  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
dn = int(sys.argv[2]) if len(sys.argv) > 2 else 192
nnz = int(sys.argv[3]) if len(sys.argv) > 3 else 4
hk = K.Hssk(_loader.lib_path())
hk.lib.hssk_last_dgemm_ms.restype = __import__("ctypes").c_float
dA = hk.empty((n, n))
hk.check(hk.lib.hssk_fill_toeplitz(hk.ctx, dA.ptr, n, n, b"T"))
rng = np.random.default_rng(0)
chunk = dn // nnz
cols = (rng.integers(0, chunk, size=(nnz, n)) + chunk * np.arange(nnz)[:, None]).astype(np.int64)
neg = rng.integers(0, 2, size=(nnz, n)).astype(np.int64)
nq = 4 if nnz <= 4 else 8
pat = np.full((n, nq), dn, dtype=np.int64)
pat[:, :nnz] = (cols | (neg << 31)).T
pat = pat.astype(np.uint32).view(np.int32)
dpat = hk.array(np.ascontiguousarray(pat).reshape(-1), dtype=np.int32)
dS = hk.empty((dn, n))
hk.sync()
gb = 8.0 * n * n * 1e-9


def run(trans, reps=3):
    best = 1e9
    for _ in range(reps):
        hk.check(hk.lib.hssk_sjlt_sketch(hk.ctx, trans, n, n, dA.ptr, n, dpat.ptr, nnz, dn, dS.ptr, dn))
        hk.sync()
        best = min(best, hk.lib.hssk_last_dgemm_ms(hk.ctx))
    return best


def sweep(trans, var, vals, mode=None):
    for v in vals:
        os.environ[var] = str(v)
        if mode is not None:
            os.environ["HSSK_SJLT_MODE"] = str(mode)
        try:
            ms = run(trans)
            print("trans=%d %s=%s mode=%s  %.2f ms  %.0f GB/s (algorithmic, %.1f GB)" % (trans, var, v, mode or 0, ms, gb / (ms * 1e-3), gb), flush=True)
        except Exception as e:  # noqa: BLE001
            print("trans=%d %s=%s failed: %s" % (trans, var, v, e), flush=True)
        os.environ.pop("HSSK_SJLT_MODE", None)
    os.environ.pop(var, None)


# variants (hssk_sjlt.hip, launch_sketch): n: 0 = 16 waves x 16 loads, 1 = 16 x 8, 2 = 8 x 16, -1 = small-tile kernel
#           t: 0 = 8 waves x 128-byte runs, 1 = 16 x 128, 2 = 16 x 64, -1 = small-tile kernel;  mode 1 = plain LDS
#           read-modify-write instead of ds_add_f64, mode 2 = no LDS updates (streaming rate of the access pattern)
sweep(1, "HSSK_SJLT_TV", (-1, 2, 1, 0))
sweep(1, "HSSK_SJLT_TV", (0,), mode=1)
sweep(1, "HSSK_SJLT_TV", (0,), mode=2)
sweep(0, "HSSK_SJLT_NV", (-1, 2, 1, 0))
sweep(0, "HSSK_SJLT_NV", (0,), mode=1)
sweep(0, "HSSK_SJLT_NV", (0,), mode=2)
run(0, reps=1)   # leave the product path's result in dS for the check below
# checksum against a column of the exact product (row sums of the pattern applied to one column of A)
S = dS.get()
i = np.arange(n)
k0 = 12345 % n
col = 1.0 / (1.0 + np.abs(i - k0))
R = np.zeros((n, dn))
for q in range(nnz):
    R[np.arange(n), cols[q]] = np.where(neg[q] == 1, -1.0, 1.0)
print("check row %d: max err %.2e" % (k0, np.abs(S[:, k0] - col @ R).max()))
