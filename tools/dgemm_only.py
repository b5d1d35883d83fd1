"""The two sketch GEMMs of BASELINE configs[2] on their own (for PMC passes): S^T = R^T op(A), N = 1e5, d = 192."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from strumpack_amd import _loader  # noqa: E402
from strumpack_amd import hssk as K  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
d = 192
hk = K.Hssk(_loader.lib_path())
dA = hk.empty((n, n))
hk.check(hk.lib.hssk_fill_toeplitz(hk.ctx, dA.ptr, n, n, b"T"))
dR = hk.empty((d, n))
hk.check(hk.lib.hssk_randn(hk.ctx, dR.ptr, d, n, d, 0, n, 1))
dS = hk.empty((d, n))
order = [int(c) for c in sys.argv[2]] if len(sys.argv) > 2 else [1, 0, 1, 0]
for tb in order:
    hk.check(hk.lib.hssk_dgemm(hk.ctx, tb, d, n, n, 1.0, dR.ptr, d, dA.ptr, n, 0.0, dS.ptr, d))
    hk.sync()
    print(tb, hk.lib.hssk_last_dgemm_ms(hk.ctx), flush=True)
