"""The sketch GEMM on ONE RANK'S SHARD of the headline problem (N = 1e5 split over G ranks: S^T(192 x N/G) = R^T op(A_shard)),
K-split of the main group swept inside one process.   usage: dgemm_shard.py [G] [splits, e.g. "0,6,8,12,16,24,32"]"""
import os
import sys
import time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from strumpack_amd import _loader  # noqa: E402
from strumpack_amd import hssk as K  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n, d = 100000, 192
nloc = n // G
hk = K.Hssk(_loader.lib_path())
dAc = hk.empty((n, nloc))     # column block A(:, shard)       -> Sc rows of the shard (transB = 0)
dAr = hk.empty((nloc, n))     # row block    A(shard, :)       -> Sr rows of the shard (transB = 1)
hk.check(hk.lib.hssk_fill_toeplitz_block(hk.ctx, dAc.ptr, n, nloc, n, 0, 0, b"T"))
hk.check(hk.lib.hssk_fill_toeplitz_block(hk.ctx, dAr.ptr, nloc, n, nloc, 0, 0, b"T"))
dR = hk.empty((d, n))
hk.check(hk.lib.hssk_randn(hk.ctx, dR.ptr, d, n, d, 0, n, 1))
dS = hk.empty((d, nloc))
ideal = 2.0 * d * nloc * n / 78.6e12 * 1e3
for sp in (sys.argv[2] if len(sys.argv) > 2 else "0,6,8,12,16,24,32,48").split(","):
    if int(sp):
        os.environ["HSSK_DGEMM_SPLIT"] = sp
    else:
        os.environ.pop("HSSK_DGEMM_SPLIT", None)
    res = []
    for tb, dA, ld in ((1, dAr, nloc), (0, dAc, n)):
        for rep in range(2):
            hk.sync()
            t0 = time.perf_counter()
            hk.check(hk.lib.hssk_dgemm(hk.ctx, tb, d, nloc, n, 1.0, dR.ptr, d, dA.ptr, ld, 0.0, dS.ptr, d))
            hk.sync()
            wall = (time.perf_counter() - t0) * 1e3
        res.append((hk.lib.hssk_last_dgemm_ms(hk.ctx), wall))
    print("G %d split %2s | T main %.3f call %.3f | N main %.3f call %.3f | calls %.3f ms (roof %.3f ms for both: %.3f)"
          % (G, sp, res[0][0], res[0][1], res[1][0], res[1][1], res[0][1] + res[1][1], 2 * ideal, 2 * ideal / (res[0][1] + res[1][1])), flush=True)
