"""hssk_id_vbatched on its own: per-step cost of the register-resident truncated QRCP (slope over the rank at which the tolerance stops it) for the panel
shapes of BASELINE configs[2] -- 1024 leaf panels 192 x 195, inner-level panels 192 x 82."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from strumpack_amd import _loader  # noqa: E402
from strumpack_amd import hssk as K  # noqa: E402

hk = K.Hssk(_loader.lib_path())
r = np.random.default_rng(0)


def panels(count, d, m):
    k = min(d, m)
    sig = 10.0 ** (-np.arange(k) / 8.0)
    base = []
    for _ in range(4):
        U, _ = np.linalg.qr(r.standard_normal((d, k)))
        V, _ = np.linalg.qr(r.standard_normal((m, k)))
        base.append((U * sig) @ V.T)
    W = np.concatenate([base[i % 4] for i in range(count)], axis=1)   # d x (count m)
    return W


for (count, d, m) in [(1024, 192, 195), (256, 192, 195), (512, 192, 82), (64, 192, 82), (8, 192, 82), (256, 192, 128), (8, 192, 40)]:
    W = panels(count, d, m)
    res = []
    for mr in (4, 12, 20, 28, 36):
        dW = hk.array(W)
        dperm, drank, dwork = hk.empty((count * m,), np.int32), hk.empty((count,), np.int32), hk.empty((3 * m * count,))
        descs = [K.IdDesc(dW.ptr + 8 * d * m * i, d, d, m, 10.0 ** (-mr / 8.0), 0.0, 10000, dperm.ptr + 4 * m * i, drank.ptr + 4 * i, dwork.ptr + 24 * m * i)
                 for i in range(count)]
        best = 1e9
        for rep in range(3):
            dW.set(W)
            hk.sync()
            t0 = time.perf_counter()
            hk.batch("hssk_id_vbatched", descs)
            hk.sync()
            best = min(best, time.perf_counter() - t0)
        res.append(best * 1e6)
        dW.free(); dperm.free(); drank.free(); dwork.free()
    slope = (res[-1] - res[0]) / 32.0
    print("count %4d  %3d x %3d: us at rank 4/12/20/28/36 = %s   per step %.2f us, intercept %.1f us" %
          (count, d, m, " ".join("%.0f" % x for x in res), slope, res[0] - 4 * slope), flush=True)
