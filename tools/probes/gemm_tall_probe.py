"""Times one batched launch of the leaf-level shapes of a 64-right-hand-side mat-vec (512 problems 195 x 64 x 195, and
391 x 256 x 64 x 256) through hssk_gemm_vbatched, tall kernel on / off (HSSK_GEMM_NO_TALL): python tools/probes/gemm_tall_probe.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from strumpack_amd import hssk as K, _loader

hk = K.Hssk(_loader.lib_path())
for (cnt, m, n, k) in [(512, 195, 64, 195), (391, 256, 64, 256), (512, 195, 64, 154)]:
    r = np.random.default_rng(0)
    A = hk.array(r.standard_normal((k * cnt, m)).T.copy(order="F").reshape(m, k * cnt, order="F"))   # cnt blocks m x k side by side
    B = hk.array(np.asfortranarray(r.standard_normal((k, n * cnt))))
    C = hk.empty((m, n * cnt))
    descs = [K.GemmDesc(A.ptr + 8 * m * k * p, B.ptr + 8 * k * n * p, C.ptr + 8 * m * n * p, m, n, k, m, k, m, 0, 0, 1.0, 0.0) for p in range(cnt)]
    for it in range(3):
        hk.sync(); t0 = time.perf_counter()
        for rep in range(10):
            hk.batch("hssk_gemm_vbatched", descs)
        hk.sync(); dt = (time.perf_counter() - t0) / 10
    fl = 2.0 * m * n * k * cnt
    print("tall_off" if os.environ.get("HSSK_GEMM_NO_TALL") == "1" else "tall_on ", (cnt, m, n, k), "%.1f us  %.1f TFLOP/s  %.2f TB/s" % (dt * 1e6, fl / dt / 1e12, 8.0 * cnt * (m * k + k * n + m * n) / dt / 1e12))
