"""Quick on-box performance probe of the dominant kernels (not a test)."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from strumpack_amd import _loader  # noqa: E402
from strumpack_amd import hssk as K  # noqa: E402

hk = K.Hssk(_loader.lib_path())
out = {}
out["mfma_f64_peak_tflops"] = hk.lib.hssk_mfma_f64_peak_tflops(hk.ctx, 40000)
print("mfma peak", out["mfma_f64_peak_tflops"], flush=True)
for (n, d) in [(8192, 192), (32768, 192), (65536, 192), (100000, 192)]:
    dA = hk.empty((n, n))
    hk.check(hk.lib.hssk_fill_toeplitz(hk.ctx, dA.ptr, n, n, b"T"))
    dR = hk.empty((d, n))
    hk.check(hk.lib.hssk_randn(hk.ctx, dR.ptr, d, n, d, 0, n, 1))
    dS = hk.empty((d, n))
    for tb in (1, 0):
        for rep in range(3):
            hk.check(hk.lib.hssk_dgemm(hk.ctx, tb, d, n, n, 1.0, dR.ptr, d, dA.ptr, n, 0.0, dS.ptr, d))
            hk.sync()
            ms = hk.lib.hssk_last_dgemm_ms(hk.ctx)
        t0 = time.time()
        hk.check(hk.lib.hssk_dgemm(hk.ctx, tb, d, n, n, 1.0, dR.ptr, d, dA.ptr, n, 0.0, dS.ptr, d))
        hk.sync()
        wall = time.time() - t0
        ms = hk.lib.hssk_last_dgemm_ms(hk.ctx)
        tf = hk.lib.hssk_last_dgemm_flops(hk.ctx) / (ms * 1e-3) * 1e-12   # main launch only
        out_wall_tf = 2.0 * d * n * n / wall * 1e-12
        out[f"dgemm_n{n}_d{d}_tb{tb}"] = dict(ms=ms, tflops=tf, wall_ms=wall * 1e3)
        ghz = hk.lib.hssk_last_dgemm_clock_ghz(hk.ctx)
        out[f"dgemm_n{n}_d{d}_tb{tb}"]["clock_ghz"] = ghz
        print(n, d, tb, "main kernel ms %.3f  TF/s %.2f  wall ms %.3f (%.2f TF/s whole call)  clock %.3f GHz" % (ms, tf, wall * 1e3, out_wall_tf, ghz), flush=True)
    dA.free(); dR.free(); dS.free()
json.dump(out, open("gpurun_out/perf_kernels.json", "w"), indent=1)
