"""A/B of the two forms of the sketch GEMM's interior tiles inside one process (same box, same thermal state):
HSSK_DGEMM_V1=1 -> four-wave register-staged form (rounds 1-3), 0 -> eight-wave LDS-DMA form (round 4).
usage: dgemm_ab.py [n] [variants, e.g. "1:0,0:0,0:5,0:6,0:8"]  (form:split, split 0 = the model's choice)"""
import ctypes as C
import os
import sys
import time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from strumpack_amd import _loader  # noqa: E402
from strumpack_amd import hssk as K  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
d = 192
hk = K.Hssk(_loader.lib_path())
hk.lib.hssk_last_dgemm_clock_ghz.restype = C.c_double
hk.lib.hssk_last_dgemm_clock_ghz.argtypes = [C.c_void_p]
dA = hk.empty((n, n))
hk.check(hk.lib.hssk_fill_toeplitz(hk.ctx, dA.ptr, n, n, b"T"))
dR = hk.empty((d, n))
hk.check(hk.lib.hssk_randn(hk.ctx, dR.ptr, d, n, d, 0, n, 1))
dS = hk.empty((d, n))
variants = (sys.argv[2] if len(sys.argv) > 2 else "1:0,0:0,1:0,0:0,0:5,0:6,0:7,0:8,0:12").split(",")
for v in variants:
    form, sp = v.split(":")
    os.environ["HSSK_DGEMM_V1"] = form
    if int(sp):
        os.environ["HSSK_DGEMM_SPLIT"] = sp
    else:
        os.environ.pop("HSSK_DGEMM_SPLIT", None)
    res = []
    for tb in (1, 0):
        hk.sync()
        t0 = time.perf_counter()
        hk.check(hk.lib.hssk_dgemm(hk.ctx, tb, d, n, n, 1.0, dR.ptr, d, dA.ptr, n, 0.0, dS.ptr, d))
        hk.sync()
        wall = (time.perf_counter() - t0) * 1e3
        ms = hk.lib.hssk_last_dgemm_ms(hk.ctx)
        fl = hk.lib.hssk_last_dgemm_flops(hk.ctx)
        res.append((ms, wall, fl / ms / 1e9 / 78.6, hk.lib.hssk_last_dgemm_clock_ghz(hk.ctx)))
    print("form %s split %2s | T main %.2f ms (call %.2f) frac %.3f clk %.2f | N main %.2f ms (call %.2f) frac %.3f clk %.2f | calls %.2f ms"
          % ("v1" if form == "1" else "v2", sp, res[0][0], res[0][1], res[0][2], res[0][3], res[1][0], res[1][1], res[1][2], res[1][3],
             res[0][1] + res[1][1]), flush=True)
