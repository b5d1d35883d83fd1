"""Device-clock timing of the apply / solve sweeps alone (N = 1e5 Toeplitz, generated operand: no 80 GB fill); the results go
to gpurun_out/sweep_ab_{yN,yT,x}_<SWEEP_AB_TAG>.npy for comparisons between builds / environment switches.
usage: python tools/sweep_ab.py [n] [leaf] [nrhs]   (HSSK_* / STRUMPACK_AMD_* environment switches apply)"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from strumpack_amd import _loader, capi, dist as sdist
    from strumpack_amd import hssk as K
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    leaf = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    nrhs = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    lib = os.environ.get("SWEEP_AB_LIB", _loader.lib_path())
    L = capi.load(lib)
    hk = K.Hssk(lib, device=0)
    opts = capi.StructuredMatrix.options(L, rel_tol=1e-4, abs_tol=1e-8, leaf_size=leaf, max_rank=50000)
    hopts = capi.StructuredMatrix.hss_options(L, random_engine="philox", sketch="gaussian", factor_ahead=False, symmetric=0)
    H = sdist.from_generator(L, n, 1, opts, hopts, comm=None, exchange_cb=None)
    H.factor()
    dB, dY, dS = hk.empty((n, nrhs)), hk.empty((n, nrhs)), hk.empty((n, nrhs))
    hk.check(hk.lib.hssk_randn(hk.ctx, dB.ptr, n, nrhs, n, 0, nrhs, 7))
    hk.sync()
    mctx = L.SPX_d_struct_hssk_ctx(H.h)
    hk.lib.hssk_watch_read_ms.restype = C.c_double
    hk.lib.hssk_watch_read_ms.argtypes = [C.c_void_p, C.c_int, C.c_void_p]

    def dev_ms(fn, reps=20):
        vals = []
        for _ in range(reps):
            hk.sync()
            hk.lib.hssk_watch_start(C.c_void_p(mctx), 6)
            fn()
            hk.lib.hssk_watch_stop(C.c_void_p(mctx), 6)
            vals.append(hk.lib.hssk_watch_read_ms(C.c_void_p(mctx), 6, None))
        vals.sort()
        return vals[len(vals) // 2], vals[0]

    st = H.stats()
    for trans in ("N", "T"):
        a_med, a_min = dev_ms(lambda: H.mult_device(dB.ptr, dY.ptr, nrhs, trans))
        by = st["b_mult"] + 16.0 * n * nrhs if "b_mult" in st else 0
        st = H.stats()
        by = st["b_mult"] + 16.0 * n * nrhs
        print("apply %s: median %.4f ms (min %.4f)  %.0f GB/s" % (trans, a_med, a_min, by / a_med * 1e-6))
        y = dY.get().copy()
        if nrhs == 1: np.save(os.path.join(ROOT, "gpurun_out", "sweep_ab_y%s_%s.npy" % (trans, os.environ.get("SWEEP_AB_TAG", "x"))), y)

    def one_solve():
        hk.check(hk.lib.hssk_memcpy_d2d(hk.ctx, dS.ptr, dB.ptr, 8 * n * nrhs))
        H.solve_device(dS.ptr, nrhs)

    def only_solve():
        H.solve_device(dS.ptr, nrhs)
    one_solve()
    x = dS.get().copy()
    if nrhs == 1: np.save(os.path.join(ROOT, "gpurun_out", "sweep_ab_x_%s.npy" % os.environ.get("SWEEP_AB_TAG", "x")), x)
    s_med, s_min = dev_ms(only_solve)
    st = H.stats()
    by = st["b_solve"] + 16.0 * n * nrhs
    print("solve  : median %.4f ms (min %.4f)  %.0f GB/s" % (s_med, s_min, by / s_med * 1e-6))
    r = H.mult(x) - dB.get()
    print("residual |H x - b| / |b| = %.2e" % (np.linalg.norm(r) / np.linalg.norm(dB.get())))
    H.destroy()


if __name__ == "__main__":
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    os.environ.setdefault("SWEEP_AB_TAG", "x")
    main()
