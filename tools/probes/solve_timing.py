import sys, time, os
sys.path.insert(0, "/root/repo")
import torch
from strumpack_amd import _loader, capi, hssk as K, dist as sdist
L = capi.load(_loader.lib_path()); hk = K.Hssk(_loader.lib_path())
n = 100000
dA = hk.empty((n, n)); hk.check(hk.lib.hssk_fill_toeplitz(hk.ctx, dA.ptr, n, n, b"T"))
dB = hk.empty((n, 1)); dX = hk.empty((n, 1)); dY = hk.empty((n, 1))
hk.check(hk.lib.hssk_randn(hk.ctx, dB.ptr, n, 1, n, 0, 1, 7)); hk.sync()
opts = capi.StructuredMatrix.options(L, rel_tol=1e-4, abs_tol=1e-8, leaf_size=256, max_rank=50000)
hopts = capi.StructuredMatrix.hss_options(L, random_engine="philox")
H = sdist.from_dense_device(L, dA.ptr, n, n, opts, hopts, None); H.factor()
for name, buf in (("dX", dX), ("dY", dY)):
    for it in range(4):
        hk.check(hk.lib.hssk_memcpy_d2d(hk.ctx, buf.ptr, dB.ptr, 8 * n)); hk.sync(); torch.cuda.synchronize()
        t0 = time.perf_counter(); H.solve_device(buf.ptr, 1); torch.cuda.synchronize(); t1 = time.perf_counter()
        print(name, it, "solve ms", (t1 - t0) * 1e3, "stat", H.stats()["t_solve"] * 1e3, flush=True)
for it in range(4):
    t0 = time.perf_counter(); H.solve_device(dY.ptr, 1); torch.cuda.synchronize(); t1 = time.perf_counter()
    print("repeat", it, "solve ms", (t1 - t0) * 1e3, "stat", H.stats()["t_solve"] * 1e3, flush=True)
for it in range(3):
    t0 = time.perf_counter(); H.mult_device(dB.ptr, dY.ptr, 1); torch.cuda.synchronize(); t1 = time.perf_counter()
    print("mult", it, (t1 - t0) * 1e3, flush=True)
    t0 = time.perf_counter(); H.solve_device(dY.ptr, 1); torch.cuda.synchronize(); t1 = time.perf_counter()
    print("solve after mult", it, (t1 - t0) * 1e3, "stat", H.stats()["t_solve"] * 1e3, flush=True)
import numpy as np
E = np.zeros((n, 64)); E[np.arange(64) * 100, np.arange(64)] = 1.0
Y = H.mult(E)
for it in range(3):
    t0 = time.perf_counter(); H.solve_device(dY.ptr, 1); torch.cuda.synchronize(); t1 = time.perf_counter()
    print("solve after host mult(64)", it, (t1 - t0) * 1e3, "stat", H.stats()["t_solve"] * 1e3, flush=True)
x = H.solve(np.ones((n, 1)))
for it in range(3):
    t0 = time.perf_counter(); H.solve_device(dY.ptr, 1); torch.cuda.synchronize(); t1 = time.perf_counter()
    print("solve after host solve", it, (t1 - t0) * 1e3, "stat", H.stats()["t_solve"] * 1e3, flush=True)
for cyc in range(3):
    H.destroy()
    H = sdist.from_dense_device(L, dA.ptr, n, n, opts, hopts, None); H.factor()
    hk.check(hk.lib.hssk_memcpy_d2d(hk.ctx, dX.ptr, dB.ptr, 8 * n)); hk.sync()
    H.solve_device(dX.ptr, 1)
    print("cycle", cyc, "stat solve", H.stats()["t_solve"] * 1e3, flush=True)
Xh = dX.get(); HX = H.mult(Xh); Y = H.mult(E)
for it in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); H.solve_device(dY.ptr, 1); torch.cuda.synchronize(); t1 = time.perf_counter()
    print("solve after cycles", it, (t1 - t0) * 1e3, "stat", H.stats()["t_solve"] * 1e3, flush=True)
for trial in range(3):
    hk.check(hk.lib.hssk_memcpy_d2d(hk.ctx, dY.ptr, dB.ptr, 8 * n)); hk.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(10):
        H.solve_device(dY.ptr, 1)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print("10 back-to-back solves: ms each", (t1 - t0) * 100, "stat last", H.stats()["t_solve"] * 1e3, flush=True)
    t0 = time.perf_counter()
    for it in range(10):
        H.mult_device(dB.ptr, dY.ptr, 1)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print("10 back-to-back mults: ms each", (t1 - t0) * 100, flush=True)
    t0 = time.perf_counter()
    for it in range(10):
        H.mult_device(dB.ptr, dY.ptr, 1)
        H.solve_device(dY.ptr, 1)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print("10 x (mult + solve): ms each pair", (t1 - t0) * 100, flush=True)
