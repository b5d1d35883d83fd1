"""Leaf-level batched sample update (the bulk of the batched-GEMM phase): 2 x 512 problems of
C(192 x b) -= A(192 x b) D^T(b x b), b = 195/196, as in BASELINE configs[2] (N = 1e5, leaf 256)."""
import sys
import time

import numpy as np

import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from strumpack_amd import _loader  # noqa: E402
from strumpack_amd import hssk as K  # noqa: E402

hk = K.Hssk(_loader.lib_path())
nleaf, d = 512, 192
sizes = [195 + (i % 2) for i in range(nleaf)]
N = sum(sizes)
Rt = hk.empty((d, N)); St = hk.empty((d, N))
hk.check(hk.lib.hssk_randn(hk.ctx, Rt.ptr, d, N, d, 0, N, 1))
hk.check(hk.lib.hssk_randn(hk.ctx, St.ptr, d, N, d, 0, N, 2))
Ds = [hk.empty((b, b)) for b in sizes]
for i, D in enumerate(Ds):
    hk.check(hk.lib.hssk_randn(hk.ctx, D.ptr, sizes[i], sizes[i], sizes[i], 0, sizes[i], 3 + i))
descs = []
lo = 0
for i, b in enumerate(sizes):
    for tb in (1, 0):
        descs.append(K.GemmDesc(Rt.ptr + 8 * d * lo, Ds[i].ptr, St.ptr + 8 * d * lo, d, b, b, d, b, d, 0, tb, -1.0, 1.0))
    lo += b
flops = sum(2.0 * d * b * b * 2 for b in sizes)
for rep in range(5):
    hk.sync()
    t0 = time.perf_counter()
    hk.batch("hssk_gemm_vbatched", descs)
    hk.sync()
    dt = time.perf_counter() - t0
    print("batched sample update: %.1f us  %.2f TFLOP/s (%.1f%% of 78.6)" % (dt * 1e6, flops / dt * 1e-12, flops / dt * 1e-12 / 78.6 * 100), flush=True)

# fused variant: Sr and Sc updated in one pass over the shared R panel
Sc = hk.empty((d, N))
hk.check(hk.lib.hssk_randn(hk.ctx, Sc.ptr, d, N, d, 0, N, 9))
lu = []
lo = 0
for i, b in enumerate(sizes):
    lu.append(K.LeafUpdateDesc(Rt.ptr + 8 * d * lo, Ds[i].ptr, St.ptr + 8 * d * lo, Sc.ptr + 8 * d * lo, d, b, d, b, d))
    lo += b
for rep in range(5):
    hk.sync()
    t0 = time.perf_counter()
    hk.batch("hssk_leaf_update_vbatched", lu)
    hk.sync()
    dt = time.perf_counter() - t0
    print("fused leaf update:     %.1f us  %.2f TFLOP/s (%.1f%% of 78.6)" % (dt * 1e6, flops / dt * 1e-12, flops / dt * 1e-12 / 78.6 * 100), flush=True)
