#!/usr/bin/env python
"""Fixtures of the kernel-matrix front end, produced by the REFERENCE itself (oracle/_ref, built
from /root/reference by oracle/ref/Makefile) -- run in the build container only:

    LD_LIBRARY_PATH=/usr/lib/x86_64-linux-gnu:/opt/conda/lib MKL_THREADING_LAYER=GNU python tests/golden/make_golden_kernel.py

Inputs: tests/golden/data/susy_10Kn_*.csv (the data files the reference ships with examples/dense/KernelRegression.cpp).
Outputs: tests/golden/kernel_golden.json (+ .npz for the larger arrays).
"""
import ctypes as C
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import ref_lib as R  # noqa: E402

L = R.lib()
vp = C.c_void_p
L.ref_kernel_regression.argtypes = [C.c_int, C.c_int, vp, vp, C.c_int, vp, C.c_int, C.c_double, C.c_double, C.c_int,
                                    C.c_double, C.c_double, C.c_int, C.c_int, C.c_int, vp, vp, vp]
L.ref_kernel_hss_create.restype = vp
L.ref_kernel_hss_create.argtypes = [C.c_int, C.c_int, vp, C.c_int, C.c_double, C.c_double, C.c_int, C.c_double, C.c_double,
                                    C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
L.ref_kernel_hss_destroy.argtypes = [vp]
L.ref_kernel_hss_info.argtypes = [vp, vp]
L.ref_kernel_hss_node_info.argtypes = [vp, vp, C.c_int]
L.ref_kernel_hss_data.argtypes = [vp, vp, vp]
L.ref_clustering.argtypes = [C.c_int, C.c_int, vp, C.c_int, C.c_int, vp, vp, C.c_int]
L.ref_ann.argtypes = [C.c_int, C.c_int, vp, C.c_int, C.c_int, vp, vp]


def load(name):
    return np.loadtxt(os.path.join(HERE, "data", "susy_10Kn_" + name + ".csv"), delimiter=",")


train, test = load("train"), load("test")
ytrain, ytest = load("train_label").ravel(), load("test_label").ravel()
out, arrays = {}, {}

# ---- clustering: permutation and leaf sizes per algorithm (full 10K set, leaf 512, and a 1000-point subset, leaf 64)
for tag, X, leaf in (("full", train, 512), ("sub1000", train[:1000], 64)):
    for algo, name in ((1, "2means"), (2, "kdtree"), (4, "cobble"), (0, "natural"), (3, "pca")):
        a = np.ascontiguousarray(X).copy()
        n, d = a.shape
        perm = np.zeros(n, np.int32)
        ls = np.zeros(4096, np.int32)
        c = L.ref_clustering(n, d, a.ctypes.data, algo, leaf, perm.ctypes.data, ls.ctypes.data, 4096)
        out["clustering_%s_%s" % (tag, name)] = dict(n=n, leaf=leaf, leaves=ls[:c].tolist())
        arrays["perm_%s_%s" % (tag, name)] = perm


def regression(tag, n, m, ktype, h, lam, p, rtol, leaf, clustering, ann):
    X, y, T = np.ascontiguousarray(train[:n]), np.ascontiguousarray(ytrain[:n]), np.ascontiguousarray(test[:m])
    d = X.shape[1]
    w, pr = np.zeros(n), np.zeros(m)
    info = (C.c_longlong * 4)()
    L.ref_kernel_regression(n, d, X.ctypes.data, y.ctypes.data, m, T.ctypes.data, ktype, h, lam, p, rtol, 1e-8, leaf,
                            clustering, ann, w.ctypes.data, pr.ctypes.data, C.addressof(info))
    H = L.ref_kernel_hss_create(n, d, X.ctypes.data, ktype, h, lam, p, rtol, 1e-8, leaf, 50000, clustering, ann, 5)
    hi = (C.c_longlong * 4)()
    L.ref_kernel_hss_info(H, hi)
    ni = np.zeros((1 << 14, 6), np.int32)
    cnt = L.ref_kernel_hss_node_info(H, ni.ctypes.data, 1 << 14)
    # the reference's own neighbour lists on the clustered points (deterministic: fixed seed in NeighborSearch.cpp)
    data = np.zeros_like(X)
    perm = np.zeros(n, np.int32)
    L.ref_kernel_hss_data(H, data.ctypes.data, perm.ctypes.data)
    L.ref_kernel_hss_destroy(H)
    # compress_with_coordinates doubles the neighbour count until the tree compresses (compress_kernel.hpp:56-76) and
    # recomputes everything each round: find the round that succeeded by replaying the rounds with the numpy
    # restatement on the reference's own lists, and keep that round's lists
    from oracle import hss_oracle as O
    k = min(n, ann)
    while True:
        annl = np.zeros((n, k), np.uint32)
        sc = np.zeros((n, k))
        L.ref_ann(n, d, data.ctypes.data, 5, k, annl.ctypes.data, sc.ctypes.data)
        Ho = O.HSSMatrix.from_kernel(data, O.kernel_function(ktype, h, p), lam, ni[:cnt, 1], ni[:cnt, 5], annl.astype(np.int64),
                                     O.Options(rel_tol=rtol, abs_tol=1e-8, leaf_size=leaf))
        if Ho.is_compressed() or k >= n:
            break
        k = min(2 * k, n)
    oracle_ranks = [nd.rU for nd in Ho.nodes]
    print(tag, "final neighbour count", k, "oracle ranks == reference ranks:", oracle_ranks == ni[:cnt, 3].tolist(), flush=True)
    acc = float(np.mean((pr >= 0) == (ytest[:m] >= 0)))
    out["regression_" + tag] = dict(n=n, m=m, ktype=ktype, h=h, lam=lam, p=p, rel_tol=rtol, leaf=leaf, clustering=clustering,
                                    ann=ann, ann_final=int(k), accuracy=acc, compressed=int(hi[0]), levels=int(hi[1]), rank=int(hi[2]),
                                    memory=int(hi[3]), nodes=ni[:cnt].tolist())
    arrays["weights_" + tag] = w
    arrays["prediction_" + tag] = pr
    arrays["perm_" + tag] = perm
    arrays["ann_" + tag] = annl.astype(np.int32)
    print(tag, "accuracy", acc, "rank", hi[2], "levels", hi[1], "memory MB", hi[3] / 1e6, flush=True)


regression("gauss_400", 400, 200, 0, 1.3, 3.11, 1, 1e-2, 64, 1, 64)
regression("gauss_1500", 1500, 300, 0, 1.3, 3.11, 1, 1e-2, 128, 1, 64)
regression("laplace_400", 400, 200, 1, 1.3, 3.11, 1, 1e-2, 64, 1, 64)
regression("anova_400", 400, 200, 2, 1.3, 3.11, 2, 1e-2, 64, 2, 64)
regression("gauss_10k", 10000, 1000, 0, 1.3, 3.11, 1, 1e-2, 512, 1, 64)   # examples/dense/KernelRegression.cpp defaults

json.dump(out, open(os.path.join(HERE, "kernel_golden.json"), "w"), indent=0)
np.savez_compressed(os.path.join(HERE, "kernel_golden.npz"), **arrays)
print("written", len(out), "entries")
