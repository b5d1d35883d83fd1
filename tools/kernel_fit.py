"""Kernel ridge regression fit + predict through the C-ABI on the susy_10Kn data set (examples/dense/KernelRegression.cpp
defaults) or on synthetic points: timing of the kernel-matrix front end."""
import os
import sys
import time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
from strumpack_amd import _loader
from strumpack_amd import kernel as KM

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
leaf = sys.argv[2] if len(sys.argv) > 2 else "512"
L = KM.load(_loader.lib_path())
if n <= 10000:
    import kernel_golden as KG
    X, y, T, yt = KG.susy()
    X, y = X[:n], y[:n]
else:
    r = np.random.default_rng(2025)
    X = r.random((n, 8)); T = r.random((1000, 8))
    wtrue = r.standard_normal(8)
    y = np.sign((X - 0.5) @ wtrue); yt = np.sign((T - 0.5) @ wtrue)
for rep in range(2):
    t0 = time.time()
    kr = KM.KernelRegression(L, h=1.3, lam=3.11, argv=["--hss_leaf_size", leaf] + sys.argv[3:]).fit(X, y)
    t1 = time.time()
    pred = kr.decision_function(T)
    t2 = time.time()
    print("n=%d leaf=%s fit %.3f s predict %.3f s accuracy %.3f" % (n, leaf, t1 - t0, t2 - t1, np.mean((pred >= 0) == (yt >= 0))), kr.info(), flush=True)
    kr.destroy()
