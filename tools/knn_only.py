"""hssk_knn on its own (PMC passes): n points uniform in [0,1)^8, k = 64."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from strumpack_amd import _loader, hssk as K
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
hk = K.Hssk(_loader.lib_path())
X = np.random.default_rng(1).random((n, 8))
dX = hk.array(X.T)
out = hk.empty((64, n), dtype=np.int32)
for rep in range(2):
    hk.sync(); t0 = time.time()
    hk.check(hk.lib.hssk_knn(hk.ctx, dX.ptr, 8, n, 64, 0, n, out.ptr)); hk.sync()
    print("knn ms", (time.time() - t0) * 1e3, flush=True)
