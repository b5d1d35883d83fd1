import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, time
from strumpack_amd import _loader
from strumpack_amd import kernel as KM
lib = KM.load(_loader.lib_path())
r = np.random.default_rng(2025)
X = r.random((100000, 8))
for algo in ("cobble","kdtree"):
    for rep in range(3):
        t0=time.perf_counter(); st, Xd, pd, ld = KM.clustering_device(lib, X, algo, 256); t1=time.perf_counter()
        Xh, ph, lh = KM.clustering(lib, X, algo, 256); t2=time.perf_counter()
        print(algo, "status", st, "device %.2f ms host %.2f ms"%((t1-t0)*1e3,(t2-t1)*1e3), "equal", np.array_equal(pd,ph))
