"""Median-split clustering (cobble, kd) of 1e5 points in R^8, leaf 256: the device form (hssk_cluster_median through
SPX_clustering_device, copies included) against the host form (SPX_clustering) -- times and whether the permutations are equal.
usage (GPU box): python tools/cluster_ab.py   (HSSK_CLUSTER_WIDE_MIN=0: large clusters by one workgroup each)"""
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
