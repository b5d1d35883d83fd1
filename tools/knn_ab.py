"""Exact k-nearest-neighbour search at BASELINE configs[3]'s size (N = 1e5 points in R^8, k = 64, points in cluster order):
the filtered search (bound over a window -> FP32 MFMA filter -> exact selection) against the heap kernel (HSSK_KNN_FILTER=0 in
a second process), same sets.   usage (GPU box):  python tools/knn_ab.py [n] [d] [k]"""
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    d = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    k = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    from strumpack_amd import _loader, hssk as K, kernel as KM
    lib = KM.load(_loader.lib_path())
    X = np.random.default_rng(2025).random((n, d))
    Xp, perm, leaves = KM.clustering(lib, X, "cobble", 256)
    hk = K.Hssk(_loader.lib_path())
    dX = hk.array(Xp.T)
    out = hk.empty((k, n), dtype=np.int32)
    ts = []
    for rep in range(5):
        hk.sync()
        t0 = time.perf_counter()
        hk.check(hk.lib.hssk_knn(hk.ctx, dX.ptr, d, n, k, 0, n, out.ptr))
        hk.sync()
        ts.append((time.perf_counter() - t0) * 1e3)
    got = np.sort(out.get().T, axis=1)
    mode = "heap" if os.environ.get("HSSK_KNN_FILTER") == "0" else "filtered"
    print("%s: n=%d d=%d k=%d  ms per search: %s" % (mode, n, d, k, " ".join("%.2f" % t for t in ts)))
    if mode == "filtered":
        np.save("/tmp/knn_filtered.npy", got)
        env = dict(os.environ, HSSK_KNN_FILTER="0")
        subprocess.run([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, check=True)
        ref = np.load("/tmp/knn_heap.npy")
        same = (got == ref).all(axis=1)
        print("rows with identical sets: %d of %d" % (int(same.sum()), n))
        if not same.all():
            i = int(np.nonzero(~same)[0][0])
            print("first difference: query", i, sorted(set(got[i]) ^ set(ref[i])))
    else:
        np.save("/tmp/knn_heap.npy", got)
    hk.close()


if __name__ == "__main__":
    main()
