"""ctypes binding of the kernel ridge regression C interface (include/kernel/Kernel.h), the same entry points the
reference's src/python/STRUMPACKKernel.py.in binds: STRUMPACK_create_kernel_double / _kernel_fit_HSS_double /
_kernel_predict_double / _destroy_kernel_double."""
import ctypes as C

import numpy as np

KERNEL_SYMBOLS = [
    "STRUMPACK_create_kernel_double", "STRUMPACK_destroy_kernel_double", "STRUMPACK_kernel_fit_HSS_double",
    "STRUMPACK_kernel_predict_double", "SPX_kernel_fit_info", "SPX_kernel_permutation", "SPX_kernel_weights",
    "SPX_clustering", "SPX_clustering_device", "SPX_kernel_node_info", "SPX_kernel_set_neighbors", "SPX_approximate_neighbors",
]
KERNEL_TYPES = {"Gauss": 0, "rbf": 0, "Laplace": 1, "ANOVA": 2}
CLUSTERING = {"natural": 0, "2means": 1, "kdtree": 2, "pca": 3, "cobble": 4}


def load(path):
    import torch  # noqa: F401  (first: its bundled HIP runtime must be the one in the process)
    L = C.CDLL(path)
    vp = C.c_void_p
    L.STRUMPACK_create_kernel_double.restype = vp
    L.STRUMPACK_create_kernel_double.argtypes = [C.c_int, C.c_int, vp, C.c_double, C.c_double, C.c_int, C.c_int]
    L.STRUMPACK_destroy_kernel_double.argtypes = [vp]
    L.STRUMPACK_kernel_fit_HSS_double.argtypes = [vp, vp, C.c_int, C.POINTER(C.c_char_p)]
    L.STRUMPACK_kernel_predict_double.argtypes = [vp, C.c_int, vp, vp]
    L.SPX_kernel_fit_info.argtypes = [vp, vp]
    L.SPX_kernel_permutation.argtypes = [vp, vp]
    L.SPX_kernel_weights.argtypes = [vp, vp]
    L.SPX_kernel_node_info.argtypes = [vp, vp, C.c_int]
    L.SPX_kernel_set_neighbors.argtypes = [vp, C.c_int, vp]
    L.SPX_approximate_neighbors.argtypes = [C.c_int, C.c_int, vp, C.c_int, C.c_int, vp, vp]
    L.SPX_clustering.argtypes = [C.c_int, C.c_int, vp, C.c_int, C.c_int, vp, vp, C.c_int]
    L.SPX_clustering_device.argtypes = [C.c_int, C.c_int, vp, C.c_int, C.c_int, vp, vp, C.c_int, vp]
    return L


class KernelRegression:
    """Kernel ridge regression classifier, same shape as the reference's STRUMPACKKernel (fit / predict)."""

    def __init__(self, lib, h=1.0, lam=4.0, kernel="rbf", degree=1, argv=()):
        self.L, self.h, self.lam, self.ktype, self.p, self.argv = lib, h, lam, KERNEL_TYPES[kernel], degree, list(argv)
        self.K = None

    def fit(self, X, y, neighbors=None):
        X = np.ascontiguousarray(X, dtype=np.float64)     # n x d row-major == d x n column-major
        y = np.ascontiguousarray(y, dtype=np.float64)
        self.n, self.d = X.shape
        self.destroy()
        self.K = self.L.STRUMPACK_create_kernel_double(self.n, self.d, X.ctypes.data, self.h, self.lam, self.p, self.ktype)
        if not self.K:
            raise RuntimeError("STRUMPACK_create_kernel_double failed")
        if neighbors is not None:   # tests: k x n lists in cluster order (see include/kernel/Kernel.h)
            nb = np.ascontiguousarray(neighbors, dtype=np.int32)
            self.L.SPX_kernel_set_neighbors(self.K, nb.shape[1], nb.ctypes.data)
        args = [b"kernel"] + [a.encode() for a in self.argv]
        argv = (C.c_char_p * len(args))(*args)
        self.L.STRUMPACK_kernel_fit_HSS_double(self.K, y.ctypes.data, len(args), argv)
        return self

    def decision_function(self, T):
        T = np.ascontiguousarray(T, dtype=np.float64)
        out = np.zeros(T.shape[0])
        self.L.STRUMPACK_kernel_predict_double(self.K, T.shape[0], T.ctypes.data, out.ctypes.data)
        return out

    def predict(self, T):
        return np.where(self.decision_function(T) >= 0, 1.0, -1.0)

    def info(self):
        out = np.zeros(8, dtype=np.int64)
        if self.L.SPX_kernel_fit_info(self.K, out.ctypes.data):
            raise RuntimeError("no fit")
        return dict(zip(["compressed", "levels", "rank", "memory", "neighbors", "compress_us", "factor_us", "solve_us"], out.tolist()))

    def node_info(self):
        out = np.zeros((1 << 16, 6), dtype=np.int32)
        c = self.L.SPX_kernel_node_info(self.K, out.ctypes.data, 1 << 16)
        return out[:c].copy()

    def permutation(self):
        p = np.zeros(self.n, dtype=np.int32)
        self.L.SPX_kernel_permutation(self.K, p.ctypes.data)
        return p

    def weights(self):
        w = np.zeros(self.n)
        if self.L.SPX_kernel_weights(self.K, w.ctypes.data):
            raise RuntimeError("no fit")
        return w

    def destroy(self):
        if getattr(self, "K", None):
            self.L.STRUMPACK_destroy_kernel_double(self.K)
            self.K = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


def clustering(lib, X, algo="2means", leaf_size=512):
    """binary_tree_clustering on its own: returns (reordered points, 1-based permutation, leaf sizes)."""
    X = np.ascontiguousarray(X, dtype=np.float64).copy()
    n, d = X.shape
    perm = np.zeros(n, dtype=np.int32)
    ls = np.zeros(max(16, 4 * n // max(leaf_size, 1) + 16), dtype=np.int32)
    c = lib.SPX_clustering(n, d, X.ctypes.data, CLUSTERING[algo], leaf_size, perm.ctypes.data, ls.ctypes.data, len(ls))
    if c < 0:
        raise RuntimeError("SPX_clustering failed")
    return X, perm, ls[:c].copy()


def clustering_device(lib, X, algo="cobble", leaf_size=512):
    """The median-split partitioners on the device (SPX_clustering_device): returns (status, reordered points, 1-based
    permutation, leaf sizes); status != 0: the device form stood back (ties, long displacement chains) and nothing was moved."""
    X = np.ascontiguousarray(X, dtype=np.float64).copy()
    n, d = X.shape
    perm = np.zeros(n, dtype=np.int32)
    ls = np.zeros(max(16, 4 * n // max(leaf_size, 1) + 16), dtype=np.int32)
    st = C.c_int(0)
    c = lib.SPX_clustering_device(n, d, X.ctypes.data, CLUSTERING[algo], leaf_size, perm.ctypes.data, ls.ctypes.data, len(ls), C.byref(st))
    if c < 0:
        raise RuntimeError("SPX_clustering_device failed")
    return st.value, X, perm, ls[:c].copy()


def approximate_neighbors(lib, X, k, iterations=5):
    """find_approximate_neighbors (the reference's randomized projection-tree search, host): returns ids (n x k)."""
    X = np.ascontiguousarray(X, dtype=np.float64)
    n, d = X.shape
    out = np.zeros((n, k), dtype=np.int32)
    if lib.SPX_approximate_neighbors(n, d, X.ctypes.data, iterations, k, out.ctypes.data, None):
        raise RuntimeError("SPX_approximate_neighbors failed")
    return out
