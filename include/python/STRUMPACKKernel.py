"""STRUMPACKKernel: the scikit-learn style kernel ridge regression classifier of the reference's Python interface
(src/python/STRUMPACKKernel.py.in, installed as include/python/STRUMPACKKernel.py), on this library.

    import STRUMPACKKernel as sp                      # with <repo>/include/python on PYTHONPATH
    K = sp.STRUMPACKKernel(h, lam, degree, kernel='rbf', approximation='HSS', argv=sys.argv)
    K.fit(train_points, train_labels); pred = K.predict(test_points)

Same constructor arguments, same methods (fit / predict / decision_function), same C entry points underneath
(STRUMPACK_create_kernel_double, STRUMPACK_kernel_fit_HSS_double, STRUMPACK_kernel_predict_double: include/kernel/Kernel.h),
so the reference's examples/dense/KernelRegression.py runs unchanged.  float32 inputs are carried in double precision (the
engine's arithmetic); the MPI / HODLR variants of the reference's class are not part of this library.
The library is strumpack_amd/lib/libstrumpack_amd.so (hipcc, gfx950): there is no other route -- without it, or without a
HIP device, fit() raises."""
import ctypes
import os
import sys

import numpy as np
from sklearn.base import BaseEstimator, ClassifierMixin
from sklearn.utils.multiclass import unique_labels
from sklearn.utils.validation import check_is_fitted, check_X_y

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
from strumpack_amd import _loader  # noqa: E402
from strumpack_amd import kernel as _kernel  # noqa: E402

_lib = None


def _library():
    global _lib
    if _lib is None:
        _lib = _kernel.load(_loader.lib_path())
    return _lib


class STRUMPACKKernel(BaseEstimator, ClassifierMixin):
    """kernel: 'rbf' / 'Gauss', 'Laplace' or 'ANOVA'; approximation: 'HSS'"""

    def __init__(self, h=1., lam=4., degree=1, kernel='rbf', approximation='HSS', mpi=False, argv=None):
        self.h = h
        self.lam = lam
        self.degree = int(degree)
        self.kernel = kernel
        self.approximation = approximation
        self.mpi = mpi
        self.argv = argv

    def __del__(self):
        K = getattr(self, "K_", None)
        if K:
            try:
                _library().STRUMPACK_destroy_kernel_double(K)
            except Exception:
                pass

    def fit(self, X, y):
        if X.dtype != np.float32 and X.dtype != np.float64:
            raise ValueError("precision", X.dtype, "not supported")
        if self.kernel not in _kernel.KERNEL_TYPES:
            raise ValueError("Kernel type", self.kernel, "not recognized")
        if self.approximation != 'HSS':
            raise ValueError("Approximation type not available, should be 'HSS' (HODLR needs the reference's MPI build)")
        if self.mpi:
            raise ValueError("mpi=True: one process per GPU shares a matrix through strumpack_amd.dist, not through this class")
        X, y = check_X_y(X, y)
        self.classes_ = unique_labels(y)
        L = _library()
        Xd = np.ascontiguousarray(X, dtype=np.float64)   # n x d row-major == d x n column-major: one point per column
        yd = np.ascontiguousarray(y, dtype=np.float64)
        old = getattr(self, "K_", None)
        if old:
            L.STRUMPACK_destroy_kernel_double(old)
        self.K_ = L.STRUMPACK_create_kernel_double(Xd.shape[0], Xd.shape[1], Xd.ctypes.data, float(self.h), float(self.lam),
                                                   int(self.degree), _kernel.KERNEL_TYPES[self.kernel])
        if not self.K_:
            raise RuntimeError("STRUMPACK_create_kernel_double failed")
        args = [str(a).encode("utf-8") for a in (self.argv or [])]
        argv = (ctypes.c_char_p * (len(args) + 1))(*args, None)
        L.STRUMPACK_kernel_fit_HSS_double(self.K_, yd.ctypes.data, len(args), argv)
        return self

    def decision_function(self, X):
        check_is_fitted(self, 'K_')
        Xd = np.ascontiguousarray(X, dtype=np.float64)
        prediction = np.zeros((Xd.shape[0], 1), dtype=np.float64)
        _library().STRUMPACK_kernel_predict_double(self.K_, Xd.shape[0], Xd.ctypes.data, prediction.ctypes.data)
        return prediction.astype(X.dtype, copy=False)

    def predict(self, X):
        prediction = self.decision_function(X)
        return [self.classes_[0] if prediction[i] < 0.0 else self.classes_[1] for i in range(prediction.shape[0])]
