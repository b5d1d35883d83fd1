"""TEST INFRASTRUCTURE ONLY: ctypes loader for oracle/_ref/libstrumpack_ref.so.

The library is the reference's own CPU HSS code (pghysels/STRUMPACK v8.0.0) compiled by
oracle/ref/Makefile from the sources under /root/reference plus the thin shim
oracle/ref/ref_driver.cpp.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module; the product never does.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "_ref", "libstrumpack_ref.so")
_lib = None


def available():
    return os.path.exists(_PATH)


def lib():
    global _lib
    if _lib is None:
        os.environ.setdefault("MKL_THREADING_LAYER", "GNU")
        _lib = C.CDLL(_PATH)   # RTLD_LOCAL: its C++ symbols share names with the product (same public API)
        L = _lib
        dp = C.POINTER(C.c_double)
        L.ref_hss_create.restype = C.c_void_p
        L.ref_hss_create.argtypes = [C.c_int, dp, C.c_int, C.c_double, C.c_double, C.c_int,
                                     C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.ref_hss_create_sjlt.restype = C.c_void_p
        L.ref_hss_create_sjlt.argtypes = [C.c_int, dp, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int,
                                          C.c_int, C.c_int, C.c_int, C.c_int]
        for f in ("ref_hss_destroy", "ref_hss_factor"):
            getattr(L, f).argtypes = [C.c_void_p]
            getattr(L, f).restype = None
        for f in ("ref_hss_is_compressed", "ref_hss_levels", "ref_hss_rank"):
            getattr(L, f).argtypes = [C.c_void_p]
            getattr(L, f).restype = C.c_int
        for f in ("ref_hss_memory", "ref_hss_nonzeros"):
            getattr(L, f).argtypes = [C.c_void_p]
            getattr(L, f).restype = C.c_longlong
        L.ref_hss_node_info.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_int]
        L.ref_hss_node_info.restype = C.c_int
        L.ref_hss_dense.argtypes = [C.c_void_p, dp, C.c_int]
        L.ref_hss_mult.argtypes = [C.c_void_p, C.c_char, C.c_int, dp, C.c_int, dp, C.c_int]
        L.ref_hss_solve.argtypes = [C.c_void_p, C.c_int, dp, C.c_int]
        L.ref_hss_shift.argtypes = [C.c_void_p, C.c_double]
        L.ref_hss_schur_update.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        L.ref_hss_schur_update.restype = C.c_int
        L.ref_hss_schur_get.argtypes = [C.c_void_p, C.c_int, dp]
        L.ref_hss_schur_update_dense.argtypes = [C.c_void_p, dp]
        L.ref_hss_schur_product_direct.argtypes = [C.c_void_p, C.c_int, dp, dp, dp]
        L.ref_randn.argtypes = [C.c_longlong, dp]
        L.ref_fill_test_matrix.argtypes = [C.c_char, C.c_int, dp]
        L.ref_flops.argtypes = [C.POINTER(C.c_longlong), C.c_int]
        L.ref_hss_bench_toeplitz.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double, C.c_int,
                                             dp, dp]
        L.ref_hss_bench_toeplitz.restype = C.c_int
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def randn(count):
    out = np.empty(count)
    lib().ref_randn(count, _dp(out))
    return out


def test_matrix(kind, n):
    A = np.empty((n, n), order="F")
    lib().ref_fill_test_matrix(kind.encode(), n, _dp(A))
    return A


class RefHSS:
    """The reference's HSSMatrix<double>(A, opts) (HSS/HSSMatrix.cpp:50-54)."""

    def __init__(self, A, rel_tol=1e-2, abs_tol=1e-8, leaf=512, d0=128, dd=64, p=10,
                 max_rank=50000, algo="stable", sjlt=None):
        A = np.asfortranarray(A, dtype=np.float64)
        self.n = A.shape[0]
        if sjlt is not None:    # (algo "chunk" | "perm", nnz0, nnz): --hss_compression_sketch SJLT
            self.h = lib().ref_hss_create_sjlt(self.n, _dp(A), A.shape[0], rel_tol, abs_tol, leaf, d0, dd,
                                               0 if algo == "original" else 1, int(sjlt[0] == "perm"), sjlt[1], sjlt[2])
            return
        self.h = lib().ref_hss_create(self.n, _dp(A), A.shape[0], rel_tol, abs_tol, leaf, d0, dd,
                                      p, max_rank, 0 if algo == "original" else 1)

    @classmethod
    def toeplitz_matfree(cls, n, rel_tol=1e-2, abs_tol=1e-8, leaf=512, d0=128, dd=64, p=10, max_rank=50000, algo="stable"):
        """T(n) of test_HSS_seq.cpp through compress(Amult, Aelem): never stores the n x n matrix (ref_driver.cpp)"""
        L = lib()
        L.ref_hss_create_toeplitz_matfree.restype = C.c_void_p
        L.ref_hss_create_toeplitz_matfree.argtypes = [C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        self = cls.__new__(cls)
        self.n = n
        self.h = L.ref_hss_create_toeplitz_matfree(n, rel_tol, abs_tol, leaf, d0, dd, p, max_rank, 0 if algo == "original" else 1)
        return self

    def __del__(self):
        if getattr(self, "h", None):
            lib().ref_hss_destroy(self.h)
            self.h = None

    def is_compressed(self):
        return bool(lib().ref_hss_is_compressed(self.h))

    def levels(self):
        return lib().ref_hss_levels(self.h)

    def rank(self):
        return lib().ref_hss_rank(self.h)

    def memory(self):
        return lib().ref_hss_memory(self.h)

    def nonzeros(self):
        return lib().ref_hss_nonzeros(self.h)

    def node_info(self):
        """Pre-order rows of (row_offset, rows, U_rows, U_rank, V_rank, is_leaf)."""
        cap = 4 * self.n + 8
        out = np.zeros((cap, 6), dtype=np.int32)
        cnt = lib().ref_hss_node_info(self.h, out.ctypes.data_as(C.POINTER(C.c_int)), cap)
        return out[:cnt].copy()

    def dense(self):
        D = np.empty((self.n, self.n), order="F")
        lib().ref_hss_dense(self.h, _dp(D), self.n)
        return D

    def mult(self, B, trans="N"):
        B = np.asfortranarray(B, dtype=np.float64).reshape(self.n, -1, order="F")
        Cm = np.empty_like(B, order="F")
        lib().ref_hss_mult(self.h, trans.encode(), B.shape[1], _dp(B), self.n, _dp(Cm), self.n)
        return Cm

    def factor(self):
        lib().ref_hss_factor(self.h)

    def solve(self, B):
        X = np.array(B, dtype=np.float64, order="F").reshape(self.n, -1, order="F")
        lib().ref_hss_solve(self.h, X.shape[1], _dp(X), self.n)
        return X

    def shift(self, s):
        lib().ref_hss_shift(self.h, s)

    # ---- Schur complement of the (0,0) block, driven as sparse/fronts/FrontHSS.cpp:391-407 does
    def schur_update(self):
        """partial_factor + Schur_update -> dict(Theta, DUB01, Phi, Vhat) or None for a leaf root."""
        d = (C.c_int * 6)()
        if not lib().ref_hss_schur_update(self.h, d):
            return None
        n1, tc, pc, dc, vr, vc = list(d)
        self._n1 = n1
        out = {}
        for k, (name, shape) in enumerate([("Theta", (n1, tc)), ("DUB01", (pc, dc)), ("Phi", (n1, pc)), ("Vhat", (vr, vc))]):
            M = np.zeros(shape, order="F")
            if M.size:
                lib().ref_hss_schur_get(self.h, k, _dp(M))
            out[name] = M
        return out

    def schur_update_dense(self):
        U = np.zeros((self._n1, self._n1), order="F")
        lib().ref_hss_schur_update_dense(self.h, _dp(U))
        return U

    def schur_product_direct(self, R):
        R = np.asfortranarray(R, dtype=np.float64).reshape(self._n1, -1, order="F")
        Sr, Sc = np.empty_like(R, order="F"), np.empty_like(R, order="F")
        lib().ref_hss_schur_product_direct(self.h, R.shape[1], _dp(R), _dp(Sr), _dp(Sc))
        return Sr, Sc


def flops(reset=False):
    out = (C.c_longlong * 9)()
    lib().ref_flops(out, int(reset))
    names = ["flops", "update_sample", "reduce_sample", "ID", "QR", "ortho", "random",
             "ULV_factor", "hss_solve"]
    return dict(zip(names, list(out)))


def bench_toeplitz(n, leaf=256, rel_tol=1e-4, abs_tol=1e-8, nrhs=1):
    times = np.zeros(4)
    stats = np.zeros(4)
    rc = lib().ref_hss_bench_toeplitz(n, leaf, rel_tol, abs_tol, nrhs, _dp(times), _dp(stats))
    if rc:
        raise RuntimeError("reference compression failed")
    return dict(compress_s=times[0], factor_s=times[1], solve_s=times[2], apply_s=times[3],
                rank=int(stats[0]), levels=int(stats[1]), resid=stats[2], memory=int(stats[3]))


def blr_front(F11, F12, F21, F22, tiles1, tiles2, rel_tol, abs_tol, adm=None, bsep=None, bupd=None, ysep=None, yupd=None):
    """the reference's BLRMatrix::construct_and_partial_factor on [F11 F12; F21 F22] (ref_driver.cpp: ref_blr_front) and
    the front's two solve phases -> dict(S = Schur complement, ranks, stats, bsep, bupd (forward), ysep (backward))"""
    L = lib()
    vp = C.c_void_p
    L.ref_blr_front.argtypes = [C.c_int, C.c_int, vp, vp, vp, vp, C.c_int, vp, C.c_int, vp, C.c_char_p, C.c_double, C.c_double,
                                C.c_int, vp, vp, vp, vp, vp, vp]
    L.ref_blr_front.restype = C.c_int
    ds, du = F11.shape[0], F12.shape[1]
    f = lambda a: np.array(a, dtype=np.float64, order="F")
    A11, A12, A21, S = f(F11), f(F12), f(F21), f(F22)
    t1, t2 = np.ascontiguousarray(tiles1, dtype=np.int32), np.ascontiguousarray(tiles2, dtype=np.int32)
    nt = len(t1) + len(t2)
    ranks = np.zeros((nt, nt), dtype=np.int32, order="F")
    stats = np.zeros(8)
    nrhs = 0 if bsep is None else np.asarray(bsep).reshape(ds, -1).shape[1]
    mk = lambda a, n: f(np.asarray(a).reshape(n, nrhs, order="F")) if a is not None and n else np.zeros((n, max(nrhs, 1)), order="F")
    bs, bu, ys, yu = mk(bsep, ds), mk(bupd, du), mk(ysep, ds), mk(yupd, du)
    admb = None if adm is None else np.asfortranarray(np.asarray(adm).astype(np.int8)).tobytes(order="F")
    ptr = lambda a: a.ctypes.data if a.size else None
    rc = L.ref_blr_front(ds, du, ptr(A11), ptr(A12), ptr(A21), ptr(S), len(t1), t1.ctypes.data, len(t2), ptr(t2), admb,
                         rel_tol, abs_tol, nrhs, ptr(bs), ptr(bu), ptr(ys), ptr(yu), ranks.ctypes.data, stats.ctypes.data)
    if rc:
        raise RuntimeError("ref_blr_front failed")
    return dict(S=S, ranks=ranks, stats=stats, bsep=bs, bupd=bu, ysep=ys)
