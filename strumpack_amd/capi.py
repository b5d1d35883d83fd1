"""ctypes binding of the reference-shaped C interface (include/structured/StructuredMatrix.h):
SP_d_struct_* plus the SPX_* device-operand extensions.  This is the binding a Python user of
STRUMPACK's C API would write; all arithmetic happens in the native library."""
import ctypes as C

import numpy as np

SP_TYPE_HSS, SP_TYPE_BLR = 0, 1


class CSPOptions(C.Structure):
    _fields_ = [("type", C.c_int), ("rel_tol", C.c_double), ("abs_tol", C.c_double),
                ("leaf_size", C.c_int), ("max_rank", C.c_int), ("verbose", C.c_int)]


class SPXHSSOptions(C.Structure):
    _fields_ = [("d0", C.c_int), ("dd", C.c_int), ("p", C.c_int), ("compression_algorithm", C.c_int),
                ("random_engine", C.c_int), ("random_distribution", C.c_int),
                ("compression_sketch", C.c_int), ("sjlt_algo", C.c_int), ("nnz0", C.c_int), ("nnz", C.c_int),
                ("factor_ahead", C.c_int), ("symmetric_operand", C.c_int)]


SP_SYMBOLS = [
    "SP_d_struct_default_options", "SP_d_struct_destroy", "SP_d_struct_rows", "SP_d_struct_cols",
    "SP_d_struct_memory", "SP_d_struct_nonzeros", "SP_d_struct_rank", "SP_d_struct_from_dense",
    "SP_d_struct_from_elements", "SP_d_struct_mult", "SP_d_struct_factor", "SP_d_struct_solve",
    "SP_d_struct_shift",
    "SPX_d_struct_default_hss_options", "SPX_d_struct_from_dense_hss",
    "SPX_d_struct_from_dense_device", "SPX_d_struct_from_dense_device_sharded", "SPX_d_struct_mult_device", "SPX_d_struct_solve_device",
    "SPX_d_struct_levels", "SPX_d_struct_is_compressed", "SPX_d_struct_num_nodes",
    "SPX_d_struct_node_info", "SPX_d_struct_stats", "SPX_d_struct_hssk_ctx",
    "SPX_d_struct_from_kernel", "SPX_d_struct_from_kernel_sharded",
    "SPX_d_struct_from_dense_and_factor", "SPX_comm_unique_id", "SPX_comm_create", "SPX_comm_destroy", "SPX_comm_size", "SPX_comm_rank", "SPX_comm_selftest", "SPX_struct_shard_range",
    "SPX_d_struct_from_dense_device_comm", "SPX_d_struct_from_blocks_device", "SPX_d_struct_from_blocks_device_cb",
    "SPX_d_struct_from_kernel_comm",
    "SPX_d_struct_from_generator", "SPX_d_struct_from_generator_comm", "SPX_d_struct_from_generator_sharded",
    "SPX_d_struct_extract_blocks",
    "SPX_d_blr_front_factor", "SPX_d_blr_front_factor_device", "SPX_d_blr_front_time_phases", "SPX_blr_low_rank_algorithm", "SPX_d_blr_front_forward",
    "SPX_d_blr_front_backward", "SPX_d_blr_front_schur", "SPX_d_blr_front_schur_device", "SPX_d_blr_front_tile_ranks",
    "SPX_d_blr_front_stats", "SPX_d_blr_front_destroy",
    "SPX_tree_pass_launches", "SPX_tree_pass_fallbacks",
    "SPX_device_pool_cached_bytes", "SPX_device_pool_limit_bytes", "SPX_device_pool_trim", "SPX_device_pool_set_limit_gb",
]
ELEM_CB = C.CFUNCTYPE(C.c_double, C.c_int, C.c_int)
ALLGATHER_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_longlong)
STAT_NAMES = ["t_compress", "t_sketch", "t_random", "t_tree", "t_factor", "t_solve", "t_mult",
              "sketch_kernel_ms", "sketch_launches", "rounds", "d_final", "f_sketch", "f_local",
              "f_reduce", "f_id", "f_ortho", "f_ulv", "f_solve", "factor_memory", "sketch_kernel_flops", "sketch_kernel_bytes",
              "b_solve", "b_mult", "t_comm"]


def load(path):
    L = C.CDLL(path)
    vp, dp = C.c_void_p, C.c_void_p
    L.SP_d_struct_default_options.argtypes = [C.POINTER(CSPOptions)]
    L.SP_d_struct_destroy.argtypes = [C.POINTER(vp)]
    for f in ("SP_d_struct_rows", "SP_d_struct_cols", "SP_d_struct_rank", "SPX_d_struct_levels",
              "SPX_d_struct_is_compressed", "SPX_d_struct_num_nodes", "SP_d_struct_factor"):
        getattr(L, f).argtypes = [vp]
        getattr(L, f).restype = C.c_int
    for f in ("SP_d_struct_memory", "SP_d_struct_nonzeros"):
        getattr(L, f).argtypes = [vp]
        getattr(L, f).restype = C.c_longlong
    L.SP_d_struct_from_dense.argtypes = [C.POINTER(vp), C.c_int, C.c_int, dp, C.c_int, C.POINTER(CSPOptions)]
    L.SP_d_struct_from_elements.argtypes = [C.POINTER(vp), C.c_int, C.c_int, ELEM_CB, C.POINTER(CSPOptions)]
    L.SP_d_struct_mult.argtypes = [vp, C.c_char, C.c_int, dp, C.c_int, dp, C.c_int]
    L.SP_d_struct_solve.argtypes = [vp, C.c_int, dp, C.c_int]
    L.SP_d_struct_shift.argtypes = [vp, C.c_double]
    L.SPX_d_struct_default_hss_options.argtypes = [C.POINTER(SPXHSSOptions)]
    L.SPX_d_struct_from_dense_hss.argtypes = [C.POINTER(vp), C.c_int, C.c_int, dp, C.c_int,
                                              C.POINTER(CSPOptions), C.POINTER(SPXHSSOptions)]
    L.SPX_d_struct_from_dense_device.argtypes = [C.POINTER(vp), C.c_int, C.c_int, dp, C.c_longlong,
                                                 C.POINTER(CSPOptions), C.POINTER(SPXHSSOptions)]
    L.SPX_d_struct_from_dense_device_sharded.argtypes = [C.POINTER(vp), C.c_int, C.c_int, dp, C.c_longlong,
                                                         C.POINTER(CSPOptions), C.POINTER(SPXHSSOptions),
                                                         C.c_int, C.c_int, ALLGATHER_CB, vp]
    L.SPX_d_struct_from_kernel.argtypes = [C.POINTER(vp), C.c_int, C.c_int, dp, C.c_int, C.c_double, C.c_double, C.c_int,
                                           C.POINTER(CSPOptions), C.c_int, C.c_int, vp]
    L.SPX_d_struct_from_kernel_sharded.argtypes = [C.POINTER(vp), C.c_int, C.c_int, dp, C.c_int, C.c_double, C.c_double, C.c_int,
                                                   C.POINTER(CSPOptions), C.c_int, C.c_int, vp, C.c_int, C.c_int, ALLGATHER_CB, vp]
    L.SPX_d_struct_from_dense_and_factor.argtypes = [C.POINTER(vp), C.c_int, C.c_int, dp, C.c_int, C.POINTER(CSPOptions)]
    L.SPX_comm_unique_id.argtypes = [C.c_char_p]
    L.SPX_comm_create.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_char_p]
    L.SPX_comm_destroy.argtypes = [C.POINTER(vp)]
    L.SPX_comm_destroy.restype = None
    L.SPX_comm_selftest.argtypes = [vp]
    L.SPX_comm_size.argtypes = [vp]
    L.SPX_comm_rank.argtypes = [vp]
    L.SPX_struct_shard_range.argtypes = [C.c_int, C.POINTER(CSPOptions), C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.SPX_d_struct_from_dense_device_comm.argtypes = [C.POINTER(vp), C.c_int, C.c_int, dp, C.c_longlong,
                                                      C.POINTER(CSPOptions), C.POINTER(SPXHSSOptions), vp]
    L.SPX_d_struct_from_blocks_device.argtypes = [C.POINTER(vp), C.c_int, C.c_int, dp, C.c_longlong, dp, C.c_longlong,
                                                  C.POINTER(CSPOptions), C.POINTER(SPXHSSOptions), vp]
    L.SPX_d_struct_from_blocks_device_cb.argtypes = [C.POINTER(vp), C.c_int, C.c_int, dp, C.c_longlong, dp, C.c_longlong,
                                                     C.POINTER(CSPOptions), C.POINTER(SPXHSSOptions), C.c_int, C.c_int,
                                                     ALLGATHER_CB, vp]
    L.SPX_d_struct_from_kernel_comm.argtypes = [C.POINTER(vp), C.c_int, C.c_int, dp, C.c_int, C.c_double, C.c_double, C.c_int,
                                                C.POINTER(CSPOptions), C.c_int, C.c_int, vp, vp]
    L.SPX_d_struct_from_generator.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.POINTER(CSPOptions), C.POINTER(SPXHSSOptions)]
    L.SPX_d_struct_from_generator_comm.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.POINTER(CSPOptions), C.POINTER(SPXHSSOptions), vp]
    L.SPX_d_struct_from_generator_sharded.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.POINTER(CSPOptions), C.POINTER(SPXHSSOptions),
                                                      C.c_int, C.c_int, ALLGATHER_CB, vp]
    L.SPX_d_struct_mult_device.argtypes = [vp, C.c_char, C.c_int, dp, C.c_longlong, dp, C.c_longlong]
    L.SPX_d_struct_solve_device.argtypes = [vp, C.c_int, dp, C.c_longlong]
    L.SPX_d_struct_node_info.argtypes = [vp, C.POINTER(C.c_int)]
    L.SPX_d_struct_stats.argtypes = [vp, C.POINTER(C.c_double)]
    ll = C.c_longlong
    L.SPX_d_struct_partial_factor.argtypes = [vp]
    L.SPX_d_struct_schur_dims.argtypes = [vp, C.POINTER(C.c_int)]
    L.SPX_d_struct_schur_update.argtypes = [vp, dp, C.c_int, dp, C.c_int, dp, C.c_int, dp, C.c_int]
    L.SPX_d_struct_schur_product_direct.argtypes = [vp, C.c_int, dp, ll, dp, ll, dp, ll, C.c_int]
    L.SPX_d_struct_schur_product_indirect.argtypes = [vp, C.c_int, dp, ll, dp, ll, dp, ll, dp, ll, dp, ll, dp, ll, C.c_int]
    L.SPX_d_struct_mult_child.argtypes = [vp, C.c_int, C.c_char, C.c_int, dp, ll, dp, ll, C.c_int]
    L.SPX_d_struct_hssk_ctx.argtypes = [vp]
    L.SPX_d_struct_hssk_ctx.restype = vp
    for f in ("SPX_tree_pass_launches", "SPX_tree_pass_fallbacks", "SPX_device_pool_cached_bytes", "SPX_device_pool_limit_bytes"):
        getattr(L, f).argtypes = []
        getattr(L, f).restype = C.c_longlong
    L.SPX_device_pool_trim.argtypes = []
    L.SPX_device_pool_trim.restype = None
    L.SPX_device_pool_set_limit_gb.argtypes = [C.c_double]
    L.SPX_device_pool_set_limit_gb.restype = None
    ip = C.POINTER(C.c_int)
    L.SPX_d_struct_extract_blocks.argtypes = [vp, C.c_int, ip, ip, ip, ip, C.POINTER(C.c_void_p), ip, C.c_int, C.c_int]
    L.SPX_d_blr_front_factor.argtypes = [C.POINTER(vp), C.c_int, C.c_int, dp, C.c_int, dp, C.c_int, dp, C.c_int, dp, C.c_int,
                                         C.c_int, ip, C.c_int, ip, C.c_char_p, C.POINTER(CSPOptions)]
    L.SPX_d_blr_front_factor_device.argtypes = [C.POINTER(vp), C.c_int, C.c_int, dp, ll, dp, ll, dp, ll, dp, ll,
                                                C.c_int, ip, C.c_int, ip, C.c_char_p, C.POINTER(CSPOptions)]
    L.SPX_d_blr_front_time_phases.argtypes = [C.c_int]
    L.SPX_d_blr_front_time_phases.restype = None
    L.SPX_blr_low_rank_algorithm.argtypes = [C.c_int]
    L.SPX_blr_low_rank_algorithm.restype = C.c_int
    L.SPX_d_blr_front_forward.argtypes = [vp, C.c_int, dp, C.c_int, dp, C.c_int]
    L.SPX_d_blr_front_backward.argtypes = [vp, C.c_int, dp, C.c_int, dp, C.c_int]
    L.SPX_d_blr_front_schur.argtypes = [vp, dp, C.c_int]
    L.SPX_d_blr_front_schur_device.argtypes = [vp, C.POINTER(ll)]
    L.SPX_d_blr_front_schur_device.restype = vp
    L.SPX_d_blr_front_tile_ranks.argtypes = [vp, ip]
    L.SPX_d_blr_front_stats.argtypes = [vp, C.POINTER(C.c_double)]
    L.SPX_d_blr_front_destroy.argtypes = [C.POINTER(vp)]
    L.SPX_d_blr_front_destroy.restype = None
    return L


class BLRFront:
    """Partially factored BLR frontal matrix [F11 F12; F21 F22] behind the handle SPXBLRFront
    (BLR::BLRMatrix<double>::construct_and_partial_factor of the reference, BLR/BLRMatrix.cpp:740)."""
    STAT_NAMES = ["t_factor", "ms_lu", "ms_compress", "ms_trsm", "ms_schur", "f_schur", "f_total", "nnz11", "nnz12", "nnz21",
                  "max_rank", "schur_launches", "b_schur", "_r13", "_r14", "_r15"]

    def __init__(self, lib, handle, dsep, dupd, nt1, nt2):
        self.L, self.h, self.dsep, self.dupd, self.nt1, self.nt2 = lib, handle, dsep, dupd, nt1, nt2

    @staticmethod
    def _tiles(t):
        a = np.ascontiguousarray(t, dtype=np.int32)
        return a, a.ctypes.data_as(C.POINTER(C.c_int))

    @staticmethod
    def _adm(adm, nt1):
        if adm is None:
            return None
        a = np.asfortranarray(np.asarray(adm).astype(np.int8).reshape(nt1, nt1))
        return a.tobytes(order="F")

    @classmethod
    def factor(cls, lib, F11, F12, F21, F22, tiles1, tiles2, opts, admissible=None):
        """host operands; returns (front, Schur complement F22 - F21 F11^{-1} F12)"""
        ds, du = F11.shape[0], (F12.shape[1] if F12 is not None else 0)
        f = lambda a: None if a is None else np.asfortranarray(a, dtype=np.float64)
        F11, F12, F21 = f(F11), f(F12), f(F21)
        S = np.array(F22, dtype=np.float64, order="F") if F22 is not None else np.zeros((du, du), order="F")
        t1, p1 = cls._tiles(tiles1)
        t2, p2 = cls._tiles(tiles2)
        ptr = lambda a: a.ctypes.data if a is not None and a.size else None
        h = C.c_void_p()
        if lib.SPX_d_blr_front_factor(C.byref(h), ds, du, ptr(F11), max(ds, 1), ptr(F12), max(ds, 1), ptr(F21), max(du, 1),
                                      ptr(S), max(du, 1), len(t1), p1, len(t2), p2, cls._adm(admissible, len(t1)), C.byref(opts)):
            raise RuntimeError("SPX_d_blr_front_factor failed")
        return cls(lib, h, ds, du, len(t1), len(t2)), S

    @classmethod
    def factor_device(cls, lib, dsep, dupd, dF11, ld11, dF12, ld12, dF21, ld21, dF22, ld22, tiles1, tiles2, opts, admissible=None):
        t1, p1 = cls._tiles(tiles1)
        t2, p2 = cls._tiles(tiles2)
        h = C.c_void_p()
        if lib.SPX_d_blr_front_factor_device(C.byref(h), dsep, dupd, dF11, ld11, dF12, ld12, dF21, ld21, dF22, ld22,
                                             len(t1), p1, len(t2), p2, cls._adm(admissible, len(t1)), C.byref(opts)):
            raise RuntimeError("SPX_d_blr_front_factor_device failed")
        return cls(lib, h, dsep, dupd, len(t1), len(t2))

    def forward(self, bsep, bupd=None):
        bs = np.array(bsep, dtype=np.float64, order="F").reshape(self.dsep, -1, order="F")
        bu = np.array(bupd, dtype=np.float64, order="F").reshape(self.dupd, bs.shape[1], order="F") if self.dupd else None
        if self.L.SPX_d_blr_front_forward(self.h, bs.shape[1], bs.ctypes.data, max(self.dsep, 1),
                                          bu.ctypes.data if bu is not None else None, max(self.dupd, 1)):
            raise RuntimeError("SPX_d_blr_front_forward failed")
        return bs, bu

    def backward(self, ysep, yupd=None):
        ys = np.array(ysep, dtype=np.float64, order="F").reshape(self.dsep, -1, order="F")
        yu = np.asfortranarray(yupd, dtype=np.float64).reshape(self.dupd, ys.shape[1], order="F") if self.dupd else None
        if self.L.SPX_d_blr_front_backward(self.h, ys.shape[1], ys.ctypes.data, max(self.dsep, 1),
                                           yu.ctypes.data if yu is not None else None, max(self.dupd, 1)):
            raise RuntimeError("SPX_d_blr_front_backward failed")
        return ys

    def solve11(self, b):
        """B11 \\ b: forward and backward phases with an empty update part"""
        ys, _ = self.forward(b, np.zeros((self.dupd, np.asarray(b).reshape(self.dsep, -1).shape[1])) if self.dupd else None)
        return self.backward(ys, np.zeros((self.dupd, ys.shape[1])) if self.dupd else None)

    def schur(self):
        S = np.zeros((self.dupd, self.dupd), order="F")
        if self.dupd and self.L.SPX_d_blr_front_schur(self.h, S.ctypes.data, self.dupd):
            raise RuntimeError("SPX_d_blr_front_schur failed")
        return S

    def schur_device(self):
        ld = C.c_longlong()
        return self.L.SPX_d_blr_front_schur_device(self.h, C.byref(ld)), ld.value

    def tile_ranks(self):
        nt = self.nt1 + self.nt2
        out = np.zeros((nt, nt), dtype=np.int32, order="F")
        if self.L.SPX_d_blr_front_tile_ranks(self.h, out.ctypes.data_as(C.POINTER(C.c_int))):
            raise RuntimeError("SPX_d_blr_front_tile_ranks failed")
        return out

    def stats(self):
        out = (C.c_double * 16)()
        self.L.SPX_d_blr_front_stats(self.h, out)
        return dict(zip(self.STAT_NAMES, list(out)))

    def destroy(self):
        if self.h:
            self.L.SPX_d_blr_front_destroy(C.byref(self.h))
            self.h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class StructuredMatrix:
    """HSS matrix behind the C handle CSPStructMat (structured::StructuredMatrix<double>)."""

    def __init__(self, lib, handle, n):
        self.L, self.h, self.n = lib, handle, n

    # ---- construction -------------------------------------------------------------------------
    @staticmethod
    def options(lib, rel_tol=None, abs_tol=None, leaf_size=None, max_rank=None, verbose=False,
                type=SP_TYPE_HSS):
        o = CSPOptions()
        lib.SP_d_struct_default_options(C.byref(o))
        o.type = type
        o.verbose = int(verbose)
        if rel_tol is not None:
            o.rel_tol = rel_tol
        if abs_tol is not None:
            o.abs_tol = abs_tol
        if leaf_size is not None:
            o.leaf_size = leaf_size
        if max_rank is not None:
            o.max_rank = max_rank
        return o

    @staticmethod
    def hss_options(lib, d0=None, dd=None, p=None, algorithm=None, random_engine=None, sketch=None, sjlt_algo=None,
                    nnz0=None, nnz=None, factor_ahead=None, symmetric=None):
        h = SPXHSSOptions()
        lib.SPX_d_struct_default_hss_options(C.byref(h))
        if d0 is not None:
            h.d0 = d0
        if dd is not None:
            h.dd = dd
        if p is not None:
            h.p = p
        if algorithm is not None:
            h.compression_algorithm = {"original": 0, "stable": 1, "hard_restart": 2}[algorithm]
        if random_engine is not None:
            h.random_engine = {"linear": 0, "mersenne": 1, "philox": 2}[random_engine]
        if sketch is not None:
            h.compression_sketch = {"gaussian": 0, "sjlt": 1}[sketch.lower()]
        if sjlt_algo is not None:
            h.sjlt_algo = {"chunk": 0, "perm": 1}[sjlt_algo]
        if nnz0 is not None:
            h.nnz0 = nnz0
        if nnz is not None:
            h.nnz = nnz
        if factor_ahead is not None:
            h.factor_ahead = int(bool(factor_ahead))
        if symmetric is not None:
            h.symmetric_operand = int(symmetric)
        return h

    @classmethod
    def from_dense(cls, lib, A, opts, hss=None):
        A = np.asfortranarray(A, dtype=np.float64)
        h = C.c_void_p()
        if hss is None:
            rc = lib.SP_d_struct_from_dense(C.byref(h), A.shape[0], A.shape[1], A.ctypes.data, A.shape[0], C.byref(opts))
        else:
            rc = lib.SPX_d_struct_from_dense_hss(C.byref(h), A.shape[0], A.shape[1], A.ctypes.data, A.shape[0],
                                                 C.byref(opts), C.byref(hss))
        if rc:
            raise RuntimeError("SP_d_struct_from_dense failed")
        return cls(lib, h, A.shape[0])

    @classmethod
    def from_dense_and_factor(cls, lib, A, opts):
        """structured::construct_and_factor_from_dense (BLR: LU while compressing; HSS: construct + factor)"""
        A = np.asfortranarray(A, dtype=np.float64)
        h = C.c_void_p()
        if lib.SPX_d_struct_from_dense_and_factor(C.byref(h), A.shape[0], A.shape[1], A.ctypes.data, A.shape[0], C.byref(opts)):
            raise RuntimeError("SPX_d_struct_from_dense_and_factor failed")
        return cls(lib, h, A.shape[0])

    @classmethod
    def from_dense_device(cls, lib, dptr, n, lda, opts, hss=None):
        h = C.c_void_p()
        rc = lib.SPX_d_struct_from_dense_device(C.byref(h), n, n, dptr, lda, C.byref(opts),
                                                C.byref(hss) if hss is not None else None)
        if rc:
            raise RuntimeError("SPX_d_struct_from_dense_device failed")
        return cls(lib, h, n)

    @classmethod
    def from_generator(cls, lib, n, kind, opts, hss=None):
        """the matrix is one of the library's formulas (kind 1: Toeplitz 1/(1+|i-j|), 2: its upper triangle): never stored"""
        h = C.c_void_p()
        if lib.SPX_d_struct_from_generator(C.byref(h), n, kind, C.byref(opts), C.byref(hss) if hss is not None else None):
            raise RuntimeError("SPX_d_struct_from_generator failed")
        return cls(lib, h, n)

    @classmethod
    def from_elements(cls, lib, n, fn, opts):
        cb = ELEM_CB(fn)
        h = C.c_void_p()
        if lib.SP_d_struct_from_elements(C.byref(h), n, n, cb, C.byref(opts)):
            raise RuntimeError("SP_d_struct_from_elements failed")
        return cls(lib, h, n)

    def destroy(self):
        if self.h:
            self.L.SP_d_struct_destroy(C.byref(self.h))
            self.h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass

    # ---- operations ---------------------------------------------------------------------------
    def mult(self, B, trans="N"):
        B = np.asfortranarray(B, dtype=np.float64).reshape(self.n, -1, order="F")
        Cm = np.zeros_like(B, order="F")
        if self.L.SP_d_struct_mult(self.h, trans.encode(), B.shape[1], B.ctypes.data, self.n, Cm.ctypes.data, self.n):
            raise RuntimeError("SP_d_struct_mult failed")
        return Cm

    def factor(self):
        if self.L.SP_d_struct_factor(self.h):
            raise RuntimeError("SP_d_struct_factor failed")

    def solve(self, B):
        X = np.array(B, dtype=np.float64, order="F").reshape(self.n, -1, order="F")
        if self.L.SP_d_struct_solve(self.h, X.shape[1], X.ctypes.data, self.n):
            raise RuntimeError("SP_d_struct_solve failed")
        return X

    def shift(self, s):
        if self.L.SP_d_struct_shift(self.h, s):
            raise RuntimeError("SP_d_struct_shift failed")

    def mult_device(self, dB, dC, nrhs, trans="N"):
        if self.L.SPX_d_struct_mult_device(self.h, trans.encode(), nrhs, dB, self.n, dC, self.n):
            raise RuntimeError("SPX_d_struct_mult_device failed")

    def solve_device(self, dB, nrhs):
        if self.L.SPX_d_struct_solve_device(self.h, nrhs, dB, self.n):
            raise RuntimeError("SPX_d_struct_solve_device failed")

    def dense(self):
        return self.mult(np.eye(self.n))

    def extract_blocks(self, I, J, add_to=None):
        """[H(I[b], J[b]) for b]: HSSMatrix::extract of a batch of requests by tree traversal on the device (one call);
        add_to: list of arrays that are incremented instead (extract_add)"""
        nb = len(I)
        ia = lambda v: np.ascontiguousarray(v, dtype=np.int32)
        r, c = ia(np.concatenate([ia(x) for x in I]) if nb else []), ia(np.concatenate([ia(x) for x in J]) if nb else [])
        roff, coff = ia(np.cumsum([0] + [len(x) for x in I])), ia(np.cumsum([0] + [len(x) for x in J]))
        out = [np.asfortranarray(a, dtype=np.float64) for a in add_to] if add_to is not None else \
            [np.zeros((len(I[b]), len(J[b])), order="F") for b in range(nb)]
        ptrs = (C.c_void_p * max(nb, 1))(*[o.ctypes.data for o in out])
        ldo = ia([max(o.shape[0], 1) for o in out])
        ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int))
        if self.L.SPX_d_struct_extract_blocks(self.h, nb, ip(r), ip(roff), ip(c), ip(coff), ptrs, ip(ldo), int(add_to is not None), 0):
            raise RuntimeError("SPX_d_struct_extract_blocks failed")
        return out

    # ---- Schur complement of the (0,0) block (HSSMatrix::partial_factor / Schur_update / Schur_product_*) ----
    def partial_factor(self):
        if self.L.SPX_d_struct_partial_factor(self.h):
            raise RuntimeError("SPX_d_struct_partial_factor failed")

    def schur_dims(self):
        d = (C.c_int * 7)()
        if self.L.SPX_d_struct_schur_dims(self.h, d):
            raise RuntimeError("SPX_d_struct_schur_dims failed")
        return dict(zip(("n0", "n1", "rV0", "mu0", "rV1", "rU0", "rU1"), list(d)))

    def schur_update(self):
        """-> Theta (n1 x rV0), DUB01 (mu0 x rV1), Phi (n1 x mu0), Vhat (mu0 x rV0)"""
        d = self.schur_dims()
        mk = lambda r, c: np.zeros((r, c), order="F")
        Th, DU, Ph, Vh = mk(d["n1"], d["rV0"]), mk(d["mu0"], d["rV1"]), mk(d["n1"], d["mu0"]), mk(d["mu0"], d["rV0"])
        ld = lambda a: max(a.shape[0], 1)
        if self.L.SPX_d_struct_schur_update(self.h, Th.ctypes.data, ld(Th), DU.ctypes.data, ld(DU), Ph.ctypes.data, ld(Ph),
                                            Vh.ctypes.data, ld(Vh)):
            raise RuntimeError("SPX_d_struct_schur_update failed")
        return Th, DU, Ph, Vh

    def schur_product_direct(self, R):
        n1 = self.schur_dims()["n1"]
        R = np.asfortranarray(R, dtype=np.float64).reshape(n1, -1, order="F")
        Sr, Sc = np.zeros_like(R, order="F"), np.zeros_like(R, order="F")
        if self.L.SPX_d_struct_schur_product_direct(self.h, R.shape[1], R.ctypes.data, n1, Sr.ctypes.data, n1, Sc.ctypes.data, n1, 0):
            raise RuntimeError("SPX_d_struct_schur_product_direct failed")
        return Sr, Sc

    def schur_product_indirect(self, R0, R1, Sr1, Sc1):
        d = self.schur_dims()
        n0, n1 = d["n0"], d["n1"]
        f = lambda a, n: np.asfortranarray(a, dtype=np.float64).reshape(n, -1, order="F")
        R0, R1, Sr1, Sc1 = f(R0, n0), f(R1, n1), f(Sr1, n1), f(Sc1, n1)
        Sr, Sc = np.zeros_like(R1, order="F"), np.zeros_like(R1, order="F")
        if self.L.SPX_d_struct_schur_product_indirect(self.h, R1.shape[1], R0.ctypes.data, max(n0, 1), R1.ctypes.data, n1,
                                                      Sr1.ctypes.data, n1, Sc1.ctypes.data, n1, Sr.ctypes.data, n1,
                                                      Sc.ctypes.data, n1, 0):
            raise RuntimeError("SPX_d_struct_schur_product_indirect failed")
        return Sr, Sc

    def mult_child(self, child, B, trans="N"):
        d = self.schur_dims()
        n = d["n0"] if child == 0 else d["n1"]
        B = np.asfortranarray(B, dtype=np.float64).reshape(n, -1, order="F")
        Cm = np.zeros_like(B, order="F")
        if self.L.SPX_d_struct_mult_child(self.h, child, trans.encode(), B.shape[1], B.ctypes.data, n, Cm.ctypes.data, n, 0):
            raise RuntimeError("SPX_d_struct_mult_child failed")
        return Cm

    # ---- introspection ------------------------------------------------------------------------
    def rows(self):
        return self.L.SP_d_struct_rows(self.h)

    def rank(self):
        return self.L.SP_d_struct_rank(self.h)

    def memory(self):
        return self.L.SP_d_struct_memory(self.h)

    def nonzeros(self):
        return self.L.SP_d_struct_nonzeros(self.h)

    def levels(self):
        return self.L.SPX_d_struct_levels(self.h)

    def is_compressed(self):
        return bool(self.L.SPX_d_struct_is_compressed(self.h))

    def node_info(self):
        nn = self.L.SPX_d_struct_num_nodes(self.h)
        out = np.zeros((nn, 6), dtype=np.int32)
        self.L.SPX_d_struct_node_info(self.h, out.ctypes.data_as(C.POINTER(C.c_int)))
        return out

    def stats(self):
        out = (C.c_double * 24)()
        self.L.SPX_d_struct_stats(self.h, out)
        return dict(zip(STAT_NAMES, list(out)))
