"""ctypes binding of the thin kernel-level C-ABI (include/hssk.h).

Used by the parity tests and bench.py to drive individual HIP kernels; the HSS engine itself is
native C++ (strumpack_amd/csrc/host) and calls the same entry points directly.
"""
import ctypes as C

import numpy as np

c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int)


class GemmDesc(C.Structure):
    _fields_ = [("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p),
                ("m", C.c_int), ("n", C.c_int), ("k", C.c_int),
                ("lda", C.c_int), ("ldb", C.c_int), ("ldc", C.c_int),
                ("transA", C.c_int), ("transB", C.c_int),
                ("alpha", C.c_double), ("beta", C.c_double)]


class LeafUpdateDesc(C.Structure):
    _fields_ = [("R", C.c_void_p), ("D", C.c_void_p), ("Sr", C.c_void_p), ("Sc", C.c_void_p),
                ("d", C.c_int), ("m", C.c_int), ("ldr", C.c_int), ("ldd", C.c_int), ("lds", C.c_int)]


class KernelSpec(C.Structure):
    _fields_ = [("X", C.c_void_p), ("n", C.c_longlong), ("d", C.c_int), ("type", C.c_int), ("p", C.c_int),
                ("h", C.c_double), ("lam", C.c_double)]


class KevalDesc(C.Structure):
    _fields_ = [("ri", C.c_void_p), ("ci", C.c_void_p), ("out", C.c_void_p),
                ("nr", C.c_int), ("nc", C.c_int), ("ldo", C.c_int), ("r0", C.c_int), ("c0", C.c_int)]


class ColsetDesc(C.Structure):
    _fields_ = [("src0", C.c_void_p), ("src1", C.c_void_p), ("n0", C.c_int), ("n1", C.c_int), ("lo", C.c_int), ("hi", C.c_int),
                ("out", C.c_void_p), ("count", C.c_void_p), ("n0_dev", C.c_void_p), ("n1_dev", C.c_void_p)]


class ColGatherDesc(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("idx", C.c_void_p),
                ("rows", C.c_int), ("ncols", C.c_int), ("lds", C.c_int), ("ldd", C.c_int),
                ("scatter", C.c_int)]


class RowGatherDesc(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("idx", C.c_void_p),
                ("nrows", C.c_int), ("cols", C.c_int), ("lds", C.c_int), ("ldd", C.c_int),
                ("scatter", C.c_int), ("accumulate", C.c_int)]


class ElemDesc(C.Structure):
    _fields_ = [("A", C.c_void_p), ("lda", C.c_longlong), ("I", C.c_void_p), ("J", C.c_void_p),
                ("i0", C.c_int), ("j0", C.c_int), ("B", C.c_void_p),
                ("m", C.c_int), ("n", C.c_int), ("ldb", C.c_int), ("transpose", C.c_int),
                ("rlo", C.c_int), ("rhi", C.c_int), ("clo", C.c_int), ("chi", C.c_int)]   # ownership windows (0, 0: none)


class TransposeDesc(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p),
                ("rows", C.c_int), ("cols", C.c_int), ("lds", C.c_int), ("ldd", C.c_int)]


class IdDesc(C.Structure):
    _fields_ = [("W", C.c_void_p), ("ldw", C.c_int), ("d", C.c_int), ("m", C.c_int),
                ("rtol", C.c_double), ("atol", C.c_double), ("max_rank", C.c_int),
                ("perm", C.c_void_p), ("rank", C.c_void_p), ("work", C.c_void_p),
                ("src", C.c_void_p), ("lds", C.c_int), ("defer_x", C.c_int)]


class CombineDesc(C.Structure):
    _fields_ = [("G0", C.c_void_p), ("G1", C.c_void_p), ("ldg", C.c_int), ("gsplit", C.c_int), ("gidx", C.c_void_p),
                ("M0", C.c_void_p), ("M1", C.c_void_p), ("ldm", C.c_int), ("msplit", C.c_int), ("midx", C.c_void_p),
                ("C", C.c_void_p), ("csj", C.c_int), ("csk", C.c_int), ("alpha", C.c_double),
                ("out", C.c_void_p), ("ldo", C.c_int), ("rows", C.c_int), ("J", C.c_int), ("K", C.c_int)]


class UlvSplitDesc(C.Structure):
    _fields_ = [("D", C.c_void_p), ("ldd", C.c_int), ("m", C.c_int), ("r", C.c_int), ("perm", C.c_void_p),
                ("X", C.c_void_p), ("ldx", C.c_int), ("W1", C.c_void_p), ("ldw", C.c_int), ("W0t", C.c_void_p), ("ldt", C.c_int)]


class TpqrDesc(C.Structure):
    _fields_ = [("R1", C.c_void_p), ("ld1", C.c_int), ("R2", C.c_void_p), ("ld2", C.c_int), ("m", C.c_int)]


class XsolveDesc(C.Structure):
    _fields_ = [("W", C.c_void_p), ("ldw", C.c_int), ("rank", C.c_int), ("m", C.c_int),
                ("X", C.c_void_p), ("ldx", C.c_int), ("solved", C.c_int)]


class PcholDesc(C.Structure):
    _fields_ = [("G", C.c_void_p), ("ldg", C.c_int), ("m", C.c_int), ("rtol", C.c_double), ("atol", C.c_double),
                ("max_rank", C.c_int), ("perm", C.c_void_p), ("rank", C.c_void_p), ("R", C.c_void_p), ("ldr", C.c_int)]


class SumDesc(C.Structure):
    _fields_ = [("P", C.c_void_p), ("stride", C.c_longlong), ("n", C.c_longlong), ("count", C.c_int), ("out", C.c_void_p)]


class GramDesc(C.Structure):
    _fields_ = [("W", C.c_void_p), ("ldw", C.c_int), ("rows", C.c_int), ("m", C.c_int), ("G", C.c_void_p), ("ldg", C.c_int)]


class GramGenDesc(C.Structure):
    _fields_ = [("ri", C.c_void_p), ("r0", C.c_int), ("ci", C.c_void_p), ("c0", C.c_int), ("rows", C.c_int), ("m", C.c_int),
                ("G", C.c_void_p), ("ldg", C.c_int)]


class QrDesc(C.Structure):
    _fields_ = [("A", C.c_void_p), ("lda", C.c_int), ("rows", C.c_int), ("cols", C.c_int),
                ("Q", C.c_void_p), ("ldq", C.c_int), ("nq", C.c_int),
                ("rdiag", C.c_void_p), ("work", C.c_void_p), ("stair", C.c_int),
                ("stop_rel", C.c_double), ("stop_abs", C.c_double), ("r_only", C.c_int)]


class TrsmDesc(C.Structure):
    _fields_ = [("T", C.c_void_p), ("B", C.c_void_p), ("n", C.c_int), ("nrhs", C.c_int),
                ("ldt", C.c_int), ("ldb", C.c_int), ("lower", C.c_int), ("transT", C.c_int),
                ("unit", C.c_int), ("Tinv", C.c_void_p)]


class TrtriDesc(C.Structure):
    _fields_ = [("R", C.c_void_p), ("Tinv", C.c_void_p), ("n", C.c_int), ("ldr", C.c_int), ("mode", C.c_int)]


class LuDesc(C.Structure):
    _fields_ = [("A", C.c_void_p), ("n", C.c_int), ("lda", C.c_int), ("piv", C.c_void_p),
                ("info", C.c_void_p)]


class LuSolveDesc(C.Structure):
    _fields_ = [("LU", C.c_void_p), ("piv", C.c_void_p), ("B", C.c_void_p),
                ("n", C.c_int), ("nrhs", C.c_int), ("lda", C.c_int), ("ldb", C.c_int)]


class NormDesc(C.Structure):
    _fields_ = [("P", C.c_void_p), ("rows", C.c_int), ("cols", C.c_int), ("ld", C.c_int),
                ("out", C.c_void_p)]


class ShiftDesc(C.Structure):
    _fields_ = [("A", C.c_void_p), ("n", C.c_int), ("lda", C.c_int)]


class Gen(C.Structure):
    """hssk_gen: a matrix given by a formula (kind 1: Toeplitz 1/(1+|i-j|), 2: its upper triangle)"""
    _fields_ = [("kind", C.c_int), ("reserved", C.c_int), ("p", C.c_double * 4)]


HSSK_SYMBOLS = [
    "hssk_ctx_create", "hssk_ctx_destroy", "hssk_ctx_stream", "hssk_sync", "hssk_last_error",
    "hssk_malloc", "hssk_free", "hssk_memcpy_h2d", "hssk_memcpy_d2h", "hssk_last_dgemm_ms", "hssk_last_dgemm_flops", "hssk_last_dgemm_trace",
    "hssk_fill_toeplitz", "hssk_randn", "hssk_dgemm", "hssk_gemm_vbatched", "hssk_gather_cols",
    "hssk_gather_rows", "hssk_gather_elems", "hssk_transpose", "hssk_id_vbatched",
    "hssk_qr_vbatched", "hssk_trsm_vbatched", "hssk_getrf_vbatched", "hssk_getrs_vbatched",
    "hssk_sumsq_vbatched", "hssk_shift_diag", "hssk_mfma_f64_peak_tflops", "hssk_memcpy_d2d",
    "hssk_memcpy2d_h2d", "hssk_memcpy2d_d2h", "hssk_memset_zero", "hssk_is_device_pointer",
    "hssk_basis_dense", "hssk_mfma_f64_probe", "hssk_last_dgemm_clock_ghz", "hssk_leaf_update_vbatched", "hssk_formq_vbatched",
    "hssk_kernel_eval_vbatched", "hssk_knn", "hssk_kernel_predict", "hssk_copy_triu",
    "hssk_laswp_vbatched", "hssk_shift_diag_cplx", "hssk_upload_async", "hssk_h2d_block_async", "hssk_h2d_bytes_async", "hssk_expand_image", "hssk_copy_fence", "hssk_compute_fence", "hssk_compute_mark", "hssk_copy_wait", "hssk_id_xsolve_vbatched", "hssk_id_solves_inline", "hssk_gather_combine", "hssk_ulv_split", "hssk_tpqr_vbatched", "hssk_fill_toeplitz_block", "hssk_sum_slabs", "hssk_ulv_fwd_sweep", "hssk_ulv_bwd_sweep", "hssk_apply_sweep", "hssk_sweep_status", "hssk_sweep_arm", "hssk_trtri_diag_vbatched", "hssk_sjlt_dense", "hssk_sjlt_sketch",
    "hssk_plan_begin", "hssk_plan_end", "hssk_plan_replay", "hssk_plan_destroy", "hssk_plan_size",
    "hssk_sketch_gen", "hssk_gen_elems", "hssk_gen_fill", "hssk_colsets", "hssk_colsets_max_universe",
    "hssk_cluster_median", "hssk_pchol_id_vbatched", "hssk_pchol_id_max_m", "hssk_pchol_id_rank_cap", "hssk_sum_partials", "hssk_gram_vbatched", "hssk_gram_gen_vbatched", "hssk_gram_gen_supported",
]


class DevArray:
    """A device buffer with numpy-like shape metadata (column-major)."""

    def __init__(self, hk, shape, dtype=np.float64):
        self.hk, self.shape, self.dtype = hk, tuple(shape), np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        self.ptr = hk.lib.hssk_malloc(max(self.nbytes, 8))
        if not self.ptr:
            raise MemoryError(hk.error())

    def free(self):
        if self.ptr:
            self.hk.lib.hssk_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def get(self):
        out = np.empty(self.shape, dtype=self.dtype, order="F")
        if self.nbytes:
            self.hk.check(self.hk.lib.hssk_memcpy_d2h(self.hk.ctx, out.ctypes.data, self.ptr, self.nbytes))
        return out

    def set(self, arr):
        a = np.asfortranarray(arr, dtype=self.dtype)
        assert a.shape == self.shape
        if self.nbytes:
            self.hk.check(self.hk.lib.hssk_memcpy_h2d(self.hk.ctx, self.ptr, a.ctypes.data, self.nbytes))
        return self

    def at(self, *idx):
        """device address of element idx (column-major)"""
        off, stride = 0, 1
        for i, n in zip(idx, self.shape):
            off += i * stride
            stride *= n
        return self.ptr + off * self.dtype.itemsize


class Hssk:
    def __init__(self, path, device=0):
        self.lib = L = C.CDLL(path)
        L.hssk_malloc.restype = C.c_void_p
        L.hssk_malloc.argtypes = [C.c_longlong]
        L.hssk_free.argtypes = [C.c_void_p]
        L.hssk_last_error.restype = C.c_char_p
        L.hssk_ctx_create.argtypes = [C.POINTER(C.c_void_p), C.c_int]
        L.hssk_ctx_destroy.argtypes = [C.c_void_p]
        L.hssk_ctx_stream.restype = C.c_void_p
        L.hssk_ctx_stream.argtypes = [C.c_void_p]
        L.hssk_sync.argtypes = [C.c_void_p]
        L.hssk_memcpy_h2d.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong]
        L.hssk_memcpy_d2h.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong]
        L.hssk_memcpy_d2d.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong]
        L.hssk_memset_zero.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong]
        for f in ("hssk_memcpy2d_h2d", "hssk_memcpy2d_d2h"):
            getattr(L, f).argtypes = [C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p, C.c_longlong,
                                      C.c_longlong, C.c_longlong]
        L.hssk_is_device_pointer.argtypes = [C.c_void_p]
        L.hssk_last_dgemm_clock_ghz.restype = C.c_double
        L.hssk_last_dgemm_clock_ghz.argtypes = [C.c_void_p]
        L.hssk_last_dgemm_ms.restype = C.c_float
        L.hssk_last_dgemm_ms.argtypes = [C.c_void_p]
        L.hssk_last_dgemm_trace.restype = C.c_longlong
        L.hssk_last_dgemm_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong]
        L.hssk_last_dgemm_flops.restype = C.c_double
        L.hssk_last_dgemm_flops.argtypes = [C.c_void_p]
        L.hssk_fill_toeplitz.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_longlong, C.c_char]
        L.hssk_fill_toeplitz_block.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_longlong, C.c_int, C.c_int, C.c_char]
        L.hssk_sum_slabs.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong, C.c_longlong, C.c_int, C.c_void_p]
        L.hssk_upload_async.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong]
        L.hssk_h2d_block_async.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p, C.c_longlong, C.c_longlong, C.c_longlong]
        L.hssk_copy_fence.argtypes = [C.c_void_p]
        L.hssk_compute_fence.argtypes = [C.c_void_p]
        L.hssk_h2d_bytes_async.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p, C.c_longlong, C.c_longlong, C.c_longlong]
        L.hssk_expand_image.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p, C.c_longlong, C.c_longlong, C.c_longlong, C.c_int]
        L.hssk_compute_mark.argtypes = [C.c_void_p, C.c_int]
        L.hssk_copy_wait.argtypes = [C.c_void_p, C.c_int]
        L.hssk_sweep_status.argtypes = [C.c_void_p]
        L.hssk_randn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_longlong, C.c_longlong,
                                 C.c_int, C.c_longlong, C.c_ulonglong]
        L.hssk_dgemm.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_longlong, C.c_longlong,
                                 C.c_double, C.c_void_p, C.c_longlong, C.c_void_p, C.c_longlong,
                                 C.c_double, C.c_void_p, C.c_longlong]
        for name in ("hssk_gemm_vbatched", "hssk_gather_cols", "hssk_gather_rows",
                     "hssk_gather_elems", "hssk_transpose", "hssk_id_vbatched", "hssk_qr_vbatched",
                     "hssk_trsm_vbatched", "hssk_getrf_vbatched", "hssk_getrs_vbatched",
                     "hssk_sumsq_vbatched", "hssk_leaf_update_vbatched", "hssk_formq_vbatched",
                     "hssk_id_xsolve_vbatched", "hssk_gather_combine", "hssk_ulv_split", "hssk_tpqr_vbatched"):
            getattr(L, name).argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.hssk_sketch_gen.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_longlong, C.c_longlong, C.c_longlong,
                                      C.c_double, C.c_void_p, C.c_longlong, C.c_double, C.c_void_p, C.c_longlong]
        L.hssk_gen_elems.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.hssk_gen_fill.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_longlong, C.c_longlong, C.c_longlong,
                                    C.c_longlong, C.c_int]
        L.hssk_id_solves_inline.argtypes = [C.c_int, C.c_int]
        L.hssk_shift_diag.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double]
        L.hssk_kernel_eval_vbatched.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.hssk_knn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.hssk_kernel_predict.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.hssk_colsets.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.hssk_cluster_median.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        for fn in ("hssk_pchol_id_vbatched", "hssk_sum_partials", "hssk_gram_vbatched"):
            getattr(L, fn).argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.hssk_pchol_id_rank_cap.argtypes = [C.c_int]
        L.hssk_gram_gen_vbatched.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.hssk_gram_gen_supported.argtypes = [C.c_void_p, C.c_int]
        L.hssk_colsets_max_universe.argtypes = []
        L.hssk_colsets_max_universe.restype = C.c_longlong
        L.hssk_sjlt_dense.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_longlong, C.c_longlong, C.c_void_p, C.c_int]
        L.hssk_sjlt_sketch.argtypes = [C.c_void_p, C.c_int, C.c_longlong, C.c_longlong, C.c_void_p, C.c_longlong,
                                       C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_longlong]
        L.hssk_mfma_f64_peak_tflops.restype = C.c_double
        L.hssk_mfma_f64_peak_tflops.argtypes = [C.c_void_p, C.c_int]
        L.hssk_mfma_f64_probe.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
        ctx = C.c_void_p()
        if L.hssk_ctx_create(C.byref(ctx), device):
            raise RuntimeError(self.error())
        self.ctx = ctx

    def error(self):
        return self.lib.hssk_last_error().decode()

    def check(self, rc):
        if rc:
            raise RuntimeError(self.error())

    def close(self):
        if self.ctx:
            self.lib.hssk_ctx_destroy(self.ctx)
            self.ctx = None

    def sync(self):
        self.check(self.lib.hssk_sync(self.ctx))

    def empty(self, shape, dtype=np.float64):
        return DevArray(self, shape, dtype)

    def array(self, arr, dtype=None):
        a = np.asarray(arr)
        d = DevArray(self, a.shape, dtype or a.dtype)
        return d.set(a)

    def batch(self, fn_name, descs):
        if not descs:
            return
        arr = (type(descs[0]) * len(descs))(*descs)
        self.check(getattr(self.lib, fn_name)(self.ctx, arr, len(descs)))
