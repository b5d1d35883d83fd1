"""GPU parity tests of the hand-written HIP kernels (product library) through the hssk C-ABI, at
sizes representative of the HSS levels of BASELINE.json's configs."""
import numpy as np
import pytest

import kernel_cases as KC
from strumpack_amd import _loader
from strumpack_amd import hssk as K

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hk():
    h = K.Hssk(_loader.lib_path())
    yield h
    h.close()


def test_gemm_vbatched_ragged(hk):
    KC.case_gemm_vbatched(hk, [(5, 7, 3, 0, 0, 1.0, 0.0), (70, 33, 20, 0, 1, -1.0, 1.0),
                               (16, 130, 17, 1, 0, 2.0, 0.5), (65, 65, 65, 1, 1, 1.0, 0.0),
                               (3, 4, 0, 0, 0, 1.0, 2.0), (0, 4, 3, 0, 0, 1.0, 0.0)])


def test_gemm_vbatched_hss_level_shapes(hk):
    # leaf sample update 192 x 390 x 390, inner 192 x 31 x 28, reduce 192 x 16 x 240, ULV shapes
    shapes = [(192, 390, 390, 0, 1, -1.0, 1.0)] * 8 + [(192, 31, 28, 0, 1, -1.0, 1.0)] * 16 + \
             [(192, 16, 240, 0, 0, 1.0, 1.0)] * 8 + [(350, 390, 40, 1, 0, -1.0, 1.0), (40, 40, 390, 0, 1, 1.0, 0.0),
                                                     (256, 1, 256, 0, 0, 1.0, 0.0), (256, 64, 256, 1, 0, 1.0, 0.0)]
    KC.case_gemm_vbatched(hk, shapes, seed=11)


def test_gemm_vbatched_tall_narrow_updates(hk):
    # the shapes of a BLR front's trailing updates: many rows, a tile's width of columns, A not transposed, B transposed
    KC.case_gemm_vbatched(hk, [(30000, 156, 7, 0, 1, -1.0, 1.0), (5000, 256, 96, 0, 1, -1.0, 1.0), (321, 65, 1, 0, 1, 2.0, 0.0),
                               (7000, 200, 33, 0, 1, 0.5, -1.5), (256, 129, 16, 0, 1, 1.0, 1.0)] + [(2000, 156, 50, 0, 1, -1.0, 1.0)] * 20, seed=21)


def test_gemm_vbatched_tall_path(hk):
    # few columns, B resident in the LDS (leaf-level products with many right-hand sides)
    KC.case_gemm_vbatched(hk, [(256, 64, 256, 0, 0, 1.0, 0.0)] * 40 + [(200, 40, 215, 0, 0, -1.0, 1.0), (130, 17, 33, 1, 0, 2.0, 0.5),
                               (96, 64, 100, 1, 0, 1.0, 1.0), (97, 20, 64, 0, 0, 1.0, 0.0), (256, 64, 215, 0, 0, 1.0, 0.0)], seed=7)


@pytest.mark.parametrize("m,n,k,tb", [(192, 4096, 4096, 1), (192, 4096, 4096, 0), (64, 1000, 3001, 1),
                                      (130, 777, 2050, 0), (200, 65, 50, 0), (16, 64, 16, 1)])
def test_dgemm(hk, m, n, k, tb):
    KC.case_dgemm(hk, m, n, k, tb, alpha=-1.5, beta=0.5)


def test_dgemm_deep_split_edge_tile(hk):
    # a ragged edge tile with >= 32 K-partials: the wide reduce (16 elements x 16 z-lanes per workgroup)
    KC.case_dgemm(hk, 192, 96, 100000, 1, alpha=-1.5, beta=0.5, lda_pad=0, ldb_pad=0)
    KC.case_dgemm(hk, 30, 5, 13000, 0, alpha=1.0, beta=0.0)


@pytest.mark.parametrize("m,n,k,tb", [(192, 4100, 4096, 1), (192, 4100, 4096, 0), (64, 1000, 3008, 1), (128, 640, 2048, 0)])
def test_dgemm_aligned_fast_path(hk, m, n, k, tb):
    KC.case_dgemm(hk, m, n, k, tb, alpha=1.0, beta=0.0, lda_pad=0, ldb_pad=0)
    KC.case_dgemm(hk, m, n, k, tb, alpha=-1.0, beta=1.0, lda_pad=2, ldb_pad=4)


@pytest.mark.parametrize("m,n,k,tb", [(192, 260, 16, 1), (192, 260, 16, 0), (192, 128, 32, 1), (128, 3000, 1600, 0),
                                      (64, 2560, 4800, 1), (64, 256, 80, 0), (192, 33000, 20000, 0), (192, 33000, 20000, 1)])
def test_dgemm_lds_dma_form(hk, m, n, k, tb):
    # the eight-wave form of the interior tiles (BM x 128 per workgroup, operands by LDS DMA into a ring of three stages)
    KC.case_dgemm(hk, m, n, k, tb, alpha=1.0, beta=0.0, lda_pad=0, ldb_pad=0)
    KC.case_dgemm(hk, m, n, k, tb, alpha=-0.5, beta=2.0, lda_pad=4, ldb_pad=2)


@pytest.mark.parametrize("m,n,k,j0,tr,kind", [(192, 3000, 4096, 0, 0, 1), (192, 3000, 4096, 517, 1, 2), (128, 1280, 2048, 5, 0, 2),
                                               (64, 4000, 32000, 100, 1, 1), (192, 33000, 20000, 1000, 0, 1)])
def test_sketch_gen_fused(hk, m, n, k, j0, tr, kind):
    # the operand evaluated inside the eight-wave kernel: bitwise the stored-operand result
    assert KC.case_sketch_gen(hk, m, n, k, j0, tr, kind)
    assert KC.case_sketch_gen(hk, m, n, k, j0, tr, kind, alpha=-0.5, beta=2.0, lda_pad=4)


@pytest.mark.parametrize("m,n,k,j0,tr,kind", [(100, 130, 33, 3, 0, 1), (192, 70, 40, 0, 1, 2), (192, 3000, 5001, 10, 0, 1)])
def test_sketch_gen_written_out_blocks(hk, m, n, k, j0, tr, kind):
    KC.case_sketch_gen(hk, m, n, k, j0, tr, kind, alpha=1.5, beta=0.5, lda_pad=1)


def test_gen_elems(hk):
    KC.case_gen_elems(hk)


def test_leaf_update(hk):
    KC.case_leaf_update(hk, [(24, 20), (192, 45), (64, 33), (2, 1)])
    KC.case_leaf_update(hk, [(192, 195), (192, 196)] * 8 + [(64, 390)], seed=13)


def test_formq_from_stored_reflectors(hk):
    KC.case_qr_lazy(hk, [(40, 12, 12), (100, 64, 64), (195, 128, 128)] * 3)
    KC.case_qr_lazy(hk, [(390, 128, 128), (300, 70, 70)] * 2, seed=18)
    KC.case_qr_lazy(hk, [(600, 20, 20), (1300, 200, 200)], seed=19)


def test_generators(hk):
    KC.case_toeplitz_randn(hk, n=700)


def test_gathers(hk):
    KC.case_gathers(hk)


def test_gemm_vbatched_few_columns(hk):
    # gemv_small_kernel: the shapes of a BLR front's solve phases (V^T x, U t per tile) and of leaf-512 dense products
    KC.case_gemm_vbatched(hk, [(256, 1, 13, 0, 0, -1.0, 1.0)] * 40 + [(13, 1, 256, 1, 0, 1.0, 0.0)] * 40 + [(512, 1, 512, 0, 0, 1.0, 0.0)] * 20 +
                          [(512, 2, 512, 1, 0, 1.0, 1.0)] * 20 + [(300, 4, 1000, 1, 0, 2.0, 0.5), (513, 3, 77, 0, 1, 1.0, 1.0), (100, 2, 130, 1, 1, -1.5, 0.0),
                           (5, 1, 1, 0, 0, 1.0, 0.0), (40, 3, 1100, 0, 0, 1.0, 0.0), (40, 5, 64, 0, 0, 1.0, 0.0)], seed=9)


def test_id(hk):
    KC.case_id(hk, [(24, 40, 1e-6, 1e-12, 1000, 7), (24, 16, 1e-10, 1e-14, 1000, None),
                    (12, 30, 1.0, 1e-10, 1000, None), (24, 40, 1e-8, 1e-12, 5, 9),
                    (70, 20, 1e-4, 1e-10, 1000, 4), (8, 1, 1e-4, 1e-10, 1000, None)])
    KC.case_id(hk, [(192, 390, 1e-4, 1e-10, 50000, 40)] * 6 + [(192, 256, 1e-8, 1e-12, 50000, 60)] * 4 +
               [(192, 54, 1e-4, 1e-10, 50000, 30)] * 8 + [(192, 300, 1e-13, 1e-15, 50000, None)], seed=21)
    # wide (multi-workgroup) path: a few large panels
    KC.case_id(hk, [(192, 195, 1e-4, 1e-10, 50000, 36)] * 5 + [(192, 82, 1e-6, 1e-12, 50000, 40), (100, 90, 1e-9, 1e-14, 1000, 80)], seed=23, deferred=True)
    KC.case_id(hk, [(48, 260, 1e-6, 1e-12, 1000, 6)], seed=24, deferred=True)
    # streaming kernel: BLR tiles (256 x 256, ranks 13 .. 127), leaf-512 sample panels (192 x 391), more than 256 rows
    KC.case_id(hk, [(256, 256, 1e-4, 1e-12, 5000, 13)] * 6 + [(256, 256, 1e-4, 1e-12, 5000, 127), (200, 256, 1e-6, 1e-12, 5000, 70)] +
               [(192, 391, 1e-4, 1e-10, 50000, 41)] * 4 + [(400, 300, 1e-8, 1e-13, 5000, 90), (512, 512, 1e-6, 1e-12, 300, 200)], seed=25)
    KC.case_id(hk, [(600, 260, 1e-6, 1e-12, 1000, 40), (1500, 1200, 1e-8, 1e-12, 50000, 500), (900, 1000, 1e-6, 1e-12, 200, 400)], seed=22)
    # several workgroups per panel, its columns in their registers (id_group_kernel): a BLR block row's worth of 256 x 256 tiles
    # (more workgroups than CUs: the groups at the dispatch frontier wait for their partners), leaf-512 sample panels
    import ctypes
    hk.lib.hssk_id_group_launches.restype = ctypes.c_longlong
    g0 = hk.lib.hssk_id_group_launches()
    KC.case_id(hk, [(256, 256, 1e-4, 1e-12, 129, 13)] * 150 + [(256, 256, 1e-4, 1e-12, 129, 127), (200, 250, 1e-6, 1e-12, 129, 70), (256, 256, 1e-6, 1e-12, 129, None)], seed=26)
    KC.case_id(hk, [(192, 391, 1e-4, 1e-10, 50000, 41)] * 100 + [(150, 500, 1e-6, 1e-12, 12, 60)], seed=27)
    KC.case_id(hk, [(256, 240, 1e-6, 1e-12, 1000, 20), (140, 256, 1e-6, 1e-12, 1000, 5)] * 3, seed=28, deferred=True)
    # the other two instantiations: 256 rows x 512 columns on four workgroups, 192 rows x 256 columns on two
    KC.case_id(hk, [(250, 400, 1e-6, 1e-12, 1000, 90), (256, 512, 1e-8, 1e-13, 1000, 150), (130, 300, 1e-6, 1e-12, 40, 60)] * 4, seed=29)
    KC.case_id(hk, [(160, 250, 1e-6, 1e-12, 1000, 45), (192, 256, 1e-9, 1e-14, 1000, 100), (129, 129, 1e-6, 1e-12, 1000, None)] * 4, seed=30)
    import os
    if "HSSK_ID_NO_GROUP" not in os.environ:
        assert hk.lib.hssk_id_group_launches() >= g0 + 5


def test_qr(hk):
    KC.case_qr(hk, [(40, 12, 12), (30, 30, 30), (33, 20, 33), (10, 1, 10), (70, 10, 0)])
    KC.case_qr(hk, [(390, 350, 390), (390, 128, 128), (256, 240, 256), (54, 24, 54)] * 2, seed=23)
    KC.case_qr(hk, [(1400, 600, 1400), (900, 300, 0), (2100, 130, 130)], seed=24)     # tall blocked path
    KC.case_qr(hk, [(512, 472, 512), (500, 330, 200), (300, 420, 300), (512, 512, 512), (511, 470, 511), (391, 350, 391)] * 3, seed=26)   # panel groups of the fused block reflector
    KC.case_qr(hk, [(195, 159, 195), (208, 160, 0), (200, 180, 200), (196, 161, 196)], seed=25)   # five- and seven-slot register variants


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_random_shapes(hk, seed):
    KC.case_random_shapes(hk, seed, rounds=4)


def test_laswp(hk):
    KC.case_laswp(hk, [(256, 1), (256, 600), (200, 70), (5, 3), (1, 1), (513, 9), (1024, 130), (1100, 3)])


def test_trsm_lu(hk):
    KC.case_trsm_lu(hk)
    KC.case_trsm_lu(hk, seed=10, big_lu=(2100, 5), extra_lu=[(384, 3), (391, 1), (512, 2)])


def test_mfma_peak_probe(hk):
    tf = hk.lib.hssk_mfma_f64_peak_tflops(hk.ctx, 20000)
    print("FP64 MFMA probe: %.1f TFLOP/s" % tf)
    assert tf > 20.0


def test_kernel_matrix_entries(hk):
    KC.case_kernel_eval(hk, n=300)


def test_gram_pchol_id(hk):
    KC.case_gram_pchol_id(hk, [(5000, 195, 1e-3, 1e-12, 1000, 40, 4), (3000, 256, 1e-2, 1e-10, 1000, None, 3), (1520, 100, 1e-4, 1e-12, 17, 30, 2), (2600, 130, 1e-8, 1e-14, 1000, None, 1),
                              (777, 1, 1e-2, 1e-12, 10, None, 1), (600, 255, 1e-12, 1e-14, 1000, None, 2), (40001, 82, 1e-3, 1e-12, 1000, 25, 16)])


def test_gram_gen(hk):
    KC.case_gram_gen(hk)


def test_knn_filtered(hk):
    """The filtered search (FP32 matrix-core filter + exact FP64 selection: large point sets) on small sets, against numpy --
    the same checks as the heap search: exactly a set of k nearest by the float keys, ties by index."""
    import os
    os.environ["HSSK_KNN_FILTER_MIN"] = "600"
    try:
        KC.case_knn(hk, n=2000, d=8, k=64)
        KC.case_knn(hk, n=900, d=3, k=20, seed=24)
        KC.case_knn(hk, n=1100, d=12, k=70, seed=25)
        KC.case_knn(hk, n=1300, d=8, k=128, seed=28)
        KC.case_knn(hk, n=800, d=20, k=8, seed=26)
        KC.case_knn(hk, n=900, d=4, k=70, seed=27, lattice=True)      # exact ties, duplicates
    finally:
        os.environ.pop("HSSK_KNN_FILTER_MIN")


def test_knn(hk):
    KC.case_knn(hk, n=2000, d=8, k=64)
    KC.case_knn(hk, n=300, d=3, k=150, seed=24)
    KC.case_knn(hk, n=40, d=20, k=64, seed=25)
    KC.case_knn(hk, n=500, d=40, k=16, seed=26)
    KC.case_knn(hk, n=900, d=4, k=70, seed=27, lattice=True)   # exact ties, duplicates, two pages
    KC.case_knn(hk, n=700, d=12, k=64, seed=28)                # the 16-coordinate instantiation


def test_kernel_predict(hk):
    KC.case_kernel_predict(hk)


def test_qr_staircase(hk):
    KC.case_qr_staircase(hk, [(195, 2), (196, 2), (140, 2), (130, 4), (200, 3), (256, 2)] * 3)
    KC.case_qr_staircase(hk, [(300, 2), (260, 4)], seed=28)    # tall path (more than 512 rows)


def test_sjlt(hk):
    KC.case_sjlt(hk, n_out=45, K=300, dn=24, nnz=4)
    KC.case_sjlt(hk, n_out=1000, K=3001, dn=192, nnz=4, seed=4)
    KC.case_sjlt(hk, n_out=777, K=2050, dn=64, nnz=2, seed=5)
    KC.case_sjlt(hk, n_out=300, K=1500, dn=300, nnz=8, seed=6)
    KC.case_sjlt(hk, n_out=100, K=900, dn=1000, nnz=3, seed=7)


def test_gather_combine(hk):
    KC.case_gather_combine(hk, [(40, 5, 7, 6, 3, 4, 5, 0, 1), (70, 66, 65, 9, 0, 70, 0, 1, 1), (33, 9, 0, 5, 5, 0, 0, 0, 1),
                                (300, 3, 130, 2, 2, 50, 90, 1, 0)])
    KC.case_gather_combine(hk, [(192, 41, 159, 100, 95, 195, 0, 0, 1), (192, 82, 41, 60, 22, 30, 11, 1, 1), (500, 130, 70, 64, 64, 40, 40, 0, 1)], seed=33)


def test_qr_early_exit(hk):
    KC.case_qr_early_exit(hk, [(195, 192, 1e-4), (195, 128, 1e-6), (64, 48, 1e-3), (100, 60, 0.0), (120, 100, 1e-30)])


def test_ulv_split(hk):
    KC.case_ulv_split(hk, [(40, 7), (33, 33), (70, 0), (82, 41), (5, 2)])
    KC.case_ulv_split(hk, [(195, 36), (196, 30), (256, 100)] * 3, seed=53)


def test_tpqr(hk):
    KC.case_tpqr(hk, [1, 5, 33, 64, 70, 130])
    KC.case_tpqr(hk, [195, 196, 106, 208, 224, 209] * 2, seed=63)


def test_qr_r_only(hk):
    KC.case_qr_r_only(hk, [(60, 40), (128, 100), (208, 195), (256, 120), (300, 64)])


def test_contract_codes(hk):
    KC.case_contract_codes(hk)


def test_expand_image(hk):
    KC.case_expand_image(hk)


def test_upload_two_threads(hk):
    KC.case_upload_two_threads(hk)


def test_colsets(hk):
    KC.case_colsets(hk)
    KC.case_colsets(hk, universe=100000, seed=82)
    KC.case_colsets(hk, universe=1_250_000, seed=83)    # a bitmap of nearly the whole LDS

