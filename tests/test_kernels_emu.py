"""CPU tests of the HIP kernel sources on the fiber emulator (small shapes): index arithmetic,
masking, descriptor staging, wave-collective structure.  The GPU run of the same cases is
tests/test_kernels_gpu.py."""
import pytest

import emu_lib
import kernel_cases as KC
from strumpack_amd import hssk as K


@pytest.fixture(scope="module")
def hk():
    h = K.Hssk(emu_lib.build())
    yield h
    h.close()


def test_gemm_vbatched(hk):
    KC.case_gemm_vbatched(hk, [(5, 7, 3, 0, 0, 1.0, 0.0), (70, 33, 20, 0, 1, -1.0, 1.0),
                               (16, 130, 17, 1, 0, 2.0, 0.5), (65, 65, 65, 1, 1, 1.0, 0.0),
                               (3, 4, 0, 0, 0, 1.0, 2.0), (0, 4, 3, 0, 0, 1.0, 0.0)])


@pytest.mark.parametrize("m,n,k,tb", [(192, 70, 40, 1), (100, 130, 33, 0), (16, 64, 16, 1),
                                      (200, 65, 50, 0)])
def test_dgemm(hk, m, n, k, tb):
    KC.case_dgemm(hk, m, n, k, tb, alpha=-1.5, beta=0.5)


def test_leaf_update(hk):
    KC.case_leaf_update(hk, [(24, 20), (192, 45), (64, 33), (2, 1)])


def test_gemm_vbatched_tall_narrow_updates(hk):
    # the shapes of a BLR front's trailing updates: many rows, a tile's width of columns, A not transposed, B transposed
    KC.case_gemm_vbatched(hk, [(300, 156, 7, 0, 1, -1.0, 1.0), (256, 256, 40, 0, 1, -1.0, 1.0), (321, 65, 1, 0, 1, 2.0, 0.0),
                               (700, 200, 33, 0, 1, 0.5, -1.5), (256, 129, 16, 0, 1, 1.0, 1.0)], seed=21)


def test_gemm_vbatched_panel_path(hk):
    # m = sample count (even, <= 192), A contiguous and aligned: the 192 x 32 panel kernel
    KC.case_gemm_vbatched(hk, [(192, 45, 45, 0, 1, -1.0, 1.0), (192, 45, 45, 0, 0, -1.0, 1.0), (96, 33, 20, 0, 1, 1.0, 0.0),
                               (128, 7, 50, 0, 0, 2.0, 0.5), (66, 64, 17, 0, 1, 1.0, 1.0)], seed=4, even_ld=True)


def test_gemm_vbatched_tall_path(hk):
    # few columns, B resident in the LDS, 128-row blocks of C per workgroup (the leaf-level products of a mat-vec / solve with
    # many right-hand sides): ragged rows and columns, either form of A, beta, k not a multiple of the chunk
    KC.case_gemm_vbatched(hk, [(256, 64, 256, 0, 0, 1.0, 0.0), (200, 40, 215, 0, 0, -1.0, 1.0), (130, 17, 33, 1, 0, 2.0, 0.5),
                               (96, 64, 100, 1, 0, 1.0, 1.0), (97, 20, 64, 0, 0, 1.0, 0.0)], seed=7)


def test_gemm_vbatched_few_columns(hk):
    # at most four columns, k <= 1024: the vector entries in the LDS, A streamed once (gemv_small_kernel): a thread per row of
    # C (A not transposed) or a wave per row (transposed), either form of B, beta, ragged blocks; k beyond the LDS copy and
    # five columns stay with the tile kernel
    KC.case_gemm_vbatched(hk, [(300, 1, 256, 0, 0, 1.0, 0.0), (13, 1, 256, 1, 0, 1.0, 0.0), (256, 1, 13, 0, 0, -1.0, 1.0),
                               (70, 4, 1000, 1, 0, 2.0, 0.5), (513, 3, 77, 0, 1, 1.0, 1.0), (100, 2, 130, 1, 1, -1.5, 0.0),
                               (5, 1, 1, 0, 0, 1.0, 0.0), (40, 3, 1100, 0, 0, 1.0, 0.0), (40, 5, 64, 0, 0, 1.0, 0.0)], seed=9)


@pytest.mark.parametrize("m,n,k,tb", [(192, 150, 64, 1), (192, 150, 64, 0), (64, 130, 48, 1), (128, 64, 32, 0)])
def test_dgemm_aligned_fast_path(hk, m, n, k, tb):
    # even leading dimensions + 16-byte aligned operands: interior tiles take the unmasked kernel
    KC.case_dgemm(hk, m, n, k, tb, alpha=1.0, beta=0.0, lda_pad=0, ldb_pad=0)
    KC.case_dgemm(hk, m, n, k, tb, alpha=-1.0, beta=1.0, lda_pad=2, ldb_pad=4)


@pytest.mark.parametrize("m,n,k,tb", [(192, 260, 16, 1), (192, 260, 16, 0), (192, 128, 32, 1), (128, 300, 160, 0),
                                      (64, 256, 48, 1), (64, 256, 80, 0), (192, 390, 112, 0), (192, 384, 3200, 1)])
def test_dgemm_lds_dma_form(hk, m, n, k, tb):
    # the eight-wave form of the interior tiles (BM x 128 per workgroup, operands by LDS DMA into a ring of three stages):
    # one, two, many stages per chunk (ring wrap-around, clamped copies), all three heights, both operand images, ragged
    # rest columns through the masked kernel, several K-chunks
    KC.case_dgemm(hk, m, n, k, tb, alpha=1.0, beta=0.0, lda_pad=0, ldb_pad=0)
    KC.case_dgemm(hk, m, n, k, tb, alpha=-0.5, beta=2.0, lda_pad=4, ldb_pad=2)


@pytest.mark.parametrize("m,n,k,j0,tr,kind", [(192, 260, 64, 0, 0, 1), (192, 260, 64, 37, 1, 2), (128, 128, 32, 5, 0, 2),
                                               (64, 400, 3200, 100, 1, 1), (192, 384, 16, 0, 0, 1)])
def test_sketch_gen_fused(hk, m, n, k, j0, tr, kind):
    # the operand evaluated inside the eight-wave kernel: bitwise the stored-operand result
    assert KC.case_sketch_gen(hk, m, n, k, j0, tr, kind)
    assert KC.case_sketch_gen(hk, m, n, k, j0, tr, kind, alpha=-0.5, beta=2.0, lda_pad=4)


@pytest.mark.parametrize("m,n,k,j0,tr,kind", [(100, 130, 33, 3, 0, 1), (192, 70, 40, 0, 1, 2), (192, 300, 50, 10, 0, 1)])
def test_sketch_gen_written_out_blocks(hk, m, n, k, j0, tr, kind):
    # shapes outside the fused kernel (ragged k, odd sample counts, narrow outputs): column blocks are written out
    KC.case_sketch_gen(hk, m, n, k, j0, tr, kind, alpha=1.5, beta=0.5, lda_pad=1)


def test_gen_elems(hk):
    KC.case_gen_elems(hk)


def test_dgemm_splitk(hk):
    KC.case_dgemm(hk, 24, 70, 3000, 1)


def test_dgemm_deep_split_edge_tile(hk):
    # a ragged edge tile with >= 32 K-partials: the wide reduce (16 elements x 16 z-lanes per workgroup)
    KC.case_dgemm(hk, 64, 72, 12800, 1, alpha=-1.5, beta=0.5, lda_pad=0, ldb_pad=0)
    KC.case_dgemm(hk, 30, 5, 13000, 0, alpha=1.0, beta=0.0)


def test_generators(hk):
    KC.case_toeplitz_randn(hk)


def test_gathers(hk):
    KC.case_gathers(hk)


def test_id(hk):
    KC.case_id(hk, [(24, 40, 1e-6, 1e-12, 1000, 7), (24, 16, 1e-10, 1e-14, 1000, None),
                    (12, 30, 1.0, 1e-10, 1000, None), (24, 40, 1e-8, 1e-12, 5, 9),
                    (70, 20, 1e-4, 1e-10, 1000, 4), (8, 1, 1e-4, 1e-10, 1000, None)])
    KC.case_id(hk, [(192, 150, 1e-6, 1e-12, 1000, 18), (130, 100, 1e-6, 1e-12, 1000, 11)], seed=6)   # <3,13>, <3,8>
    KC.case_id(hk, [(200, 70, 1e-6, 1e-12, 1000, 9), (48, 210, 1e-6, 1e-12, 1000, 6)], seed=7)        # <4,8>, fallback
    KC.case_id(hk, [(192, 150, 1e-6, 1e-12, 1000, 18), (64, 40, 1e-6, 1e-12, 1000, 7), (100, 90, 1e-9, 1e-14, 1000, 80)], seed=9, deferred=True)   # in-place source, deferred X (rank 80: the large-rank branch)
    KC.case_id(hk, [(48, 260, 1e-6, 1e-12, 1000, 6)], seed=10, deferred=True)   # a batch the in-place kernels take: X is copied
    KC.case_id(hk, [(600, 260, 1e-6, 1e-12, 1000, 40), (520, 300, 1e-8, 1e-12, 25, 60)], seed=8)      # wide (multi-workgroup) path
    # streaming kernel (panels beyond the register tiles): 4 / 8 rows per lane, ranks below and above 64 (blocked X solve),
    # a panel that never meets its tolerance, max_rank cut-off, deferred X on such a batch
    KC.case_id(hk, [(96, 250, 1e-6, 1e-12, 1000, 12), (130, 230, 1e-8, 1e-13, 1000, 70), (64, 300, 1e-6, 1e-12, 9, 20)], seed=12)
    KC.case_id(hk, [(300, 120, 1e-6, 1e-12, 1000, 30), (270, 240, 1e-9, 1e-14, 1000, 150), (260, 100, 1e-13, 1e-16, 1000, None)], seed=13)
    KC.case_id(hk, [(100, 240, 1e-6, 1e-12, 1000, 70)], seed=14, deferred=True)
    # two / four workgroups per panel with its columns in their registers (id_group_kernel): 256 x 256 tiles (BLR), the 192 x 391
    # sample panels of leaf size 512, ranks below and above 64, a max_rank cut-off, a full-rank panel, deferred X from a source
    import ctypes
    hk.lib.hssk_id_group_launches.restype = ctypes.c_longlong
    g0 = hk.lib.hssk_id_group_launches()
    KC.case_id(hk, [(256, 256, 1e-6, 1e-12, 1000, 30), (200, 250, 1e-8, 1e-13, 1000, 100), (256, 256, 1e-6, 1e-12, 129, None)], seed=15)
    KC.case_id(hk, [(192, 391, 1e-6, 1e-12, 1000, 40), (150, 500, 1e-6, 1e-12, 12, 60)], seed=16)
    KC.case_id(hk, [(256, 240, 1e-6, 1e-12, 1000, 20), (140, 256, 1e-6, 1e-12, 1000, 5)], seed=17, deferred=True)
    KC.case_id(hk, [(250, 400, 1e-6, 1e-12, 1000, 90), (130, 300, 1e-6, 1e-12, 40, 60)], seed=18)        # 256 rows, four workgroups
    KC.case_id(hk, [(160, 250, 1e-6, 1e-12, 1000, 45), (129, 129, 1e-6, 1e-12, 1000, None)], seed=19)   # 192 rows, two workgroups
    import os
    if (os.cpu_count() or 1) >= 4 and "HSSK_ID_NO_GROUP" not in os.environ and "HSSK_EMU_THREADS" not in os.environ:
        assert hk.lib.hssk_id_group_launches() == g0 + 5


def test_qr(hk):
    KC.case_qr(hk, [(40, 12, 12), (30, 30, 30), (33, 20, 33), (10, 1, 10), (70, 10, 0)])
    KC.case_qr(hk, [(100, 128, 100), (120, 60, 120)], seed=8)       # <2,8,16>
    KC.case_qr(hk, [(195, 128, 128)], seed=9)                       # <4,8,16>
    KC.case_qr(hk, [(195, 160, 195), (130, 100, 130)], seed=10)     # <4,26,8>
    KC.case_qr(hk, [(300, 40, 300), (260, 250, 260), (390, 350, 390), (300, 60, 0), (280, 300, 200)], seed=11)   # blocked (compact WY + batched GEMM)
    KC.case_qr(hk, [(512, 472, 512), (500, 330, 200), (300, 420, 300), (511, 470, 511)], seed=13)   # panel groups: several groups, a ragged last one, thin Q, columns beyond the last panel, odd row counts (unaligned 16-byte loads)
    KC.case_qr(hk, [(600, 20, 30), (700, 90, 100), (530, 70, 0)], seed=12)   # tall blocked path (GEMM-assembled compact WY)


def test_formq_from_stored_reflectors(hk):
    KC.case_qr_lazy(hk, [(40, 12, 12), (100, 64, 64), (195, 128, 128)])   # register kernels
    KC.case_qr_lazy(hk, [(300, 70, 70)], seed=18)                          # blocked
    KC.case_qr_lazy(hk, [(600, 20, 20), (640, 70, 70)], seed=19)           # tall blocked path


@pytest.mark.parametrize("seed", [1, 2])
def test_random_shapes(hk, seed):
    KC.case_random_shapes(hk, seed, rounds=3)


def test_laswp(hk):
    KC.case_laswp(hk, [(256, 1), (200, 70), (5, 3), (1, 1), (513, 9), (1024, 2), (1100, 3)])


def test_trsm_lu(hk):
    KC.case_trsm_lu(hk)
    KC.case_trsm_lu(hk, seed=11, big_lu=(530, 2), extra_lu=[(300, 2), (391, 1), (512, 1)])   # 512-lane form: 64 trailing columns per pass


def test_kernel_matrix_entries(hk):
    KC.case_kernel_eval(hk, n=200)


def test_gram_pchol_id(hk):
    KC.case_gram_pchol_id(hk, [(300, 40, 1e-3, 1e-12, 1000, 12, 1), (700, 70, 1e-2, 1e-10, 1000, None, 3), (520, 33, 1e-4, 1e-12, 5, 9, 2), (260, 130, 1e-8, 1e-14, 1000, None, 1)])


def test_gram_gen(hk):
    KC.case_gram_gen(hk)


def test_knn_filtered(hk):
    """The filtered search (FP32 matrix-core filter + exact FP64 selection: large point sets) on small sets, against numpy --
    the same checks as the heap search: exactly a set of k nearest by the float keys, ties by index."""
    import os
    os.environ["HSSK_KNN_FILTER_MIN"] = "600"
    try:
        KC.case_knn(hk, n=700, d=8, k=10)
        KC.case_knn(hk, n=900, d=3, k=20, seed=24)
        KC.case_knn(hk, n=1100, d=12, k=70, seed=25)                  # the kept list longer than a page of the heap search
        KC.case_knn(hk, n=800, d=20, k=8, seed=26)
        KC.case_knn(hk, n=900, d=4, k=12, seed=27, lattice=True)      # exact ties, duplicates
    finally:
        os.environ.pop("HSSK_KNN_FILTER_MIN")


def test_knn(hk):
    KC.case_knn(hk, n=150, d=8, k=10)
    KC.case_knn(hk, n=90, d=3, k=70, seed=24)     # two pages
    KC.case_knn(hk, n=40, d=20, k=64, seed=25)    # k > n - 1
    KC.case_knn(hk, n=300, d=4, k=70, seed=26, lattice=True)   # exact ties, duplicates, two pages


def test_kernel_predict(hk):
    KC.case_kernel_predict(hk)


def test_qr_staircase(hk):
    KC.case_qr_staircase(hk, [(140, 2), (195, 2), (196, 2), (70, 4), (150, 3), (256, 2)])
    KC.case_qr_staircase(hk, [(300, 2), (140, 4)], seed=28)      # tall path


def test_sjlt(hk):
    KC.case_sjlt(hk, n_out=45, K=300, dn=24, nnz=4)
    KC.case_sjlt(hk, n_out=130, K=77, dn=200, nnz=2, seed=4)
    KC.case_sjlt(hk, n_out=20, K=130, dn=600, nnz=8, seed=5)


def test_gather_combine(hk):
    KC.case_gather_combine(hk, [(40, 5, 7, 6, 3, 4, 5, 0, 1), (70, 66, 65, 9, 0, 70, 0, 1, 1), (33, 9, 0, 5, 5, 0, 0, 0, 1),
                                (300, 3, 130, 2, 2, 50, 90, 1, 0)])


def test_qr_early_exit(hk):
    KC.case_qr_early_exit(hk, [(195, 192, 1e-4), (195, 128, 1e-6), (64, 48, 1e-3), (100, 60, 0.0), (120, 100, 1e-30)])


def test_ulv_split(hk):
    KC.case_ulv_split(hk, [(40, 7), (33, 33), (70, 0), (82, 41), (5, 2)])


def test_tpqr(hk):
    KC.case_tpqr(hk, [1, 5, 33, 64, 70, 130])


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

