"""CPU tests of the host engine + kernel sources on the fiber emulator (tests/emu): the whole
compress -> mult -> factor -> solve path through the C interface, against the reference's fixtures
and the oracle, at sizes the emulator finishes in seconds.  The GPU twin is tests/test_hss_gpu.py."""
import pytest

import emu_lib
import hss_cases as HC
from strumpack_amd import capi

CASES = HC.golden_cases()
SMALL = ["HSS_seq_%d" % i for i in range(1, 23)]     # the reference's whole CTest sweep for test_HSS_seq


@pytest.fixture(scope="module")
def L():
    return capi.load(emu_lib.build())


@pytest.mark.parametrize("name", SMALL)
def test_ctest_case(L, name):
    HC.check_against_golden(L, CASES[name])


@pytest.mark.parametrize("name", ["HSS_seq_1", "HSS_seq_2", "HSS_seq_5", "HSS_seq_8", "HSS_seq_11", "HSS_seq_12", "HSS_seq_14", "HSS_seq_22"])
def test_schur_complement(L, name):
    from strumpack_amd import hssk as K
    hk = K.Hssk(emu_lib.PATH) if name == "HSS_seq_2" else None
    HC.check_schur(L, CASES[name], hk=hk)
    if hk:
        hk.close()


@pytest.mark.parametrize("name", sorted(HC.sjlt_golden()))
def test_sjlt_sketch(L, name):
    HC.check_sjlt(L, HC.sjlt_golden()[name])


RAGGED = [("T", 2, 1, "stable"), ("T", 3, 1, "original"), ("T", 5, 2, "stable"), ("U", 17, 3, "stable"),
          ("T", 33, 16, "original"), ("L", 37, 5, "stable"), ("T", 64, 16, "stable"), ("U", 65, 16, "original"),
          ("T", 127, 17, "stable"), ("T", 129, 128, "stable"), ("L", 131, 33, "original"), ("T", 257, 7, "stable")]


@pytest.mark.parametrize("prob,n,leaf,algo", RAGGED)
def test_ragged_sizes_vs_oracle(L, prob, n, leaf, algo):
    HC.check_vs_oracle(L, prob, n, leaf, 1e-6, 1e-12, algo, 16, 8)


def test_leaves_beyond_256_rows_vs_oracle(L):
    # leaf size above the register / single-launch kernels' 256 rows (the reference's default leaf size is 512): streaming
    # ID, blocked QR, leaf level of the sweeps as batched launches with the inner levels in the single launch
    HC.check_vs_oracle(L, "T", 640, 320, 1e-6, 1e-12, "stable", 32, 16)


def test_many_samples_vs_oracle(L):
    # more than 256 sample rows: the ID panels take the TSQR pre-reduction (copied out of the samples, not read in place)
    HC.check_vs_oracle(L, "T", 300, 64, 1e-6, 1e-12, "stable", 260, 16)


def test_api_semantics(L):
    HC.check_api_semantics(L)


def test_sweep_plans(L):
    from strumpack_amd import hssk as K
    hk = K.Hssk(emu_lib.PATH)
    HC.check_sweep_plans(L, hk, n=130)
    HC.check_sweep_plans(L, hk, n=130, nrhs=20)
    hk.close()


def test_ulv_inner_levels_one_launch_each(L):
    HC.check_ulv_node(L, n=400)


def test_chain_blocks(L):
    from strumpack_amd import hssk as K
    hk = K.Hssk(emu_lib.PATH)
    HC.check_chain_blocks(L, hk, n=300, leaf=32)
    hk.close()


def test_float_and_complex_instantiations(L):
    HC.check_scz(L)


def test_concurrent_operations_on_one_matrix(L):
    HC.check_concurrent_ops(L)


def test_host_operand_streamed_in_blocks(L):
    HC.check_host_stream_blocks(L)


@pytest.mark.parametrize("name", ["HSS_seq_1", "HSS_seq_4", "HSS_seq_5", "HSS_seq_11", "HSS_seq_14", "HSS_seq_22"])
def test_extract_by_tree_traversal(L, name):
    HC.check_extract(L, CASES[name])


def test_multi_rhs_hybrid_sweeps(L):
    HC.check_multi_rhs(L, n=500, leaf=32, nrhs_list=(5, 13, 20, 64, 100))


def test_multi_rhs_matrix_core_sweeps_large_leaves(L):
    """leaves of 200 rows: the leaf level is its own launch of the matrix-core forward sweep, with 1024 threads per workgroup
    and 32 right-hand sides per pass"""
    HC.check_multi_rhs(L, n=800, leaf=200, nrhs_list=(20, 40))


def test_multi_rhs_matrix_core_sweeps_rank_56(L):
    """inner nodes of 112 rows and rank 56: more 16-row tiles per stage than a wave prefetches, more k-steps than a tile
    prefetches, and vectors that only fit the LDS as 32-wide rows (kernels/hssk_sweep_mma.h)"""
    HC.check_multi_rhs(L, leaf=128, nrhs_list=(40, 70), A=HC.low_rank_plus_identity(1024, 56), d0=96)


def test_blr_dense_slice(L):
    HC.check_blr(L, max_n=1000)


@pytest.mark.parametrize("n,leaf,kind,d0,dd", [(1536, 64, 1, 64, 64), (1024, 128, 2, 128, 64), (700, 64, 1, 64, 32)])
def test_generated_operand_fused_sketch(L, n, leaf, kind, d0, dd):
    # the operand is a formula evaluated inside the sketch kernel (n a multiple of 16 and 64-aligned sample counts: the fused
    # kernel; n = 700: the written-out blocks) -- bitwise the compression of the stored matrix
    from strumpack_amd import hssk as K
    hk = K.Hssk(emu_lib.build())
    HC.check_generator(L, hk, n, leaf, kind, d0=d0, dd=dd)
    hk.close()


def test_inner_levels_in_one_launch(L):
    from strumpack_amd import hssk as K
    hk = K.Hssk(emu_lib.build())
    HC.check_tree_pass(L, hk, sizes=((1024, 64, 1e-6), (515, 64, 1e-6)))   # (the GPU tier: 515 ... 32768 and the N = 1e5 operand)
    hk.close()


def test_factor_ahead_of_the_compression(L):
    HC.check_factor_ahead(L, n=640)   # (the GPU tier: n = 6000, leaf 128)


def test_symmetric_operand_hint(L):
    from strumpack_amd import hssk as K
    hk = K.Hssk(emu_lib.build())
    HC.check_symmetric_hint(L, hk, n=1024, leaf=64)
    hk.close()
