"""GPU parity tests of the HSS hot path (product library, hand-written HIP kernels) through the
reference's C interface: the reference's whole CTest sweep for test_HSS_seq against fixtures generated
by the reference itself, BASELINE.json configs 1/2, and size-independent properties at full size."""
import os

import numpy as np
import pytest

import hss_cases as HC
from oracle import hss_oracle as O
from strumpack_amd import _loader, capi
from strumpack_amd import hssk as K

pytestmark = pytest.mark.gpu
CASES = HC.golden_cases()


@pytest.fixture(scope="module")
def L():
    return capi.load(_loader.lib_path())


@pytest.mark.parametrize("name", sorted(CASES))
def test_ctest_case(L, name):
    c = CASES[name]
    HC.check_against_golden(L, c, compare_oracle=c["n"] <= 4096)


@pytest.mark.parametrize("name", ["HSS_seq_1", "HSS_seq_2", "HSS_seq_5", "HSS_seq_8", "HSS_seq_11", "HSS_seq_12",
                                  "HSS_seq_14", "HSS_seq_22", "config1_T4096_defaults",
                                  "config2shape_T8192_leaf256_rtol1e-4"])
def test_schur_complement(L, name):
    hk = K.Hssk(_loader.lib_path())
    HC.check_schur(L, CASES[name], dense_check=CASES[name]["n"] <= 4096, hk=hk)
    hk.close()


@pytest.mark.parametrize("name", sorted(HC.sjlt_golden()))
def test_sjlt_sketch(L, name):
    HC.check_sjlt(L, HC.sjlt_golden()[name])


def test_sjlt_sketch_full_size(L):
    """SJLT sketch at BASELINE size with A in HBM: the streaming kernels (one pass over A per product) must give a
    matrix as accurate as the Gaussian sketch's (test_full_size_properties)."""
    n = 32768
    hk = K.Hssk(_loader.lib_path())
    dA = hk.empty((n, n))
    hk.check(hk.lib.hssk_fill_toeplitz(hk.ctx, dA.ptr, n, n, b"T"))
    hk.sync()
    o = capi.StructuredMatrix.options(L, rel_tol=1e-4, abs_tol=1e-8, leaf_size=256, max_rank=50000)
    h = capi.StructuredMatrix.hss_options(L, sketch="sjlt")
    H = capi.StructuredMatrix.from_dense_device(L, dA.ptr, n, n, o, h)
    assert H.is_compressed() and 26 <= H.rank() <= 40, H.rank()
    st = H.stats()
    assert st["sketch_kernel_bytes"] >= 2 * 8.0 * n * n     # both products went through the streaming kernels
    rng = np.random.default_rng(0)
    cols = rng.integers(0, n, 16)
    E = np.zeros((n, 16))
    E[cols, np.arange(16)] = 1.0
    i = np.arange(n)
    Acols = 1.0 / (1.0 + np.abs(i[:, None] - cols[None, :]))
    err = np.linalg.norm(H.mult(E) - Acols) / np.linalg.norm(Acols)
    assert err < 2e-4, err
    H.factor()
    b = rng.standard_normal((n, 2))
    X = H.solve(b)
    assert np.linalg.norm(H.mult(X) - b) / np.linalg.norm(b) <= 1e-12
    H.destroy()
    dA.free()
    hk.close()


def test_many_samples_vs_oracle(L):
    # more than 256 sample rows: the ID panels take the TSQR pre-reduction (copied out of the samples, not read in place)
    HC.check_vs_oracle(L, "T", 300, 64, 1e-6, 1e-12, "stable", 260, 16)


def test_api_semantics(L):
    HC.check_api_semantics(L)


def test_sweep_plans(L):
    from strumpack_amd import hssk as K
    hk = K.Hssk(_loader.lib_path())
    HC.check_sweep_plans(L, hk, n=3000)
    HC.check_sweep_plans(L, hk, n=3000, nrhs=20)
    hk.close()


def test_ulv_inner_levels_one_launch_each(L):
    HC.check_ulv_node(L, n=20000, leaf=128)


def test_chain_blocks(L):
    from strumpack_amd import hssk as K
    hk = K.Hssk(_loader.lib_path())
    HC.check_chain_blocks(L, hk, n=6000, leaf=64)
    hk.close()


def test_native_code_is_loaded():
    import os
    maps = open("/proc/self/maps").read()
    assert "libstrumpack_amd.so" in maps and os.path.exists(_loader.LIB_PATH)


def fullsize_golden():
    import json
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hss_fullsize_golden.json")) as f:
        return {c["name"]: c for c in json.load(f)["cases"]}


@pytest.mark.parametrize("name", ["config2_T32768_leaf256", "config3_T100000_leaf256", "config2_T32768_leaf512", "config3_T100000_leaf512"])
def test_full_size_against_reference(L, name):
    """BASELINE.json configs[1] / configs[2] AT FULL SIZE against the reference run at the same size
    (tests/golden/make_golden_fullsize.py: HSSMatrix(A, opts) at N = 32768, compress(Amult, Aelem) with an O(N^2) Toeplitz
    product at N = 100000): A generated in HBM, the reference's default random stream (engine "linear"), then every node's
    rank, levels, memory / nonzeros, ||H b||, ||H^T b||, ||x|| for the reference's b, sampled error, ULV residual."""
    c = fullsize_golden()[name]
    n, leaf = c["n"], c["leaf_size"]
    hk = K.Hssk(_loader.lib_path())
    dA = hk.empty((n, n))
    hk.check(hk.lib.hssk_fill_toeplitz(hk.ctx, dA.ptr, n, n, b"T"))
    hk.sync()
    o = capi.StructuredMatrix.options(L, rel_tol=c["rel_tol"], abs_tol=c["abs_tol"], leaf_size=leaf, max_rank=50000)
    h = capi.StructuredMatrix.hss_options(L, d0=c["d0"], dd=c["dd"], random_engine="linear")
    H = capi.StructuredMatrix.from_dense_device(L, dA.ptr, n, n, o, h)
    assert H.is_compressed() and H.levels() == c["levels"]
    ni, ref = H.node_info(), np.array(c["nodes"])
    assert np.array_equal(ni[:, [0, 1, 5]], ref[:, [0, 1, 5]]), "tree shape differs from the reference"
    dr = np.abs(ni[:, 3] - ref[:, 3]) + np.abs(ni[:, 4] - ref[:, 4])
    assert dr.max() <= 1 and (dr > 0).mean() <= 0.05, f"node ranks differ from the reference: {int(dr.sum())} over {len(dr)} nodes"
    assert abs(H.rank() - c["rank"]) <= 1
    HC.check_memory(H, c, ni, ref)
    b = HC.randn(n)
    assert np.isclose(np.linalg.norm(H.mult(b)), c["mult_b_norm"], rtol=1e-5)
    assert np.isclose(np.linalg.norm(H.mult(b, "T")), c["multT_b_norm"], rtol=1e-5)
    cols = np.random.default_rng(0).integers(0, n, 64)
    E = np.zeros((n, 64))
    E[cols, np.arange(64)] = 1.0
    HE = H.mult(E)
    i = np.arange(n)
    Acols = np.where(i[:, None] == cols[None, :], 1.0, 1.0 / (1.0 + np.abs(i[:, None] - cols[None, :])))
    err = np.linalg.norm(HE - Acols) / np.linalg.norm(Acols)
    assert err <= 1e2 * c["rel_tol"] and err <= 1.1 * c["rel_err_sampled"] + 1e-12, (err, c["rel_err_sampled"])
    # transpose consistency on the sampled block, linearity
    HtE = H.mult(E, "T")
    assert np.allclose(HE[cols], HtE[cols].T, atol=1e-12)
    rng = np.random.default_rng(1)
    x, y = rng.standard_normal(n), rng.standard_normal(n)
    assert np.allclose(H.mult(2 * x - 3 * y)[:, 0], 2 * H.mult(x)[:, 0] - 3 * H.mult(y)[:, 0], atol=1e-9)
    H.factor()
    xs = H.solve(b)[:, 0]
    assert np.linalg.norm(H.mult(xs)[:, 0] - b) <= 1e-12 * np.linalg.norm(b)
    assert np.isclose(np.linalg.norm(xs), c["x_norm"], rtol=1e-6)
    assert np.allclose(xs[:64], np.array(c["x_head"]), rtol=1e-6, atol=1e-9)
    # the solution of the compressed system solves the dense one to O(rel_tol)
    assert np.linalg.norm(Acols.T @ xs - b[cols]) <= 1e-2 * np.linalg.norm(b[cols])
    H.destroy()
    dA.free()
    hk.close()


def test_full_size_philox_sketch_same_quality(L):
    """the bench's configuration (Philox samples drawn on the device instead of the host stream): not bit-comparable with the
    reference, held to its rank within 15 % and its sampled error"""
    c = fullsize_golden()["config3_T100000_leaf256"]
    n = c["n"]
    hk = K.Hssk(_loader.lib_path())
    dA = hk.empty((n, n))
    hk.check(hk.lib.hssk_fill_toeplitz(hk.ctx, dA.ptr, n, n, b"T"))
    hk.sync()
    o = capi.StructuredMatrix.options(L, rel_tol=c["rel_tol"], abs_tol=c["abs_tol"], leaf_size=c["leaf_size"], max_rank=50000)
    H = capi.StructuredMatrix.from_dense_device(L, dA.ptr, n, n, o, capi.StructuredMatrix.hss_options(L, random_engine="philox"))
    assert H.is_compressed() and H.levels() == c["levels"] and abs(H.rank() - c["rank"]) <= 0.15 * c["rank"]
    assert abs(H.memory() - c["memory"]) <= 0.03 * c["memory"]
    cols = np.random.default_rng(0).integers(0, n, 64)
    E = np.zeros((n, 64))
    E[cols, np.arange(64)] = 1.0
    i = np.arange(n)
    Acols = np.where(i[:, None] == cols[None, :], 1.0, 1.0 / (1.0 + np.abs(i[:, None] - cols[None, :])))
    err = np.linalg.norm(H.mult(E) - Acols) / np.linalg.norm(Acols)
    assert err <= 1.5 * c["rel_err_sampled"], (err, c["rel_err_sampled"])
    H.destroy()
    dA.free()
    hk.close()


@pytest.mark.parametrize("name", ["HSS_seq_1", "HSS_seq_4", "HSS_seq_5", "HSS_seq_11", "HSS_seq_12", "HSS_seq_14", "HSS_seq_22",
                                  "config1_T4096_defaults", "config2shape_T8192_leaf256_rtol1e-4"])
def test_extract_by_tree_traversal(L, name):
    HC.check_extract(L, CASES[name])


def test_extract_thousand_blocks_full_size(L):
    """1000 random 8 x 8 blocks of the N = 100000 matrix in one call: against products with unit vectors on a sample, and
    the device path (outputs resident in HBM) timed -- the assembly of a front asks for blocks like these"""
    import ctypes as C
    import time
    n = 100000
    hk = K.Hssk(_loader.lib_path())
    dA = hk.empty((n, n))
    hk.check(hk.lib.hssk_fill_toeplitz(hk.ctx, dA.ptr, n, n, b"T"))
    hk.sync()
    o = capi.StructuredMatrix.options(L, rel_tol=1e-4, abs_tol=1e-8, leaf_size=256, max_rank=50000)
    H = capi.StructuredMatrix.from_dense_device(L, dA.ptr, n, n, o, capi.StructuredMatrix.hss_options(L, random_engine="philox"))
    dA.free()
    rng = np.random.default_rng(4)
    nb = 1000
    I = [rng.integers(0, n, 8) for _ in range(nb)]
    J = [rng.integers(0, n, 8) for _ in range(nb)]
    out = H.extract_blocks(I, J)
    # reference on 16 of the requests: columns of H through products with unit vectors
    for b in range(0, nb, 64):
        E = np.zeros((n, 8))
        E[J[b], np.arange(8)] = 1.0
        # (repeated column indices: the unit vectors add up; compare column by column instead)
        for q in range(8):
            e = np.zeros((n, 1))
            e[J[b][q]] = 1.0
            col = H.mult(e)[:, 0]
            assert np.abs(out[b][:, q] - col[I[b]]).max() <= 1e-12
    # device outputs: the two launches plus the index upload
    ia = lambda v: np.ascontiguousarray(v, dtype=np.int32)
    r, c = ia(np.concatenate(I)), ia(np.concatenate(J))
    roff, coff = ia(np.arange(nb + 1) * 8), ia(np.arange(nb + 1) * 8)
    dout = hk.empty((64, nb))
    ptrs = (C.c_void_p * nb)(*[dout.ptr + 8 * 64 * b for b in range(nb)])
    ldo = ia([8] * nb)
    ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int))
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        assert L.SPX_d_struct_extract_blocks(H.h, nb, ip(r), ip(roff), ip(c), ip(coff), ptrs, ip(ldo), 0, 1) == 0
        ts.append((time.perf_counter() - t0) * 1e3)
    got = dout.get()
    for b in range(0, nb, 97):
        assert np.array_equal(got[:, b].reshape(8, 8, order="F"), out[b])
    # (reported, not asserted: a timing threshold in the correctness tier turns a loaded box into a red parity run)
    print("extract 1000 8x8 blocks at N = 1e5: %.2f ms (best of 5: %s)" % (min(ts), ["%.2f" % t for t in ts]))
    H.destroy()
    hk.close()


def test_multi_rhs_hybrid_sweeps(L):
    HC.check_multi_rhs(L, n=4000, leaf=128, nrhs_list=(5, 13, 64))
    HC.check_multi_rhs(L, n=3001, leaf=256, nrhs_list=(12, 20, 64, 100, 300))


def test_multi_rhs_matrix_core_sweeps_rank_56(L):
    """inner nodes of 112 rows and rank 56: vectors that only fit the LDS as 32-wide rows (kernels/hssk_sweep_mma.h)"""
    HC.check_multi_rhs(L, leaf=128, nrhs_list=(40, 70), A=HC.low_rank_plus_identity(2048, 56), d0=96)


def test_rccl_exchange_hook_single_rank():
    """The multi-GPU exchange hook (strumpack_amd/dist.py) on real device memory with a 1-rank RCCL
    group: zero-copy tensor views of engine buffers + in-place all_gather_into_tensor."""
    import ctypes as C
    import os
    import torch
    import torch.distributed as dist
    from strumpack_amd import dist as sdist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        L = capi.load(_loader.lib_path())
        hk = K.Hssk(_loader.lib_path())
        ld, cols = 16, 50
        a = np.arange(ld * cols, dtype=np.float64).reshape(ld, cols, order="F")
        dS = hk.array(a)
        ex = sdist.make_exchange(L, 1, 0)
        ex(None, dS.ptr, 8 * ld * cols)          # SPXAllGatherFn(user, device buffer, bytes per rank)
        assert np.array_equal(dS.get(), a)
        di = hk.array(np.arange(7, dtype=np.int32))
        ex(None, di.ptr, 4 * 7)                  # int payload of odd length: 4-byte view
        assert np.array_equal(di.get(), np.arange(7, dtype=np.int32))
        # and the sharded constructor itself with world = 1
        n = 512
        dA = hk.empty((n, n))
        hk.check(hk.lib.hssk_fill_toeplitz(hk.ctx, dA.ptr, n, n, b"T"))
        hk.sync()
        o = capi.StructuredMatrix.options(L, rel_tol=1e-6, abs_tol=1e-10, leaf_size=64)
        h = capi.StructuredMatrix.hss_options(L, d0=32, dd=16)
        H = sdist.from_dense_device(L, dA.ptr, n, n, o, h, ex, world=1, rank=0)
        assert H.is_compressed()
        x = np.ones(n)
        assert np.linalg.norm(H.mult(x)[:, 0] - O.toeplitz(n) @ x) / np.linalg.norm(x) < 1e-4
    finally:
        dist.destroy_process_group()


def test_blr_dense_slice(L):
    HC.check_blr(L)


def test_blr_32k_toeplitz_front(L):
    """BASELINE configs[4]-sized dense front: 32768 x 32768 Toeplitz, BLR compress + mult and compress-and-factor + solve at
    full size, checked through size-independent properties (products against exact rows of A, residual of the solve)."""
    n = 32768
    i = np.arange(n)
    A = np.asfortranarray(1.0 / (1.0 + np.abs(i[:, None] - i[None, :])))
    o = capi.StructuredMatrix.options(L, rel_tol=1e-6, abs_tol=1e-12, leaf_size=128, type=capi.SP_TYPE_BLR)
    B = capi.StructuredMatrix.from_dense(L, A, o)
    assert 0 < B.rank() <= 40 and B.memory() < 0.12 * 8 * n * n
    X = np.random.default_rng(9).standard_normal((n, 2))
    Y = B.mult(X)
    AX = A @ X
    assert np.linalg.norm(Y - AX) <= 1e-5 * np.linalg.norm(AX)
    B.destroy()
    F = capi.StructuredMatrix.from_dense_and_factor(L, A, o)
    Z = F.solve(X)
    assert np.linalg.norm(A @ Z - X) <= 1e-5 * np.linalg.norm(X)
    F.destroy()


def test_float_and_complex_instantiations(L):
    HC.check_scz(L)


def test_concurrent_operations_on_one_matrix(L):
    HC.check_concurrent_ops(L)


def test_host_operand_streamed_in_blocks(L):
    HC.check_host_stream_blocks(L)


def test_native_comm_and_sharded_operand_single_rank(L):
    """The library's own RCCL communicator (SPX_comm_*) with one rank on the one GPU of the test box: the collectives'
    self test, then the sharded-operand construction through both of its sketch paths (row block + column block; column
    block only, i.e. the reduce-scatter route) against the plain device construction."""
    import ctypes as C
    import numpy as np
    from strumpack_amd import _loader, dist as sdist, hssk as K
    from oracle import hss_oracle as O
    buf = C.create_string_buffer(128)
    assert L.SPX_comm_unique_id(buf) == 0
    h = C.c_void_p()
    assert L.SPX_comm_create(C.byref(h), 1, 0, buf.raw) == 0
    assert L.SPX_comm_size(h) == 1 and L.SPX_comm_rank(h) == 0
    assert L.SPX_comm_selftest(h) == 0

    class Comm:
        pass
    comm = Comm()
    comm.h = h
    hk = K.Hssk(_loader.lib_path())
    n = 3000
    A = O.toeplitz(n) + 0.01 * np.random.default_rng(3).standard_normal((n, n))
    o = capi.StructuredMatrix.options(L, rel_tol=1e-6, abs_tol=1e-10, leaf_size=128)
    ho = capi.StructuredMatrix.hss_options(L)
    lo, hi = sdist.shard_range(L, n, o, 1, 0)
    assert (lo, hi) == (0, n)
    dA = hk.array(A)
    H1 = capi.StructuredMatrix.from_dense_device(L, dA.ptr, n, n, o, ho)
    B = np.random.default_rng(4).standard_normal((n, 2))
    y1 = H1.mult(B)
    for with_rows in (True, False):
        H = sdist.from_blocks_device(L, dA.ptr if with_rows else None, n, dA.ptr, n, n, o, ho, comm=comm)
        assert np.array_equal(H.node_info(), H1.node_info())
        assert np.linalg.norm(H.mult(B) - y1) <= 1e-10 * np.linalg.norm(y1)
        H.factor()
        x = H.solve(B)
        assert np.linalg.norm(H.mult(x) - B) <= 1e-12 * np.linalg.norm(B)
        H.destroy()
    H1.destroy()
    hk.close()
    L.SPX_comm_destroy(C.byref(h))


RCCL_WORKER = r"""
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(rank)
os.environ["STRUMPACK_AMD_DEVICE"] = str(rank)
dist.init_process_group("nccl", device_id=torch.device("cuda", rank))
from strumpack_amd import _loader, capi, dist as sdist, hssk as K
from oracle import hss_oracle as O
L = capi.load(_loader.lib_path())
hk = K.Hssk(_loader.lib_path(), device=rank)
comm = sdist.NativeComm(L)
ok = L.SPX_comm_selftest(comm.h) == 0
n = 6000
A = O.toeplitz(n) + 0.01 * np.random.default_rng(3).standard_normal((n, n))
o = capi.StructuredMatrix.options(L, rel_tol=1e-6, abs_tol=1e-10, leaf_size=128)
ho = capi.StructuredMatrix.hss_options(L)
lo, hi = sdist.shard_range(L, n, o, world, rank)
dR = hk.array(np.asfortranarray(A[lo:hi, :]))
dC = hk.array(np.asfortranarray(A[:, lo:hi]))
dA = hk.array(A)
H1 = capi.StructuredMatrix.from_dense_device(L, dA.ptr, n, n, o, ho)
B = np.random.default_rng(4).standard_normal((n, 2))
y1 = H1.mult(B)
H1.factor()
x1 = H1.solve(B)
for with_rows in (True, False):
    H = sdist.from_blocks_device(L, dR.ptr if with_rows else None, hi - lo, dC.ptr, n, n, o, ho, comm=comm)
    ok = ok and np.array_equal(H.node_info(), H1.node_info())
    ok = ok and np.linalg.norm(H.mult(B) - y1) <= 1e-10 * np.linalg.norm(y1)
    H.factor()
    x = H.solve(B)
    ok = ok and np.linalg.norm(x - x1) <= 1e-8 * np.linalg.norm(x1)
    H.destroy()
Hr = sdist.from_dense_device_comm(L, dA.ptr, n, n, o, ho, comm)      # replicated operand, native collectives
ok = ok and np.linalg.norm(Hr.mult(B) - y1) <= 1e-10 * np.linalg.norm(y1)
t = torch.tensor([float(ok)], device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MIN)
if rank == 0:
    print("RCCL_OK" if t.item() == 1.0 else "RCCL_FAIL", flush=True)
dist.destroy_process_group()
"""


def test_native_rccl_two_ranks(tmp_path):
    """world_size 2 over RCCL / xGMI when the box has two GPUs (skipped on the one-GPU test boxes): sharded operand, both
    sketch paths, replicated operand, against the single-process matrix."""
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "rccl_worker.py"
    script.write_text(RCCL_WORKER % {"root": root})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29571", HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), WORLD_SIZE="2"),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert "RCCL_OK" in outs[0], "\n".join(outs)


@pytest.mark.parametrize("n,leaf,kind,d0,dd,engine", [(8192, 128, 1, 128, 64, "philox"), (4096, 256, 2, 128, 64, "linear"),
                                                       (3000, 128, 1, 64, 32, "linear")])
def test_generated_operand_fused_sketch(L, n, leaf, kind, d0, dd, engine):
    # the operand is a formula evaluated inside the sketch kernel: bitwise the compression of the stored matrix
    hk = K.Hssk(_loader.lib_path())
    HC.check_generator(L, hk, n, leaf, kind, rel_tol=1e-5, d0=d0, dd=dd, engine=engine)
    hk.close()


def test_generated_operand_beyond_hbm():
    """N = 250000: the stored matrix would be 500 GB -- more than the GPU holds; generated inside the sketch kernel it is
    compressed, factored and solved like any other (size-independent checks: residual against the compressed matrix, sampled
    entries and a sampled product against the formula)."""
    L = capi.load(_loader.lib_path())
    n = 250000
    o = capi.StructuredMatrix.options(L, rel_tol=1e-4, abs_tol=1e-8, leaf_size=256)
    H = capi.StructuredMatrix.from_generator(L, n, 1, o, capi.StructuredMatrix.hss_options(L, random_engine="philox"))
    assert H.is_compressed() and 30 <= H.rank() <= 100   # (41 at N = 1e5; 68 measured here)
    r = np.random.default_rng(9)
    b = r.standard_normal(n)
    H.factor()
    x = H.solve(b)[:, 0]
    assert np.linalg.norm(H.mult(x)[:, 0] - b) <= 1e-12 * np.linalg.norm(b)
    # rows of H x against the formula on a sample of rows: (A x)_i = sum_j x_j / (1 + |i - j|)
    rows = r.choice(n, 64, replace=False)
    jj = np.arange(n)
    Ax = np.array([np.dot(1.0 / (1.0 + np.abs(i - jj)), x) for i in rows])
    assert np.linalg.norm(Ax - b[rows]) <= 5e-3 * np.linalg.norm(b[rows])
    I = [np.sort(r.choice(n, 8, replace=False)) for _ in range(50)]
    J = [np.sort(r.choice(n, 8, replace=False)) for _ in range(50)]
    blocks = H.extract_blocks(I, J)
    for bi, (ii, jc) in enumerate(zip(I, J)):
        ref = 1.0 / (1.0 + np.abs(ii[:, None] - jc[None, :]))
        assert np.abs(blocks[bi] - ref).max() <= 1e-3
    H.destroy()


def test_factor_ahead_of_the_compression(L):
    HC.check_factor_ahead(L, n=6000, leaf=128)


def test_inner_levels_in_one_launch(L):
    """hssk_tree_inner on the MI355X (workgroups of one launch polling each other across XCDs) against the level-synchronous
    path, BASELINE configs[1]'s shape included (N = 32768, leaf 256, rel_tol 1e-4: 8 levels, 254 workgroups)."""
    from strumpack_amd import hssk as K
    hk = K.Hssk(_loader.lib_path())
    HC.check_tree_pass(L, hk, sizes=((3000, 64, 1e-6), (6000, 128, 1e-6), (1100, 32, 1e-6), (515, 64, 1e-6), (32768, 256, 1e-4)), again=True)
    hk.close()


def test_symmetric_operand_hint(L):
    hk = K.Hssk(_loader.lib_path())
    HC.check_symmetric_hint(L, hk, n=8192, leaf=128)
    hk.close()


def test_inner_levels_in_one_launch_at_full_size(L, monkeypatch):
    """BASELINE configs[2]'s tree (N = 100000, leaf 256, rel_tol 1e-4, 192 samples: 1023 nodes, 9 inner levels, 1021 workgroups
    of one launch) with the operand given by the library's Toeplitz formula (nothing stored): the single-launch tree pass against
    the level-synchronous path on the same samples -- every node's rows and ranks, memory to the byte, products and solutions to
    rounding --, and the reference's pass criteria on sampled columns."""
    n = 100000
    o = capi.StructuredMatrix.options(L, rel_tol=1e-4, abs_tol=1e-8, leaf_size=256)
    h = capi.StructuredMatrix.hss_options(L, random_engine="philox")
    rng = np.random.default_rng(3)
    b = rng.standard_normal((n, 2))
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("STRUMPACK_AMD_TREE_LAUNCH", mode)
        l0 = L.SPX_tree_pass_launches()
        H = capi.StructuredMatrix.from_generator(L, n, 1, o, h)
        assert H.is_compressed() and L.SPX_tree_pass_launches() - l0 == (1 if mode == "1" else 0)
        y = H.mult(b)
        H.factor()
        x = H.solve(b)
        assert np.linalg.norm(H.mult(x) - b) <= HC.SOLVE_TOLERANCE * np.linalg.norm(b)
        res[mode] = (H.node_info(), H.memory(), H.rank(), H.levels(), y, x)
        H.destroy()
    assert np.array_equal(res["1"][0], res["0"][0]) and res["1"][1:4] == res["0"][1:4]
    assert res["1"][2] == 41 and res["1"][3] == 10
    for k in (4, 5):
        assert np.linalg.norm(res["1"][k] - res["0"][k]) <= 1e-10 * np.linalg.norm(res["0"][k])
    cols = rng.integers(0, n, 16)
    i = np.arange(n)
    Ac = 1.0 / (1.0 + np.abs(i[:, None] - cols[None, :]))
    # y = H b against A b on sampled ROWS (A symmetric): rows cols of A b
    assert np.linalg.norm(Ac.T @ b - res["1"][4][cols]) <= 1e2 * 1e-4 * np.linalg.norm(Ac.T @ b)
