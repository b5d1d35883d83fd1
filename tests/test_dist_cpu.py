"""N > 1 path on CPU: several processes (gloo) run the subtree-distributed construction / factor /
solve / mult on the emulator build and must reproduce the single-process matrix (same ranks on every
node, same products and solutions to rounding).  world = 2, 4: subtree ownership below the cut;
world = 3: fallback (sharded sketch, replicated tree)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np, torch, torch.distributed as dist
import emu_lib
from strumpack_amd import capi, dist as sdist, hssk as K
from oracle import hss_oracle as O
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
L = capi.load(emu_lib.PATH)
hk = K.Hssk(emu_lib.PATH)
ok = True
# (n, leaf, d0, dd, algorithm, sketch): "sjlt" = the SJLT sketch through the streaming kernels on each rank's rows / columns
CASES = {2: [(120, 16, 16, 8, "stable", "gaussian"), (90, 16, 16, 8, "original", "sjlt")],
         4: [(140, 16, 8, 8, "stable", "gaussian")], 3: [(90, 16, 16, 8, "stable", "gaussian"), (90, 16, 16, 8, "stable", "sjlt")]}
for (n, leaf, d0, dd, algo, sketch) in CASES[world]:
    A = O.toeplitz(n)
    dA = hk.array(A)
    o = capi.StructuredMatrix.options(L, rel_tol=1e-6, abs_tol=1e-10, leaf_size=leaf)
    h = capi.StructuredMatrix.hss_options(L, d0=d0, dd=dd, algorithm=algo, sketch=sketch, nnz0=3, nnz=2)
    ex = sdist.make_exchange(L, world, rank)
    H = sdist.from_dense_device(L, dA.ptr, n, n, o, h, ex)
    H1 = capi.StructuredMatrix.from_dense_device(L, dA.ptr, n, n, o, h)   # single-process reference
    rng = np.random.default_rng(5)
    B = rng.standard_normal((n, 3))
    same_tree = np.array_equal(H.node_info(), H1.node_info())
    y, y1 = H.mult(B), H1.mult(B)
    yt, yt1 = H.mult(B, "T"), H1.mult(B, "T")
    H.factor(); H1.factor()
    x, x1 = H.solve(B), H1.solve(B)
    e_mult = np.linalg.norm(y - y1) / np.linalg.norm(y1)
    e_multT = np.linalg.norm(yt - yt1) / np.linalg.norm(yt1)
    e_solve = np.linalg.norm(x - x1) / np.linalg.norm(x1)
    res = np.linalg.norm(H.mult(x) - B) / np.linalg.norm(B)
    # twenty right-hand sides: the matrix-core forms of the sweeps on the rank's subtree and on the replicated top
    B20 = rng.standard_normal((n, 20))
    e_mult = max(e_mult, np.linalg.norm(H.mult(B20) - H1.mult(B20)) / np.linalg.norm(B20))
    e_solve = max(e_solve, np.linalg.norm(H.solve(B20) - H1.solve(B20)) / np.linalg.norm(B20))
    good = same_tree and e_mult < 1e-11 and e_multT < 1e-11 and e_solve < 1e-9 and res < 1e-12 and H.stats()["rounds"] == H1.stats()["rounds"]
    if not good:
        print("rank", rank, "case", n, leaf, algo, same_tree, e_mult, e_multT, e_solve, res, flush=True)
    ok = ok and good
    H.destroy(); H1.destroy()
# sharded operand: every rank passes only its row block + column block, or only its column block (column-sharded operator:
# the Sr contributions are reduced over the ranks); must reproduce the single-process matrix built from the whole A
# (the cases with 64 + 32 samples take the single-launch tree pass on the rank's own levels -- coupling blocks read from the
#  rank's column block with global indices --, the others, with too few samples for its rank bound, the level path)
BCASES = {2: [(120, 16, 16, 8, "stable", True), (120, 16, 16, 8, "stable", False), (75, 8, 8, 8, "original", False), (300, 32, 64, 32, "stable", True)],
          4: [(140, 16, 8, 8, "stable", True), (140, 16, 8, 8, "stable", False), (600, 32, 64, 32, "stable", False)], 3: []}
for (n, leaf, d0, dd, algo, with_rows) in BCASES[world]:
    A = O.toeplitz(n) + 0.01 * np.random.default_rng(2).standard_normal((n, n))     # unsymmetric: rows and columns differ
    if d0 + dd >= 96:   # (compressible in one round: an unsymmetric perturbation of rank 3 instead of full-rank noise)
        g2 = np.random.default_rng(2)
        A = O.toeplitz(n) + 0.05 * g2.standard_normal((n, 3)) @ g2.standard_normal((3, n))
    o = capi.StructuredMatrix.options(L, rel_tol=1e-6, abs_tol=1e-10, leaf_size=leaf)
    # (factor_ahead on the rank's own levels when the row block is given: the cut exchange and the replicated top follow in factor())
    h = capi.StructuredMatrix.hss_options(L, d0=d0, dd=dd, algorithm=algo, factor_ahead=with_rows)
    lo, hi = sdist.shard_range(L, n, o, world, rank)
    dR = hk.array(np.asfortranarray(A[lo:hi, :])) if with_rows else None
    dC = hk.array(np.asfortranarray(A[:, lo:hi]))
    ex = sdist.make_exchange(L, world, rank)
    tl0 = L.SPX_tree_pass_launches()
    H = sdist.from_blocks_device(L, dR.ptr if with_rows else None, hi - lo, dC.ptr, n, n, o, h, exchange_cb=ex)
    if d0 + dd >= 96 and L.SPX_tree_pass_launches() != tl0 + 1:
        print("rank", rank, "block case", n, "did not take the single-launch tree pass on its own levels", flush=True)
        ok = False
    dA = hk.array(A)
    H1 = capi.StructuredMatrix.from_dense_device(L, dA.ptr, n, n, o, h)
    B = np.random.default_rng(5).standard_normal((n, 2))
    same_tree = np.array_equal(H.node_info(), H1.node_info())
    y, y1 = H.mult(B), H1.mult(B)
    yt, yt1 = H.mult(B, "T"), H1.mult(B, "T")
    H.factor(); H1.factor()
    x, x1 = H.solve(B), H1.solve(B)
    e_mult = np.linalg.norm(y - y1) / np.linalg.norm(y1)
    e_multT = np.linalg.norm(yt - yt1) / np.linalg.norm(yt1)
    e_solve = np.linalg.norm(x - x1) / np.linalg.norm(x1)
    good = same_tree and e_mult < 1e-10 and e_multT < 1e-10 and e_solve < 1e-8
    if not good:
        print("rank", rank, "block case", n, leaf, algo, with_rows, same_tree, e_mult, e_multT, e_solve, flush=True)
    ok = ok and good
    H.destroy(); H1.destroy()
# generated operand (the library's Toeplitz formula evaluated inside the sketch kernel): no rank holds any part of the matrix;
# n = 256 with 64 + 64 samples takes the fused kernel on every rank's 128-column range, the others the written-out blocks
GCASES = {2: [(256, 32, 64, 64, 1), (120, 16, 16, 8, 2)], 4: [(512, 32, 64, 64, 1), (140, 16, 8, 8, 1)], 3: [(90, 16, 16, 8, 1)]}
for (n, leaf, d0, dd, kind) in GCASES[world]:
    A = O.toeplitz(n) if kind == 1 else np.triu(O.toeplitz(n))
    dA = hk.array(A)
    o = capi.StructuredMatrix.options(L, rel_tol=1e-6, abs_tol=1e-10, leaf_size=leaf)
    h = capi.StructuredMatrix.hss_options(L, d0=d0, dd=dd)
    ex = sdist.make_exchange(L, world, rank)
    H = sdist.from_generator(L, n, kind, o, h, exchange_cb=ex)
    H1 = capi.StructuredMatrix.from_dense_device(L, dA.ptr, n, n, o, h)
    B = np.random.default_rng(5).standard_normal((n, 2))
    same_tree = np.array_equal(H.node_info(), H1.node_info())
    y, y1 = H.mult(B), H1.mult(B)
    H.factor(); H1.factor()
    x, x1 = H.solve(B), H1.solve(B)
    e_mult = np.linalg.norm(y - y1) / np.linalg.norm(y1)
    e_solve = np.linalg.norm(x - x1) / np.linalg.norm(x1)
    good = same_tree and e_mult < 1e-11 and e_solve < 1e-9
    if not good:
        print("rank", rank, "generator case", n, leaf, kind, same_tree, e_mult, e_solve, flush=True)
    ok = ok and good
    H.destroy(); H1.destroy()
# kernel-matrix front end: subtree ownership (natural / kd trees are balanced -> cut exists), replicated otherwise
KCASES = {2: [(100, 16, "kdtree", "Gauss")], 4: [(100, 16, "natural", "Laplace")], 3: []}
for (n, leaf, clus, kern) in KCASES[world]:
    rng = np.random.default_rng(11)
    X = rng.standard_normal((n, 4))
    o = capi.StructuredMatrix.options(L, rel_tol=1e-4, abs_tol=1e-10, leaf_size=leaf)
    ex = sdist.make_exchange(L, world, rank)
    H, Xp, perm = sdist.from_kernel(L, X, o, kernel=kern, h=1.5, lam=2.0, clustering=clus, neighbors=32, exchange_cb=ex)
    H1, Xp1, perm1 = sdist.from_kernel(L, X, o, kernel=kern, h=1.5, lam=2.0, clustering=clus, neighbors=32)
    B = rng.standard_normal((n, 2))
    same_tree = np.array_equal(H.node_info(), H1.node_info()) and np.array_equal(perm, perm1)
    y, y1 = H.mult(B), H1.mult(B)
    H.factor(); H1.factor()
    x, x1 = H.solve(B), H1.solve(B)
    e_mult = np.linalg.norm(y - y1) / np.linalg.norm(y1)
    e_solve = np.linalg.norm(x - x1) / np.linalg.norm(x1)
    good = same_tree and e_mult < 1e-11 and e_solve < 1e-9 and H.is_compressed()
    if not good:
        print("rank", rank, "kernel case", n, leaf, clus, same_tree, e_mult, e_solve, flush=True)
    ok = ok and good
    H.destroy(); H1.destroy()
t = torch.tensor([float(ok)])
dist.all_reduce(t, op=dist.ReduceOp.MIN)
if rank == 0:
    print("DIST_OK" if t.item() == 1.0 else "DIST_FAIL", flush=True)
dist.destroy_process_group()
'''


@pytest.mark.parametrize("world", [2, 4, 3])
def test_distributed_hss(tmp_path, world):
    import emu_lib
    emu_lib.build()
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29533 + world), HSSK_EMU_THREADS="2")
    procs = []
    for r in range(world):
        e = dict(env, RANK=str(r), WORLD_SIZE=str(world))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=e, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=900)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert "DIST_OK" in outs[0], "\n".join(outs)


NATIVE_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np, torch, torch.distributed as dist
import emu_lib
from strumpack_amd import capi, dist as sdist, hssk as K
from oracle import hss_oracle as O
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)     # (only to hand rank 0's unique id to the others)
L = capi.load(emu_lib.PATH)
hk = K.Hssk(emu_lib.PATH)
comm = sdist.NativeComm(L)                                       # csrc/host/Comm.cpp: RcclComm over the library STRUMPACK_AMD_RCCL_LIB names
ok = L.SPX_comm_selftest(comm.h) == 0 and L.SPX_comm_size(comm.h) == world and L.SPX_comm_rank(comm.h) == rank
if not ok:
    print("rank", rank, "selftest failed", flush=True)
# sharded operand through the native communicator: row block + column block (all-gathers at the cut, all-reduce of the top
# nodes' coupling blocks), column block only (the Sr contributions reduced to their owners: the grouped ncclReduce)
CASES = {2: [(120, 16, 16, 8, "stable", True), (120, 16, 16, 8, "stable", False), (300, 32, 64, 32, "stable", True)],
         4: [(140, 16, 8, 8, "stable", True), (140, 16, 8, 8, "stable", False), (600, 32, 64, 32, "stable", False)]}
for (n, leaf, d0, dd, algo, with_rows) in CASES[world]:
    A = O.toeplitz(n) + 0.01 * np.random.default_rng(2).standard_normal((n, n))
    if d0 + dd >= 96:
        g2 = np.random.default_rng(2)
        A = O.toeplitz(n) + 0.05 * g2.standard_normal((n, 3)) @ g2.standard_normal((3, n))
    o = capi.StructuredMatrix.options(L, rel_tol=1e-6, abs_tol=1e-10, leaf_size=leaf)
    h = capi.StructuredMatrix.hss_options(L, d0=d0, dd=dd, algorithm=algo, factor_ahead=with_rows)
    lo, hi = sdist.shard_range(L, n, o, world, rank)
    dR = hk.array(np.asfortranarray(A[lo:hi, :])) if with_rows else None
    dC = hk.array(np.asfortranarray(A[:, lo:hi]))
    H = sdist.from_blocks_device(L, dR.ptr if with_rows else None, hi - lo, dC.ptr, n, n, o, h, comm=comm)
    dA = hk.array(A)
    H1 = capi.StructuredMatrix.from_dense_device(L, dA.ptr, n, n, o, h)
    B = np.random.default_rng(5).standard_normal((n, 2))
    same_tree = np.array_equal(H.node_info(), H1.node_info())
    y, y1 = H.mult(B), H1.mult(B)
    yt, yt1 = H.mult(B, "T"), H1.mult(B, "T")
    H.factor(); H1.factor()
    x, x1 = H.solve(B), H1.solve(B)
    e_mult = np.linalg.norm(y - y1) / np.linalg.norm(y1)
    e_multT = np.linalg.norm(yt - yt1) / np.linalg.norm(yt1)
    e_solve = np.linalg.norm(x - x1) / np.linalg.norm(x1)
    good = same_tree and e_mult < 1e-10 and e_multT < 1e-10 and e_solve < 1e-8
    if not good:
        print("rank", rank, "native block case", n, leaf, with_rows, same_tree, e_mult, e_multT, e_solve, flush=True)
    ok = ok and good
    H.destroy(); H1.destroy()
# generated operand, twenty right-hand sides as well
for (n, leaf, d0, dd) in {2: [(256, 32, 64, 64)], 4: [(512, 32, 64, 64)]}[world]:
    A = O.toeplitz(n)
    dA = hk.array(A)
    o = capi.StructuredMatrix.options(L, rel_tol=1e-6, abs_tol=1e-10, leaf_size=leaf)
    h = capi.StructuredMatrix.hss_options(L, d0=d0, dd=dd)
    H = sdist.from_generator(L, n, 1, o, h, comm=comm)
    H1 = capi.StructuredMatrix.from_dense_device(L, dA.ptr, n, n, o, h)
    B = np.random.default_rng(5).standard_normal((n, 20))
    same_tree = np.array_equal(H.node_info(), H1.node_info())
    y, y1 = H.mult(B), H1.mult(B)
    H.factor(); H1.factor()
    x, x1 = H.solve(B), H1.solve(B)
    good = same_tree and np.linalg.norm(y - y1) < 1e-11 * np.linalg.norm(y1) and np.linalg.norm(x - x1) < 1e-9 * np.linalg.norm(x1)
    if not good:
        print("rank", rank, "native generator case", n, same_tree, flush=True)
    ok = ok and good
    H.destroy(); H1.destroy()
comm.close()
t = torch.tensor([float(ok)])
dist.all_reduce(t, op=dist.ReduceOp.MIN)
if rank == 0:
    print("NATIVE_OK" if t.item() == 1.0 else "NATIVE_FAIL", flush=True)
dist.destroy_process_group()
'''


def fake_rccl():
    """tests/emu/fake_rccl.cpp: the nccl* symbols Comm.cpp binds, over shared memory (built with the emulator library)"""
    import emu_lib
    emu_lib.build()
    path = os.path.join(os.path.dirname(emu_lib.PATH), "libfake_rccl.so")
    assert os.path.exists(path)
    return path


@pytest.mark.parametrize("world", [2, 4])
def test_native_comm_multirank(tmp_path, world):
    """RcclComm (csrc/host/Comm.cpp) with N > 1 ranks: the in-place all-gather offsets, the all-reduce and the grouped reduce that
    serves as reduce-scatter, driven by the engine exactly as on the GPUs (SPX_comm_*, from_blocks_device / from_generator with a
    native communicator), against the single-process matrix.  The collectives run over a stand-in for librccl."""
    script = tmp_path / "native_worker.py"
    script.write_text(NATIVE_WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29543 + world), HSSK_EMU_THREADS="2",
               STRUMPACK_AMD_RCCL_LIB=fake_rccl())
    procs = []
    for r in range(world):
        e = dict(env, RANK=str(r), WORLD_SIZE=str(world))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=e, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=900)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert "NATIVE_OK" in outs[0], "\n".join(outs)


@pytest.mark.parametrize("world", [2])
def test_bench_dry_run_native_comm(world):
    """The multi-GPU bench line's own path -- STRUMPACK_AMD_BENCH_COMM=rccl: native communicator, trial run, sharded operand,
    no fall-back -- on the emulator with the librccl stand-in: the line must say it ran on the native communicator with
    `world` ranks."""
    import json
    import emu_lib
    env = dict(os.environ, STRUMPACK_AMD_RCCL_LIB=fake_rccl(), STRUMPACK_AMD_BENCH_DRYRUN_LIB=emu_lib.PATH, HSSK_EMU_THREADS="2",
               OMP_NUM_THREADS="2", STRUMPACK_AMD_BENCH_COMM="rccl")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(29581 + world), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "1", "--warmup", "0",
           "--size", "800", "--leaf", "64", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1200, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["dry_run"] is True and d["n_gpus"] == world and d["config"]["rccl_nranks"] == world
    assert d["config"]["comm"].startswith("native RCCL communicator") and d["checks"]["solve_resid_H"] < 1e-10


@pytest.mark.parametrize("world", [2])
def test_bench_rendezvous_dry_run(world):
    """bench.py launched exactly as the driver launches its multi-GPU run (python -m torch.distributed.run --nnodes=1
    --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W), on the emulator build
    with gloo (STRUMPACK_AMD_BENCH_DRYRUN_LIB): argument handling, rendezvous, the native-communicator trial and its agreed
    fall-back, operand sharding, the timed loop's barriers and the max-over-ranks reduction, one JSON line from rank 0."""
    import json
    import emu_lib
    emu_lib.build()
    env = dict(os.environ, STRUMPACK_AMD_BENCH_DRYRUN_LIB=emu_lib.PATH, HSSK_EMU_THREADS="2", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(29571 + world), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "1", "--warmup", "1",
           "--size", "1500", "--leaf", "64", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1200, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["dry_run"] is True and d["value"] is None and d["n_gpus"] == world and d["steps"] == 1 and d["warmup"] == 1
    assert d["metric"] == "hss_compress_ulv_factor_solve_gflops" and d["config"]["n"] == 1500
    assert d["checks"]["solve_resid_H"] < 1e-10 and d["hss"]["levels"] >= 3


def test_bench_bare_gpus_spelling_spawns_the_ranks():
    """`python bench.py --gpus 2 ...` with no launcher (the shape of the driver's one-GPU command with N = 2): bench.py starts the
    two ranks itself and relays rank 0's line -- n_gpus is 2, not a single rank that ignored the flag.  A launcher whose
    WORLD_SIZE disagrees with --gpus is refused (exit code 2, no line)."""
    import json
    import emu_lib
    emu_lib.build()
    env = dict(os.environ, STRUMPACK_AMD_BENCH_DRYRUN_LIB=emu_lib.PATH, HSSK_EMU_THREADS="2", OMP_NUM_THREADS="2")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
           "--size", "1500", "--leaf", "64", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1200, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["dry_run"] is True and d["checks"]["solve_resid_H"] < 1e-10
    # the native communicator cannot be set up on the emulator: with the default (rccl, no fall-back) every rank exits non-zero,
    # and so does the parent -- no line
    r = subprocess.run(cmd, env=dict(env, STRUMPACK_AMD_BENCH_COMM="rccl"), capture_output=True, text=True, timeout=1200, cwd=ROOT)
    assert r.returncode != 0 and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")], r.stdout
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--no-cpu-baseline"],
                       env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 2 and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")], r.stdout + r.stderr


def test_bench_refuses_silent_fallback():
    """With the default process group (the library's RCCL communicator) a multi-rank bench run must not fall back to the torch
    callback path on its own: where RCCL cannot be set up (here: the emulator build, gloo) every rank exits non-zero and no
    line is printed -- a SCALE run can then never report a fall-back as RCCL."""
    import emu_lib
    emu_lib.build()
    env = dict(os.environ, STRUMPACK_AMD_BENCH_DRYRUN_LIB=emu_lib.PATH, HSSK_EMU_THREADS="2", OMP_NUM_THREADS="2",
               STRUMPACK_AMD_BENCH_COMM="rccl")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29577", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
           "--size", "1500", "--leaf", "64", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1200, cwd=ROOT)
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")], r.stdout
    assert "refusing to fall back" in r.stderr + r.stdout
