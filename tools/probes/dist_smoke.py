"""Bounded functional check of the multi-rank path on real device memory (several ranks may share one
GPU with STRUMPACK_AMD_SHARE_GPU=1 and the gloo backend).  Prints progress so a hang is localised."""
import os
import sys
import time

sys.path.insert(0, ".")
import numpy as np
import torch
import torch.distributed as dist

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
local = 0 if os.environ.get("STRUMPACK_AMD_SHARE_GPU") else int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local)
os.environ["STRUMPACK_AMD_DEVICE"] = str(local)
backend = os.environ.get("STRUMPACK_AMD_BACKEND", "nccl")
dist.init_process_group(backend, **({"device_id": torch.device("cuda", local)} if backend == "nccl" else {}))
from strumpack_amd import _loader, capi, dist as sdist, hssk as K  # noqa: E402


def say(*a):
    print("[rank %d %.2fs]" % (rank, time.time() - T0), *a, flush=True)


T0 = time.time()
L = capi.load(_loader.lib_path())
hk = K.Hssk(_loader.lib_path(), device=local)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dA = hk.empty((n, n))
hk.check(hk.lib.hssk_fill_toeplitz(hk.ctx, dA.ptr, n, n, b"T"))
hk.sync()
say("A filled")
o = capi.StructuredMatrix.options(L, rel_tol=1e-4, abs_tol=1e-8, leaf_size=256)
h = capi.StructuredMatrix.hss_options(L, random_engine="philox")
ex = sdist.make_exchange(L, world, rank)
H = sdist.from_dense_device(L, dA.ptr, n, n, o, h, ex)
say("compressed", H.is_compressed(), "rank", H.rank())
H.factor()
say("factored")
b = np.random.default_rng(1).standard_normal((n, 2))
x = H.solve(b)
say("solved")
res = np.linalg.norm(H.mult(x) - b) / np.linalg.norm(b)
say("residual %.2e" % res)
H1 = capi.StructuredMatrix.from_dense_device(L, dA.ptr, n, n, o, h)
H1.factor()
x1 = H1.solve(b)
say("vs single-process: ranks equal", np.array_equal(H.node_info(), H1.node_info()), "dx %.2e" % (np.linalg.norm(x - x1) / np.linalg.norm(x1)))
# kernel-matrix front end across the ranks (balanced kd tree -> subtree ownership)
nk = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
X = np.random.default_rng(3).random((nk, 8))
ok_o = capi.StructuredMatrix.options(L, rel_tol=1e-2, abs_tol=1e-8, leaf_size=256)
t1 = time.time()
Hk, Xp, perm = sdist.from_kernel(L, X, ok_o, kernel="Gauss", h=1.3, lam=3.11, clustering="kdtree", neighbors=64, exchange_cb=ex)
say("kernel HSS compressed", Hk.is_compressed(), "rank", Hk.rank(), "in %.3f s" % (time.time() - t1))
Hk.factor()
yk = np.random.default_rng(4).standard_normal((nk, 1))
wk = Hk.solve(yk)
resk = np.linalg.norm(Hk.mult(wk) - yk) / np.linalg.norm(yk)
Hk1, _, perm1 = sdist.from_kernel(L, X, ok_o, kernel="Gauss", h=1.3, lam=3.11, clustering="kdtree", neighbors=64)
Hk1.factor()
wk1 = Hk1.solve(yk)
say("kernel vs single-process: ranks equal", np.array_equal(Hk.node_info(), Hk1.node_info()), "perm equal", np.array_equal(perm, perm1),
    "dw %.2e" % (np.linalg.norm(wk - wk1) / np.linalg.norm(wk1)), "residual %.2e" % resk)
dist.barrier()
say("DIST_SMOKE_OK" if res < 1e-12 and resk < 1e-10 else "DIST_SMOKE_FAIL")
dist.destroy_process_group()
