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
dist.barrier()
say("DIST_SMOKE_OK" if res < 1e-12 else "DIST_SMOKE_FAIL")
dist.destroy_process_group()
