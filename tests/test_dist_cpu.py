"""N > 1 path on CPU: two processes (gloo, world_size 2) run the sharded-sketch construction on the
emulator build and must reproduce the single-process matrix bit-for-bit."""
import os
import subprocess
import sys

import numpy as np

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
n = 203                                   # odd: the last shard is ragged
A = O.toeplitz(n)
dA = hk.array(A)
o = capi.StructuredMatrix.options(L, rel_tol=1e-6, abs_tol=1e-10, leaf_size=32)
h = capi.StructuredMatrix.hss_options(L, d0=16, dd=8)
ex = sdist.make_exchange(L, world, rank)
H = sdist.from_dense_device(L, dA.ptr, n, n, o, h, ex)
H1 = capi.StructuredMatrix.from_dense_device(L, dA.ptr, n, n, o, h)   # unsharded, same process
b = np.linspace(-1, 1, n)
same = np.array_equal(H.node_info(), H1.node_info()) and np.array_equal(H.mult(b), H1.mult(b))
H.factor(); x = H.solve(b)
res = np.linalg.norm(H.mult(x) - b.reshape(-1, 1)) / np.linalg.norm(b)
t = torch.tensor([float(same), float(res < 1e-12)])
dist.all_reduce(t, op=dist.ReduceOp.MIN)
if rank == 0:
    print("DIST_OK" if t.min().item() == 1.0 else "DIST_FAIL", H.rank(), res)
dist.destroy_process_group()
'''


def test_sharded_sketch_two_ranks(tmp_path):
    import emu_lib
    emu_lib.build()
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", HSSK_EMU_THREADS="2")
    procs = []
    for r in range(2):
        e = dict(env, RANK=str(r), WORLD_SIZE="2")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=e, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert "DIST_OK" in outs[0], outs[0]
