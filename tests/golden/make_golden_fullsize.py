"""Fixtures at BASELINE.json's FULL sizes, produced by RUNNING THE REFERENCE ITSELF (oracle/_ref):
  configs[1]  T(32768), leaf 256, rel_tol 1e-4   -- HSSMatrix(A, opts) on the dense matrix (8.6 GB)
  configs[2]  T(100000), leaf 256, rel_tol 1e-4  -- HSSMatrix(n, n, opts) + compress(Amult, Aelem, opts) with an O(N^2 d)
              Toeplitz multiplication that never stores the matrix (ref_driver.cpp: ref_hss_create_toeplitz_matfree; the
              route of structured::construct_partially_matrix_free, structured/StructuredMatrix.cpp:637-651) -- the same
              random stream, hence the same compression as the 80 GB dense route
  + both at the reference's default leaf size 512.
Stored per case: options, levels, rank, memory, nonzeros, the pre-order node table (row offset, rows, U rows, U rank, V rank,
leaf), ||H b||, ||H^T b||, ||x|| and the ULV residual for b = the reference's default random vector, and H's error on 64 fixed
columns.  Run in the build container only (about 10 minutes on 8 cores) -> tests/golden/hss_fullsize_golden.json"""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import ref_lib as R  # noqa: E402

CASES = [("config2_T32768_leaf256", 32768, 256, 1e-4, "dense"), ("config3_T100000_leaf256", 100000, 256, 1e-4, "matfree"),
         ("config2_T32768_leaf512", 32768, 512, 1e-4, "matfree"), ("config3_T100000_leaf512", 100000, 512, 1e-4, "matfree")]
NCOLS = 64


def columns(n):
    return np.random.default_rng(0).integers(0, n, NCOLS)     # (bench.py samples the same columns)


def run(name, n, leaf, rtol, route):
    t0 = time.time()
    kw = dict(rel_tol=rtol, abs_tol=1e-8, leaf=leaf, d0=128, dd=64)
    H = R.RefHSS(R.test_matrix("T", n), **kw) if route == "dense" else R.RefHSS.toeplitz_matfree(n, **kw)
    t1 = time.time()
    out = dict(name=name, problem="T", n=n, leaf_size=leaf, rel_tol=rtol, abs_tol=1e-8, algorithm="stable", d0=128, dd=64, p=10,
               max_rank=50000, route=route)
    out.update(compressed=H.is_compressed(), levels=H.levels(), rank=H.rank(), memory=H.memory(), nonzeros=H.nonzeros(),
               nodes=H.node_info().tolist())
    b = R.randn(n)
    out["mult_b_norm"] = float(np.linalg.norm(H.mult(b)))
    out["multT_b_norm"] = float(np.linalg.norm(H.mult(b, "T")))
    cols = columns(n)
    E = np.zeros((n, NCOLS), order="F")
    E[cols, np.arange(NCOLS)] = 1.0
    i = np.arange(n)
    Ac = np.where(i[:, None] == cols[None, :], 1.0, 1.0 / (1.0 + np.abs(i[:, None] - cols[None, :])))
    out["rel_err_sampled"] = float(np.linalg.norm(H.mult(E) - Ac) / np.linalg.norm(Ac))
    H.factor()
    x = H.solve(b)[:, 0]
    out["solve_resid_H"] = float(np.linalg.norm(H.mult(x)[:, 0] - b) / np.linalg.norm(b))
    out["x_norm"] = float(np.linalg.norm(x))
    out["x_head"] = x[:64].tolist()
    print("%-26s compress %.1fs total %.1fs  levels %d rank %d memory %.2f MB  err %.2e  resid %.2e" %
          (name, t1 - t0, time.time() - t0, out["levels"], out["rank"], out["memory"] / 1e6, out["rel_err_sampled"], out["solve_resid_H"]), flush=True)
    return out


if __name__ == "__main__":
    cases = [run(*c) for c in CASES]
    with open(os.path.join(HERE, "hss_fullsize_golden.json"), "w") as f:
        json.dump(dict(generator="tests/golden/make_golden_fullsize.py (reference v8.0.0 via oracle/_ref, MKL, %d threads)" % (os.cpu_count() or 0),
                       cases=cases), f)
