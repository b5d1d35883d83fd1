"""Generates tests/golden/hss_seq_golden.json by RUNNING THE REFERENCE ITSELF (oracle/_ref, built by
oracle/ref/Makefile from /root/reference) on the parameter sweep of the reference's own CTest
definitions for test_HSS_seq (test/CMakeLists.txt:57-146, cases 1-22; 23-26 are the SJLT sketch,
out of scope) plus BASELINE.json configs 1 and 2.  Only data (inputs are generated, outputs are
numbers) is stored -- no reference source.  Run in the build container only:

    make -C oracle/ref && python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import ref_lib as R  # noqa: E402

# (name, problem, n, leaf, rel_tol, abs_tol, algorithm, d0, dd)  -- test/CMakeLists.txt:57-143
CTEST = [
    ("HSS_seq_1", "L", 10, 3, 1e-5, 1e-10, "stable", 32, 4),
    ("HSS_seq_2", "L", 200, 128, 1e-1, 1e-10, "original", 128, 8),
    ("HSS_seq_3", "L", 1, 128, 1, 1e-13, "original", 16, 4),
    ("HSS_seq_4", "T", 200, 1, 1e-10, 1e-10, "stable", 64, 4),
    ("HSS_seq_5", "T", 500, 128, 1e-10, 1e-13, "stable", 16, 4),
    ("HSS_seq_6", "U", 1, 1, 1e-10, 1e-13, "original", 32, 4),
    ("HSS_seq_7", "U", 500, 1, 1, 1e-10, "stable", 64, 8),
    ("HSS_seq_8", "T", 200, 128, 1, 1e-10, "stable", 128, 8),
    ("HSS_seq_9", "L", 10, 16, 1, 1e-10, "stable", 64, 4),
    ("HSS_seq_10", "L", 1, 1, 1, 1e-13, "original", 64, 4),
    ("HSS_seq_11", "U", 200, 16, 1e-1, 1e-10, "stable", 128, 4),
    ("HSS_seq_12", "L", 500, 16, 1e-10, 1e-10, "original", 128, 8),
    ("HSS_seq_13", "U", 500, 128, 1e-1, 1e-10, "stable", 32, 4),
    ("HSS_seq_14", "U", 200, 16, 1e-10, 1e-10, "stable", 128, 8),
    ("HSS_seq_15", "U", 1, 16, 1e-10, 1e-10, "stable", 128, 4),
    ("HSS_seq_16", "T", 500, 16, 1, 1e-13, "original", 64, 8),
    ("HSS_seq_17", "T", 10, 16, 1e-10, 1e-10, "original", 128, 4),
    ("HSS_seq_18", "L", 200, 1, 1e-5, 1e-13, "original", 32, 8),
    ("HSS_seq_19", "T", 500, 1, 1, 1e-10, "original", 32, 4),
    ("HSS_seq_20", "U", 200, 16, 1, 1e-13, "stable", 16, 8),
    ("HSS_seq_21", "T", 500, 1, 1, 1e-10, "stable", 64, 4),
    ("HSS_seq_22", "T", 1000, 32, 1e-5, 1e-10, "stable", 8, 8),
    # BASELINE.json configs[0] (HSSOptions defaults) and a reduced-size configs[1] shape
    ("config1_T4096_defaults", "T", 4096, 512, 1e-2, 1e-8, "stable", 128, 64),
    ("config2shape_T8192_leaf256_rtol1e-4", "T", 8192, 256, 1e-4, 1e-8, "stable", 128, 64),
]


def run_case(name, prob, n, leaf, rtol, atol, algo, d0, dd, full_dense=True):
    A = R.test_matrix(prob, n)
    H = R.RefHSS(A, rel_tol=rtol, abs_tol=atol, leaf=leaf, d0=d0, dd=dd, algo=algo)
    out = dict(name=name, problem=prob, n=n, leaf_size=leaf, rel_tol=rtol, abs_tol=atol,
               algorithm=algo, d0=d0, dd=dd, p=10, max_rank=50000)
    out["compressed"] = H.is_compressed()
    out["levels"] = H.levels()
    out["rank"] = H.rank()
    out["memory"] = H.memory()
    out["nonzeros"] = H.nonzeros()
    out["nodes"] = H.node_info().tolist()  # pre-order: row_off, rows, U_rows, U_rank, V_rank, leaf
    nA = np.linalg.norm(A)
    if full_dense:
        out["rel_err"] = float(np.linalg.norm(H.dense() - A) / nA)
    b = R.randn(n)
    out["mult_b_norm"] = float(np.linalg.norm(H.mult(b)[:, 0]))
    out["multT_b_norm"] = float(np.linalg.norm(H.mult(b, "T")[:, 0]))
    H.factor()
    x = H.solve(b)[:, 0]
    out["solve_resid_H"] = float(np.linalg.norm(H.mult(x)[:, 0] - b) / np.linalg.norm(b))
    out["solve_resid_A"] = float(np.linalg.norm(A @ x - b) / np.linalg.norm(b))
    out["x_norm"] = float(np.linalg.norm(x))
    if n <= 200:
        out["x"] = x.tolist()
    return out


def main():
    cases = [run_case(*c) for c in CTEST]
    golden = dict(
        generator="tests/golden/make_golden.py (reference v8.0.0 via oracle/_ref, MKL, 8 threads)",
        rng_first_1000=R.randn(1000).tolist(),
        L10=R.test_matrix("L", 10).tolist(),
        cases=cases)
    with open(os.path.join(HERE, "hss_seq_golden.json"), "w") as f:
        json.dump(golden, f)
    for c in cases:
        print(c["name"], c["levels"], c["rank"], c.get("rel_err"), c["solve_resid_H"])


if __name__ == "__main__":
    main()
