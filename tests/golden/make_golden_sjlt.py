"""Generates tests/golden/sjlt_golden.json by RUNNING THE REFERENCE ITSELF (oracle/_ref) on its four CTest lines for the
SJLT sketch (test/CMakeLists.txt:145-159, HSS_seq_23 .. 26) and an ORIGINAL-algorithm variant.  The reference seeds the
SJLT pattern generator from the clock (HSS/HSSMatrix.sketch.hpp:266-270), so there is no fixed answer: each case is
run REPS times and the observed rank range and worst compression error are recorded.  Build container only:

    make -C oracle/ref && python tests/golden/make_golden_sjlt.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import ref_lib as R  # noqa: E402

# name, problem, n, leaf, rel_tol, abs_tol, algorithm, d0, dd, SJLT algo, nnz0, nnz
CASES = [
    ("HSS_seq_23", "T", 1000, 32, 1e-5, 1e-10, "stable", 8, 8, "chunk", 2, 2),
    ("HSS_seq_24", "T", 1000, 32, 1e-5, 1e-10, "stable", 8, 8, "chunk", 4, 4),
    ("HSS_seq_25", "T", 1000, 32, 1e-5, 1e-10, "stable", 8, 8, "perm", 2, 2),
    ("HSS_seq_26", "T", 1000, 32, 1e-5, 1e-10, "stable", 8, 8, "perm", 4, 4),
    ("sjlt_original_T500", "T", 500, 16, 1e-8, 1e-12, "original", 32, 16, "chunk", 4, 4),
    ("sjlt_stable_U400", "U", 400, 16, 1e-6, 1e-10, "stable", 32, 8, "perm", 3, 2),
    # more than 8 nonzeros per row: beyond the device pattern (dense block instead)
    ("sjlt_dense_T600", "T", 600, 32, 1e-6, 1e-10, "stable", 48, 16, "chunk", 12, 10),
    ("sjlt_dense_perm_T600", "T", 600, 32, 1e-6, 1e-10, "stable", 48, 16, "perm", 16, 9),
]
REPS = 8


def main():
    out = {}
    for name, prob, n, leaf, rtol, atol, algo, d0, dd, sj, nnz0, nnz in CASES:
        A = R.test_matrix(prob, n)
        ranks, errs, res = [], [], []
        for _ in range(REPS):
            H = R.RefHSS(A, rel_tol=rtol, abs_tol=atol, leaf=leaf, d0=d0, dd=dd, algo=algo, sjlt=(sj, nnz0, nnz))
            assert H.is_compressed()
            ranks.append(H.rank())
            errs.append(float(np.linalg.norm(H.dense() - A) / np.linalg.norm(A)))
            H.factor()
            b = R.randn(n)
            x = H.solve(b)[:, 0]
            res.append(float(np.linalg.norm(H.mult(x)[:, 0] - b) / np.linalg.norm(b)))
        out[name] = dict(problem=prob, n=n, leaf_size=leaf, rel_tol=rtol, abs_tol=atol, algorithm=algo, d0=d0, dd=dd,
                         sjlt_algo=sj, nnz0=nnz0, nnz=nnz, rank_min=min(ranks), rank_max=max(ranks),
                         rel_err_max=max(errs), solve_resid_max=max(res))
        print(name, out[name])
    with open(os.path.join(HERE, "sjlt_golden.json"), "w") as f:
        json.dump(dict(generator="tests/golden/make_golden_sjlt.py (reference v8.0.0 via oracle/_ref)", reps=REPS,
                       cases=out), f, indent=1)


if __name__ == "__main__":
    main()
