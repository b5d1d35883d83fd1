"""Generates tests/golden/schur_golden.npz by RUNNING THE REFERENCE ITSELF (oracle/_ref): for cases of the reference's
test_HSS_seq sweep (test/CMakeLists.txt:57-146; test_HSS_seq.cpp:252-260 calls partial_factor + Schur_update there
without checking the result) and the BASELINE-shaped Toeplitz cases, the Schur complement of the (0,0) block as the
sparse HSS front forms and applies it (sparse/fronts/FrontHSS.cpp:391-407, :218): Theta, the shapes of DUB01 / Phi /
Vhat, and Sr = S R, Sc = S^T R from Schur_product_direct on a fixed R.  Only data is stored.  Build container only:

    make -C oracle/ref && python tests/golden/make_golden_schur.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(HERE, ".."))
from oracle import ref_lib as R  # noqa: E402
import hss_cases as HC  # noqa: E402

CASES = ["HSS_seq_1", "HSS_seq_2", "HSS_seq_5", "HSS_seq_8", "HSS_seq_11", "HSS_seq_12", "HSS_seq_14", "HSS_seq_22",
         "config1_T4096_defaults", "config2shape_T8192_leaf256_rtol1e-4"]
NCOL = 2


def main():
    G = HC.golden_cases()
    out, meta = {}, {}
    for name in CASES:
        c = G[name]
        n = c["n"]
        A = R.test_matrix(c["problem"], n)
        H = R.RefHSS(A, rel_tol=c["rel_tol"], abs_tol=c["abs_tol"], leaf=c["leaf_size"], d0=c["d0"], dd=c["dd"],
                     algo=c["algorithm"])
        s = H.schur_update()
        n1 = s["Theta"].shape[0]
        Rm = HC.randn(n1 * NCOL).reshape(n1, NCOL, order="F")
        Sr, Sc = H.schur_product_direct(Rm)
        meta[name] = dict(n1=n1, theta=list(s["Theta"].shape), dub01=list(s["DUB01"].shape), phi=list(s["Phi"].shape),
                          vhat=list(s["Vhat"].shape), theta_fro=float(np.linalg.norm(s["Theta"])),
                          # invariant of the reduced coordinates: Vhat^T DUB01 = V0big^T H00^{-1} U0big B01
                          vtdub01_fro=float(np.linalg.norm(s["Vhat"].T @ s["DUB01"])))
        out[name + "/Sr"] = Sr
        out[name + "/Sc"] = Sc
        if s["Theta"].size <= 10000:
            out[name + "/Theta"] = s["Theta"]
            out[name + "/VtDUB01"] = s["Vhat"].T @ s["DUB01"]
        print(name, meta[name])
    np.savez_compressed(os.path.join(HERE, "schur_golden.npz"), **out)
    with open(os.path.join(HERE, "schur_golden.json"), "w") as f:
        json.dump(dict(generator="tests/golden/make_golden_schur.py (reference v8.0.0 via oracle/_ref)", ncol=NCOL,
                       cases=meta), f, indent=1)


if __name__ == "__main__":
    main()
