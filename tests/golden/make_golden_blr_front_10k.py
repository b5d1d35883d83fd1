"""Fixtures of the reference's BLRMatrix<double>::construct_and_partial_factor (BLR/BLRMatrix.cpp:740-1037) on fronts with a
separator of 10 000 unknowns -- a 100 x 100 plane of the 3D Poisson problem, the size class of the BASELINE configs[4]
problem's upper fronts (its root is 200 x 200) -- produced like tests/golden/make_golden_blr_front.py (same driver
oracle/ref/ref_driver.cpp: ref_blr_front, compiled from the reference's sources by oracle/ref/Makefile), one-off in the
build container (the reference needs minutes per front on eight cores).
    python tests/golden/make_golden_blr_front_10k.py [case ...]   -> tests/golden/blr_front_10k_golden.npz (cases merged in)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import blr_cases as BC   # noqa: E402
import blr_fronts as BF  # noqa: E402
from oracle import ref_lib as R  # noqa: E402

PATH = os.path.join(ROOT, "tests", "golden", "blr_front_10k_golden.npz")
out = dict(np.load(PATH)) if os.path.exists(PATH) else {}
for name in (sys.argv[1:] or sorted(BC.BIG_CASES)):
    fr = BC.build_case(name)
    args = (fr["F11"], fr["F12"], fr["F21"], fr["F22"], fr["tiles1"], fr["tiles2"], fr["rel_tol"], fr["abs_tol"])
    t0 = time.time()
    ref = R.blr_front(*args, adm=fr["adm"], bsep=fr["bsep"], bupd=fr["bupd"], ysep=fr["ysep"], yupd=fr["yupd"])
    t1 = time.time() - t0
    z = np.zeros_like(fr["bupd"])
    r2 = R.blr_front(*args, adm=fr["adm"], bsep=fr["bsep"], bupd=z, ysep=fr["bsep"], yupd=z)
    r3 = R.blr_front(*args, adm=fr["adm"], bsep=fr["bsep"], bupd=z, ysep=r2["bsep"], yupd=z)
    x11 = r3["ysep"]
    S = ref["S"]
    du = S.shape[0]
    out[name + "_ranks"] = ref["ranks"]
    out[name + "_stats"] = ref["stats"]
    out[name + "_SR"] = S @ fr["R"] if du else np.zeros((0, BC.NRHS))
    out[name + "_StR"] = S.T @ fr["R"] if du else np.zeros((0, BC.NRHS))
    out[name + "_Snorm"] = np.linalg.norm(S)
    out[name + "_Serr"] = BC.err(S, BF.dense_schur(fr)) if du else 0.0
    out[name + "_fwd_sep"], out[name + "_fwd_upd"], out[name + "_bwd_sep"] = ref["bsep"], ref["bupd"], ref["ysep"]
    out[name + "_x11"] = x11
    out[name + "_x11_resid"] = BC.err(fr["F11"] @ x11, fr["bsep"])
    out[name + "_ref_seconds"] = t1
    lr = ref["ranks"][ref["ranks"] >= 0]
    print("%-18s dsep %5d dupd %5d tiles %3d+%3d  ref %.1fs  max rank %d  mean rank %.1f  nnz %s  Schur err %.2e  B11\\b resid %.2e" %
          (name, fr["F11"].shape[0], du, len(fr["tiles1"]), len(fr["tiles2"]), t1, ref["stats"][4], lr.mean() if lr.size else 0,
           ref["stats"][1:4].astype(int).tolist(), out[name + "_Serr"], out[name + "_x11_resid"]), flush=True)
    np.savez_compressed(PATH, **out)
