"""Shared checks of the BLR frontal-matrix path (BASELINE configs[4]) for the emulator tier and the GPU
tier: SPX_d_blr_front_* against fixtures produced by the reference's BLRMatrix::construct_and_partial_factor
(tests/golden/make_golden_blr_front.py) and against dense algebra."""
import os

import numpy as np

import blr_fronts as BF
from strumpack_amd import capi

HERE = os.path.dirname(os.path.abspath(__file__))

# name: (n, p_left, p_right, leaf, upd, unsym, admissibility, rel_tol, tier)
CASES = {
    "p16_weak": (16, 3, 4, 32, "both", False, "weak", 1e-4, "emu"),
    "p12_unsym_strong": (12, 2, 2, 24, "left", True, "strong", 1e-6, "emu"),
    "p10_root": (10, 2, 3, 25, "none", False, "weak", 1e-4, "emu"),
    "p40_weak": (40, 8, 8, 128, "both", False, "weak", 1e-4, "gpu"),
    "p64_weak": (64, 8, 8, 256, "both", False, "weak", 1e-4, "gpu"),
    "p64_unsym_strong": (64, 6, 10, 256, "left", True, "strong", 1e-6, "gpu"),
}
# fronts with a separator of 10 000 unknowns (a 100 x 100 plane: the size class of the 200^3 problem's upper fronts), GPU tier;
# fixtures: tests/golden/make_golden_blr_front_10k.py -> blr_front_10k_golden.npz
BIG_CASES = {
    "p100_root": (100, 12, 12, 256, "none", False, "weak", 1e-4, "gpu"),
    "p100_left": (100, 10, 14, 256, "left", False, "weak", 1e-4, "gpu"),
}
NRHS = 3
ACA_CASES = ("p16_weak", "p12_unsym_strong", "p40_weak")   # also recorded with ACA tile compression


def build_case(name):
    n, pl, pr, leaf, upd, unsym, admk, rtol, _ = (CASES.get(name) or BIG_CASES[name])
    fr = BF.poisson_front(n, pl, pr, leaf, upd=upd, unsym=unsym)
    # sparse/fronts/FrontBLR.cpp:424-429: the absolute tolerance is scaled by the norm of [F11 F12; F21 0]
    nF = np.sqrt(sum(np.linalg.norm(fr[k]) ** 2 for k in ("F11", "F12", "F21")))
    fr["rel_tol"], fr["abs_tol"] = rtol, 1e-12 * nF
    fr["adm"] = BF.strong_admissibility(fr["boxes"]) if admk == "strong" else None
    ds, du = fr["F11"].shape[0], fr["F12"].shape[1]
    rng = np.random.default_rng(sum(map(ord, name)))
    fr["bsep"], fr["bupd"] = rng.standard_normal((ds, NRHS)), rng.standard_normal((du, NRHS))
    fr["ysep"], fr["yupd"] = rng.standard_normal((ds, NRHS)), rng.standard_normal((du, NRHS))
    fr["R"] = rng.standard_normal((du, NRHS))
    return fr


def golden():
    G = dict(np.load(os.path.join(HERE, "golden", "blr_front_golden.npz")))
    big = os.path.join(HERE, "golden", "blr_front_10k_golden.npz")
    if os.path.exists(big):
        G.update(np.load(big))
    return G


def err(a, b):
    nb = np.linalg.norm(b)
    return np.linalg.norm(a - b) / (nb if nb > 0 else 1.0)


def check_front(L, name, G=None):
    G = G if G is not None else golden()
    fr = build_case(name)
    rtol = fr["rel_tol"]
    o = capi.StructuredMatrix.options(L, rel_tol=rtol, abs_tol=fr["abs_tol"], type=capi.SP_TYPE_BLR)
    F, S = capi.BLRFront.factor(L, fr["F11"], fr["F12"], fr["F21"], fr["F22"], fr["tiles1"], fr["tiles2"], o, admissible=fr["adm"])
    ds, du = F.dsep, F.dupd
    nt1, nt = len(fr["tiles1"]), len(fr["tiles1"]) + len(fr["tiles2"])
    # ---- tile ranks: same tiles, same truncated-RRQR rule as the reference
    rk, rref = F.tile_ranks(), G[name + "_ranks"]
    part = np.ones((nt, nt), dtype=bool)
    part[nt1:, nt1:] = False            # (F22 is never tiled)
    assert np.array_equal(rk[part] < 0, rref[part] < 0), (name, "dense / low-rank decisions differ")
    lr = part & (rref >= 0)
    diff = np.abs(rk[lr] - rref[lr])
    assert diff.max(initial=0) <= max(1, int(0.15 * rref[lr].max(initial=1))), (name, diff.max())
    assert (diff > 0).mean() <= 0.05 if diff.size else True, (name, (diff > 0).mean())
    st = F.stats()
    for k, q in enumerate(("nnz11", "nnz12", "nnz21")):
        refnz = G[name + "_stats"][1 + k]
        assert abs(st[q] - refnz) <= 0.02 * max(refnz, 1), (name, q, st[q], refnz)
    # ---- Schur complement F22 - F21 F11^{-1} F12: against the reference's (through products and its norm) and dense algebra
    if du:
        assert err(S @ fr["R"], G[name + "_SR"]) <= 10 * rtol
        assert err(S.T @ fr["R"], G[name + "_StR"]) <= 10 * rtol
        assert abs(np.linalg.norm(S) - G[name + "_Snorm"]) <= 10 * rtol * G[name + "_Snorm"]
        Sx = BF.dense_schur(fr)
        assert err(S, Sx) <= max(2 * float(G[name + "_Serr"]), 10 * rtol), (name, err(S, Sx), float(G[name + "_Serr"]))
        assert np.array_equal(S, F.schur())       # the device copy is what the call returned
    # ---- the front's solve phases (FrontBLR.cpp:525-570) and B11 \ b
    fs, fu = F.forward(fr["bsep"], fr["bupd"])
    assert err(fs, G[name + "_fwd_sep"]) <= 10 * rtol
    if du:
        assert err(fu, G[name + "_fwd_upd"]) <= 10 * rtol
    ys = F.backward(fr["ysep"], fr["yupd"])
    assert err(ys, G[name + "_bwd_sep"]) <= 10 * rtol
    x = F.solve11(fr["bsep"])
    assert err(x, G[name + "_x11"]) <= 10 * rtol
    assert err(fr["F11"] @ x, fr["bsep"]) <= max(3 * float(G[name + "_x11_resid"]), 10 * rtol)
    # one right-hand side: each substitution is ONE launch (hssk_blr_sweep: a workgroup per block row, waiting for the rows its
    # non-zero tiles point at) -- the same factors applied in another order than the block steps above
    fs1, fu1 = F.forward(fr["bsep"][:, :1], fr["bupd"][:, :1] if du else None)
    assert err(fs1, fs[:, :1]) <= 1e-11
    if du:
        assert err(fu1, fu[:, :1]) <= 1e-11
    ys1 = F.backward(fr["ysep"][:, :1], fr["yupd"][:, :1] if du else None)
    assert err(ys1, ys[:, :1]) <= 1e-11
    F.destroy()
    return st


def check_front_schedules(L, name, G=None):
    """Star / Comb (LUAR) are other schedules of the same factorization; selected, they run the RL schedule here
    (BLRMatrix.hpp).  The result must sit inside RL's own tolerances of the reference's Star and Comb runs: same dense /
    low-rank decisions, tile ranks at most 15 % (at least 2) apart on at most 10 % of the tiles (Comb accumulates a block
    row's updates before it recompresses: against RL the ranks move by one or two on 5.3 % (Comb) / 4.9 % (Star) of the 1027
    low-rank tiles of the largest case; against the reference's RL run 0.1 % of them differ), Schur complement and solve phases equal to 10 x the compression tolerance."""
    G = G if G is not None else golden()
    fr = build_case(name)
    rtol = fr["rel_tol"]
    o = capi.StructuredMatrix.options(L, rel_tol=rtol, abs_tol=fr["abs_tol"], type=capi.SP_TYPE_BLR)
    F, S = capi.BLRFront.factor(L, fr["F11"], fr["F12"], fr["F21"], fr["F22"], fr["tiles1"], fr["tiles2"], o, admissible=fr["adm"])
    nt1, nt = len(fr["tiles1"]), len(fr["tiles1"]) + len(fr["tiles2"])
    rk = F.tile_ranks()
    part = np.ones((nt, nt), dtype=bool)
    part[nt1:, nt1:] = False
    fs, fu = F.forward(fr["bsep"], fr["bupd"])
    ys = F.backward(fr["ysep"], fr["yupd"])
    for algo in ("star", "comb"):
        k = name + "_" + algo
        rref = G[k + "_ranks"]
        assert np.array_equal(rk[part] < 0, rref[part] < 0), (k, "dense / low-rank decisions differ")
        lr = part & (rref >= 0)
        diff = np.abs(rk[lr] - rref[lr])
        assert diff.max(initial=0) <= max(2, int(0.15 * rref[lr].max(initial=1))) and (diff > 0).mean() <= 0.10, (k, diff.max(), (diff > 0).mean())
        assert err(S @ fr["R"], G[k + "_SR"]) <= 10 * rtol
        assert abs(np.linalg.norm(S) - G[k + "_Snorm"]) <= 10 * rtol * G[k + "_Snorm"]
        assert err(fs, G[k + "_fwd_sep"]) <= 10 * rtol and err(fu, G[k + "_fwd_upd"]) <= 10 * rtol
        assert err(ys, G[k + "_bwd_sep"]) <= 10 * rtol
    F.destroy()


def check_front_aca(L, name, G=None):
    """the same front with ACA tile compression (SPX_blr_low_rank_algorithm(1); C++: BLROptions::set_low_rank_algorithm):
    the reference's ACA takes its first row from a default-seeded std::mt19937 and is deterministic from there, so the tile
    ranks are the reference's (a different pivot after a rounding tie is allowed on a few tiles) and the Schur complement and
    the solve phases agree to the compression tolerance"""
    G = G if G is not None else golden()
    fr = build_case(name)
    rtol = fr["rel_tol"]
    o = capi.StructuredMatrix.options(L, rel_tol=rtol, abs_tol=fr["abs_tol"], type=capi.SP_TYPE_BLR)
    assert L.SPX_blr_low_rank_algorithm(2) != 0      # BACA: refused
    assert L.SPX_blr_low_rank_algorithm(1) == 0
    try:
        F, S = capi.BLRFront.factor(L, fr["F11"], fr["F12"], fr["F21"], fr["F22"], fr["tiles1"], fr["tiles2"], o, admissible=fr["adm"])
    finally:
        L.SPX_blr_low_rank_algorithm(0)
    nt1, nt = len(fr["tiles1"]), len(fr["tiles1"]) + len(fr["tiles2"])
    rk, rref = F.tile_ranks(), G[name + "_aca_ranks"]
    part = np.ones((nt, nt), dtype=bool)
    part[nt1:, nt1:] = False
    assert np.array_equal(rk[part] < 0, rref[part] < 0), (name, "dense / low-rank decisions differ")
    lr = part & (rref >= 0)
    diff = np.abs(rk[lr] - rref[lr])
    assert (diff > 0).mean() <= 0.1 and diff.max(initial=0) <= max(2, int(0.15 * rref[lr].max(initial=1))), (name, (diff > 0).mean(), diff.max())
    # ACA is less accurate than the pivoted QR at the same tolerance (the reference's own Schur complement error says how much)
    tol = max(10 * rtol, 3 * float(G[name + "_aca_Serr"]))
    assert err(S @ fr["R"], G[name + "_aca_SR"]) <= tol
    assert abs(np.linalg.norm(S) - G[name + "_aca_Snorm"]) <= tol * G[name + "_aca_Snorm"]
    assert err(S, BF.dense_schur(fr)) <= max(3 * float(G[name + "_aca_Serr"]), 10 * rtol)
    fs, fu = F.forward(fr["bsep"], fr["bupd"])
    assert err(fs, G[name + "_aca_fwd_sep"]) <= tol and err(fu, G[name + "_aca_fwd_upd"]) <= tol
    assert err(F.backward(fr["ysep"], fr["yupd"]), G[name + "_aca_bwd_sep"]) <= tol
    F.destroy()


def check_front_api(L):
    """argument checking and the empty shapes: a front without update part, with one tile, bad tile sums"""
    fr = build_case("p10_root")
    o = capi.StructuredMatrix.options(L, rel_tol=1e-4, abs_tol=1e-10, type=capi.SP_TYPE_BLR)
    # one tile = a dense LU of F11
    F, S = capi.BLRFront.factor(L, fr["F11"], None, None, None, [fr["F11"].shape[0]], [], o)
    x = F.solve11(fr["bsep"])
    assert err(fr["F11"] @ x, fr["bsep"]) <= 1e-12
    assert S.shape == (0, 0)
    F.destroy()
    # tile sizes that do not add up: an error code, not a crash
    try:
        capi.BLRFront.factor(L, fr["F11"], None, None, None, [3, 4], [], o)
        raise AssertionError("bad tile sizes must fail")
    except RuntimeError:
        pass
    # F22 = NULL is a zero block: the Schur complement is -F21 F11^{-1} F12
    fr = build_case("p12_unsym_strong")
    o = capi.StructuredMatrix.options(L, rel_tol=1e-8, abs_tol=1e-14, type=capi.SP_TYPE_BLR)
    F, S = capi.BLRFront.factor(L, fr["F11"], fr["F12"], fr["F21"], None, fr["tiles1"], fr["tiles2"], o)
    Sx = -fr["F21"] @ np.linalg.solve(fr["F11"], fr["F12"])
    assert err(S, Sx) <= 1e-6
    F.destroy()
