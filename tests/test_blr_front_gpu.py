"""GPU parity tests of the BLR frontal-matrix path (BASELINE configs[4]: BLR-compressed fronts of the 3D Poisson problem,
batched LU) through the C interface SPX_d_blr_front_*: fixtures of the reference's own
BLRMatrix::construct_and_partial_factor (tests/golden/make_golden_blr_front.py, make_golden_blr_front_10k.py) on fronts with
dsep 100 ... 10000."""
import numpy as np
import pytest

import blr_cases as BC
from strumpack_amd import _loader, capi
from strumpack_amd import hssk as K

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    return capi.load(_loader.lib_path())


@pytest.mark.parametrize("name", sorted(BC.CASES))
def test_front_against_reference(L, name):
    BC.check_front(L, name)


@pytest.mark.parametrize("name", sorted(BC.BIG_CASES))
def test_front_10k_against_reference(L, name):
    # separators of 10 000 unknowns (a 100 x 100 plane: the size class of the 200^3 problem's upper fronts), fixtures of the
    # reference's own run (tests/golden/make_golden_blr_front_10k.py: 277 / 498 s per factorization on eight cores)
    BC.check_front(L, name)


@pytest.mark.parametrize("name", BC.ACA_CASES)
def test_front_aca_against_reference(L, name):
    BC.check_front_aca(L, name)


@pytest.mark.parametrize("name", BC.ACA_CASES)
def test_front_against_reference_star_and_comb(L, name):
    BC.check_front_schedules(L, name)


@pytest.mark.parametrize("la", [1, 2, 3])
@pytest.mark.parametrize("name", ("p40_weak", "p64_weak"))
def test_front_lookahead_depths(L, name, la, monkeypatch):
    # the trailing array is updated once per block of `la` block steps (left-looking inside the block): depth 1 is the
    # right-looking schedule as written; every depth must reproduce the reference's tile table and Schur complement
    monkeypatch.setenv("STRUMPACK_AMD_BLR_LOOKAHEAD", str(la))
    BC.check_front(L, name)


def test_front_api(L):
    BC.check_front_api(L)


def test_front_device_operands(L):
    """operands resident in HBM (what a multifrontal driver that assembles fronts on the device hands over): same factors,
    Schur complement left in HBM"""
    fr = BC.build_case("p40_weak")
    o = capi.StructuredMatrix.options(L, rel_tol=fr["rel_tol"], abs_tol=fr["abs_tol"], type=capi.SP_TYPE_BLR)
    Fh, Sh = capi.BLRFront.factor(L, fr["F11"], fr["F12"], fr["F21"], fr["F22"], fr["tiles1"], fr["tiles2"], o)
    hk = K.Hssk(_loader.lib_path())
    d = {k: hk.array(np.asfortranarray(fr[k])) for k in ("F11", "F12", "F21", "F22")}
    ds, du = fr["F11"].shape[0], fr["F12"].shape[1]
    Fd = capi.BLRFront.factor_device(L, ds, du, d["F11"].ptr, ds, d["F12"].ptr, ds, d["F21"].ptr, du, d["F22"].ptr, du,
                                     fr["tiles1"], fr["tiles2"], o)
    assert np.array_equal(Fd.tile_ranks(), Fh.tile_ranks())
    Sd = Fd.schur()
    assert BC.err(Sd, Sh) <= 1e-13
    ptr, ld = Fd.schur_device()
    assert ptr and ld >= du
    # the caller's operands are left untouched
    assert np.array_equal(d["F11"].get(), fr["F11"]) and np.array_equal(d["F22"].get(), fr["F22"])
    x = Fd.solve11(fr["bsep"])
    assert BC.err(x, Fh.solve11(fr["bsep"])) <= 1e-12
    Fd.destroy()
    Fh.destroy()
    hk.close()


@pytest.mark.parametrize("nx,ny,upd", [(200, 200, "none"), (200, 100, "both")])
def test_front_at_the_200cubed_problems_own_sizes(L, nx, ny, upd):
    r"""BASELINE configs[4] at ITS front sizes -- what the reference hands to BLRMatrix::construct_and_partial_factor
    (BLR/BLRMatrix.cpp:740) on the 200^3 Poisson problem: the root separator (a 200 x 200 plane: dsep 40000, no update part)
    and a second-level front (200 x 100 plane: dsep 20000, dupd 40000).  The reference's routine needs minutes to hours on
    these, so the checks are the size-independent properties of a partial factorization, on the device (torch is plumbing):
    F11 (B11 \ b) = b to the compression tolerance; the Schur complement applied to sampled vectors against dense algebra
    (F22 R - F21 F11^{-1} F12 R with a dense solve); low-rank tiles only where the rank pays (r (m + n) <= m n,
    BLRMatrix.cpp:568) and ranks that do not grow when the tolerance is loosened."""
    import torch
    import blr_fronts as BF
    from strumpack_amd import dist as sdist
    leaf = 256
    fr = BF.poisson_front_device(torch, nx, ny, 8, 8, leaf, upd=upd)
    ds, du = fr["ds"], fr["du"]
    ptr = lambda k: fr[k].data_ptr() if k in fr else None
    rng = np.random.default_rng(5)
    b, bu = rng.standard_normal((ds, 2)), (rng.standard_normal((du, 2)) if du else None)
    dev = fr["F11"].device
    ranks = {}
    for rtol in (1e-4, 1e-2):
        o = capi.StructuredMatrix.options(L, rel_tol=rtol, abs_tol=1e-12 * fr["norm"], type=capi.SP_TYPE_BLR)
        torch.cuda.synchronize()
        F = capi.BLRFront.factor_device(L, ds, du, ptr("F11"), ds, ptr("F12cm"), ds, ptr("F21cm"), max(du, 1), ptr("F22"), max(du, 1),
                                        fr["tiles1"], fr["tiles2"], o)
        rk = F.tile_ranks()
        ranks[rtol] = rk
        nt = len(fr["tiles1"]) + len(fr["tiles2"])
        sizes = np.array(list(fr["tiles1"]) + list(fr["tiles2"]))
        assert rk.shape == (nt, nt)
        lr = rk >= 0
        # (the diagonal is never compressed; a low-rank tile pays for itself)
        assert not lr[np.arange(nt), np.arange(nt)].any()
        mm, nn = np.meshgrid(sizes, sizes, indexing="ij")
        assert (rk[lr] * (mm[lr] + nn[lr]) <= mm[lr] * nn[lr]).all()
        if rtol == 1e-4:
            ys, yu = F.forward(b, bu)
            x = F.backward(ys, np.zeros_like(bu) if du else None)
            xt, bt = torch.from_numpy(np.ascontiguousarray(x)).to(dev), torch.from_numpy(b).to(dev)
            resid = float(torch.linalg.norm(fr["F11"] @ xt - bt) / torch.linalg.norm(bt))     # (F11 symmetric: row-major == column-major)
            assert resid <= 20 * rtol, resid
            if du:
                R = torch.from_numpy(rng.standard_normal((du, 4))).to(dev)
                sp, ld = F.schur_device()
                St = sdist._tensor(sp, ld * du, True).view(du, ld)[:, :du]        # St[j, i] = S(i, j)
                SR = St.t() @ R
                ref = fr["F22"] @ R - fr["F21cm"].t() @ torch.linalg.solve(fr["F11"], fr["F12cm"].t() @ R)
                err = float(torch.linalg.norm(SR - ref) / torch.linalg.norm(ref))
                assert err <= 20 * rtol, err
                del St, SR, ref, R
        F.destroy()
    both = (ranks[1e-4] >= 0) & (ranks[1e-2] >= 0)
    # (tile by tile up to the noise of the running Schur complements -- the two factorizations update with different
    # approximations --, strictly in the sum and in the largest rank)
    assert (ranks[1e-2][both] <= ranks[1e-4][both] + 2).all()
    assert ranks[1e-2][both].sum() < ranks[1e-4][both].sum() and ranks[1e-2].max() <= ranks[1e-4].max()
    del fr
    torch.cuda.empty_cache()
    L.SPX_device_pool_trim()
