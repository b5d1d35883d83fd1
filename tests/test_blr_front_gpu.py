"""GPU parity tests of the BLR frontal-matrix path (BASELINE configs[4]: BLR-compressed fronts of the 3D Poisson problem,
batched LU) through the C interface SPX_d_blr_front_*: fixtures of the reference's own
BLRMatrix::construct_and_partial_factor (tests/golden/make_golden_blr_front.py) on fronts with dsep 100 ... 4096."""
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
