"""CPU tier of the BLR frontal-matrix path (BASELINE configs[4]): the product's host engine + kernel sources on the fiber
emulator against fixtures of the reference's BLRMatrix::construct_and_partial_factor, and the test generator against a
brute-force Schur complement of the assembled 7-point Laplacian."""
import numpy as np
import pytest

import blr_cases as BC
import blr_fronts as BF
import emu_lib
from strumpack_amd import capi


@pytest.fixture(scope="module")
def L():
    return capi.load(emu_lib.build())


@pytest.mark.parametrize("name", [k for k, v in BC.CASES.items() if v[-1] == "emu"])
def test_front_against_reference(L, name):
    BC.check_front(L, name)


@pytest.mark.parametrize("name", ("p16_weak", "p12_unsym_strong"))
def test_front_aca_against_reference(L, name):
    BC.check_front_aca(L, name)


@pytest.mark.parametrize("name", ("p16_weak", "p12_unsym_strong"))
def test_front_against_reference_star_and_comb(L, name):
    BC.check_front_schedules(L, name)


@pytest.mark.parametrize("la", [1, 2, 3])
@pytest.mark.parametrize("name", ("p16_weak", "p12_unsym_strong"))
def test_front_lookahead_depths(L, name, la, monkeypatch):
    # the trailing array is updated once per block of `la` block steps (left-looking inside the block): depth 1 is the
    # right-looking schedule as written; every depth must reproduce the reference's tile table and Schur complement
    monkeypatch.setenv("STRUMPACK_AMD_BLR_LOOKAHEAD", str(la))
    BC.check_front(L, name)


def test_front_api(L):
    BC.check_front_api(L)


def test_generator_is_the_exact_front():
    """closed form (sine basis) == Schur complement of the assembled 3D operator onto separator + update planes"""
    n, pl, pr, leaf = 5, 2, 3, 7
    fr = BF.poisson_front(n, pl, pr, leaf)
    Lz = pl + pr + 3
    N = n * n * Lz
    A = np.zeros((N, N))
    idx = lambda ix, iy, iz: (iz * n + ix) * n + iy
    for iz in range(Lz):
        for ix in range(n):
            for iy in range(n):
                i = idx(ix, iy, iz)
                A[i, i] = 6.0
                for dx, dy, dz in ((1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)):
                    jx, jy, jz = ix + dx, iy + dy, iz + dz
                    if 0 <= jx < n and 0 <= jy < n and 0 <= jz < Lz:
                        A[i, idx(jx, jy, jz)] = -1.0
    perm, tiles, _ = BF.plane_order(n, leaf)
    assert sorted(perm.tolist()) == list(range(n * n)) and sum(tiles) == n * n and max(tiles) <= leaf
    c = pl + 1
    S = np.concatenate([c * n * n + perm, perm, (Lz - 1) * n * n + perm])
    I = np.array([i for z in range(Lz) if z not in (0, c, Lz - 1) for i in range(z * n * n, (z + 1) * n * n)])
    F = A[np.ix_(S, S)] - A[np.ix_(S, I)] @ np.linalg.solve(A[np.ix_(I, I)], A[np.ix_(I, S)])
    ds = n * n
    assert np.abs(F[:ds, :ds] - fr["F11"]).max() < 1e-13
    assert np.abs(F[:ds, ds:] - fr["F12"]).max() < 1e-13
    assert np.abs(F[ds:, :ds] - fr["F21"]).max() < 1e-13
    # F22 of a front holds the children's contributions only: the update planes' own rows are assembled by the parent
    assert np.abs(F[ds:, ds:] - A[np.ix_(S[ds:], S[ds:])] - fr["F22"]).max() < 1e-13
