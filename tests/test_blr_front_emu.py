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


@pytest.mark.parametrize("n,ny", [(5, 5), (6, 3)])
def test_generator_is_the_exact_front(n, ny):
    """closed form (sine basis) == Schur complement of the assembled 3D operator onto separator + update planes (square and
    rectangular separator planes: the 200^3 problem's second-level separators are 200 x 100)"""
    pl, pr, leaf = 2, 3, 7
    fr = BF.poisson_front(n, pl, pr, leaf, ny=ny)
    Lz = pl + pr + 3
    npl = n * ny
    N = npl * Lz
    A = np.zeros((N, N))
    idx = lambda ix, iy, iz: (iz * n + ix) * ny + iy
    for iz in range(Lz):
        for ix in range(n):
            for iy in range(ny):
                i = idx(ix, iy, iz)
                A[i, i] = 6.0
                for dx, dy, dz in ((1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)):
                    jx, jy, jz = ix + dx, iy + dy, iz + dz
                    if 0 <= jx < n and 0 <= jy < ny and 0 <= jz < Lz:
                        A[i, idx(jx, jy, jz)] = -1.0
    perm, tiles, _ = BF.plane_order(n, leaf, ny)
    assert sorted(perm.tolist()) == list(range(npl)) and sum(tiles) == npl and max(tiles) <= leaf
    c = pl + 1
    S = np.concatenate([c * npl + perm, perm, (Lz - 1) * npl + perm])
    I = np.array([i for z in range(Lz) if z not in (0, c, Lz - 1) for i in range(z * npl, (z + 1) * npl)])
    F = A[np.ix_(S, S)] - A[np.ix_(S, I)] @ np.linalg.solve(A[np.ix_(I, I)], A[np.ix_(I, S)])
    ds = npl
    assert np.abs(F[:ds, :ds] - fr["F11"]).max() < 1e-13
    assert np.abs(F[:ds, ds:] - fr["F12"]).max() < 1e-13
    assert np.abs(F[ds:, :ds] - fr["F21"]).max() < 1e-13
    # F22 of a front holds the children's contributions only: the update planes' own rows are assembled by the parent
    assert np.abs(F[ds:, ds:] - A[np.ix_(S[ds:], S[ds:])] - fr["F22"]).max() < 1e-13


def test_device_side_generator_matches():
    """the torch form of the generator (bench.py builds the large fronts with it on the GPU and leaves them there) against the
    numpy one, incl. the memory layouts it promises (column-major blocks)"""
    import torch
    n, ny, pl, pr, leaf = 7, 4, 3, 2, 9
    fr = BF.poisson_front(n, pl, pr, leaf, ny=ny)
    fd = BF.poisson_front_device(torch, n, ny, pl, pr, leaf, device=torch.device("cpu"))
    ds, du = fd["ds"], fd["du"]
    assert fd["tiles1"] == fr["tiles1"] and fd["tiles2"] == fr["tiles2"]
    cm = lambda t, r, c: t.numpy().reshape(-1).reshape((r, c), order="F")   # the tensor's memory read as a column-major block
    assert np.abs(cm(fd["F11"], ds, ds) - fr["F11"]).max() < 1e-13
    assert np.abs(cm(fd["F12cm"], ds, du) - fr["F12"]).max() < 1e-13
    assert np.abs(cm(fd["F21cm"], du, ds) - fr["F21"]).max() < 1e-13
    assert np.abs(cm(fd["F22"], du, du) - fr["F22"]).max() < 1e-13
    nF = np.sqrt(sum(np.linalg.norm(fr[k]) ** 2 for k in ("F11", "F12", "F21")))
    assert abs(fd["norm"] - nF) < 1e-10 * nF
