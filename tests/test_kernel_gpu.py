"""Kernel-matrix front end on the MI355X against fixtures made by the reference
(tests/golden/make_golden_kernel.py): clustering, compression from coordinates, fit + predict through the C-ABI."""
import pytest

import kernel_golden as KG
from strumpack_amd import _loader
from strumpack_amd import kernel as KM

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    return KM.load(_loader.lib_path())


@pytest.mark.parametrize("tag", ["gauss_400", "laplace_400", "anova_400", "gauss_1500"])
def test_regression_with_reference_neighbours(lib, tag):
    # the reference's own neighbour lists -> same columns, same IDs: per-node ranks equal, weights to 1e-7
    KG.check_regression(KM, lib, tag, inject=True, acc_tol=0.0, rank_tol=0.0, w_tol=1e-7)


@pytest.mark.parametrize("tag", ["gauss_400", "gauss_1500"])
def test_regression_with_device_neighbours(lib, tag):
    # exact neighbours (hssk_knn) instead of the reference's approximate ones: same classifier, slightly different sample
    KG.check_regression(KM, lib, tag, inject=False, acc_tol=0.02, rank_tol=0.15, w_tol=2e-2)


@pytest.mark.parametrize("tag", ["gauss_400", "laplace_400", "gauss_1500", "gauss_10k"])
def test_regression_reference_neighbour_search_end_to_end(lib, tag):
    """--hss_neighbor_search ann: the reference's randomized neighbour search on the host -> the whole pipeline (clustering,
    neighbours, compression rounds, ULV, solve) reproduces the reference from the raw data: same permutation, same
    per-node ranks, same weights."""
    import numpy as np
    J, Z = KG.golden()
    g = J["regression_" + tag]
    X, y, T, yt = KG.susy()
    n = g["n"]
    kr = KM.KernelRegression(lib, h=g["h"], lam=g["lam"], kernel=KG.KERNEL_NAME[g["ktype"]], degree=g["p"],
                             argv=KG.fit_args(g) + ["--hss_neighbor_search", "ann"]).fit(X[:n], y[:n])
    assert np.array_equal(kr.permutation(), Z["perm_" + tag])
    nodes, ref = kr.node_info(), np.array(g["nodes"])
    dr = np.abs(nodes[:, 3] - ref[:, 3])
    assert dr.max() <= 1 and (dr > 0).mean() <= 0.1, (nodes[:, 3], ref[:, 3])
    assert kr.info()["neighbors"] == g["ann_final"]
    w, wr = kr.weights(), Z["weights_" + tag]
    assert np.linalg.norm(w - wr) <= 1e-6 * np.linalg.norm(wr), np.linalg.norm(w - wr) / np.linalg.norm(wr)
    kr.destroy()


def test_kernel_regression_example_10k(lib):
    """examples/dense/KernelRegression.cpp on its shipped data set (susy_10Kn, h = 1.3, lambda = 3.11, defaults)."""
    # the weights of this ill-conditioned system move by a few percent with the column sample (rel_tol = 1e-2)
    info = KG.check_regression(KM, lib, "gauss_10k", inject=False, acc_tol=0.01, rank_tol=0.15, w_tol=5e-2)
    print("kernel regression 10k (device neighbours):", info)
    info = KG.check_regression(KM, lib, "gauss_10k", inject=True, acc_tol=0.0, rank_tol=0.0, w_tol=1e-6)
    print("kernel regression 10k (reference neighbours):", info)
