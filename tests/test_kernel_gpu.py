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


@pytest.mark.parametrize("algo", ["cobble", "kdtree"])
def test_clustering_device_form_is_the_host_form(lib, algo):
    """Median-split clustering on the device (kernels/hssk_cluster.hip: one launch per tree level) against the host form
    (libstdc++ calls on the reference's data, host/Clustering.hpp) -- the same permutation element for element, at the
    reference fixture's size and at BASELINE configs[3]'s; a lattice is handed back untouched."""
    import numpy as np
    J, Z = KG.golden()
    X = KG.susy()[0]
    g = J["clustering_full_%s" % algo]
    st, Xp, perm, leaves = KM.clustering_device(lib, X, algo, g["leaf"])
    if st == 0:
        assert np.array_equal(perm, Z["perm_full_%s" % algo]) and leaves.tolist() == g["leaves"]
    else:
        assert algo == "kdtree"   # (duplicated coordinate values in the data set: a tie at a median)
    import os
    r = np.random.default_rng(2025)
    for pts, leaf in ((r.random((100000, 8)), 256), (r.standard_normal((70001, 5)), 600)):
        Xh, ph, lh = KM.clustering(lib, pts, algo, leaf)
        # large clusters by several workgroups (the default from 16384 points), by one workgroup each, and from 5000 points
        for wide in (None, "0", "5000"):
            if wide is not None:
                os.environ["HSSK_CLUSTER_WIDE_MIN"] = wide
            try:
                st, Xp, perm, leaves = KM.clustering_device(lib, pts, algo, leaf)
            finally:
                os.environ.pop("HSSK_CLUSTER_WIDE_MIN", None)
            assert st == 0 and np.array_equal(perm, ph) and np.array_equal(Xp, Xh) and leaves.tolist() == lh.tolist()
    lat = r.integers(0, 3, (20000, 4)).astype(float)
    st, Xp, perm, leaves = KM.clustering_device(lib, lat, algo, 100)
    assert st > 0 and np.array_equal(Xp, lat)


def test_tsqr_staircase_matches_dense_sweep():
    """The TSQR pre-reduction of tall ID panels stacks R factors with interleaved rows and factors only the staircase
    (hssk_qr_desc.stair); the dense sweep of the same stack (STRUMPACK_AMD_TSQR_DENSE=1) must give the same matrix."""
    import os
    import numpy as np
    from strumpack_amd import capi, dist as sdist
    L = capi.load(_loader.lib_path())
    n, d = 20000, 8
    X = np.random.default_rng(7).random((n, d))
    o = capi.StructuredMatrix.options(L, rel_tol=1e-4, abs_tol=1e-10, leaf_size=256, max_rank=50000)
    res = []
    for dense in ("1", "0"):
        os.environ["STRUMPACK_AMD_TSQR_DENSE"] = dense
        H, Xp, perm = sdist.from_kernel(L, X.copy(), o, kernel="Gauss", h=1.3, lam=3.11, clustering="kdtree", neighbors=64)
        b = np.linspace(-1, 1, n)
        res.append((H.node_info().copy(), H.mult(b)[:, 0]))
        H.destroy()
    os.environ.pop("STRUMPACK_AMD_TSQR_DENSE")
    assert np.array_equal(res[0][0], res[1][0]), "ranks differ between the staircase and the dense TSQR sweeps"
    assert np.linalg.norm(res[0][1] - res[1][1]) <= 1e-10 * np.linalg.norm(res[0][1])


@pytest.mark.parametrize("kern,h,dim,clus,rtol", [("Gauss", 1.3, 8, "cobble", 1e-2), ("Laplace", 2.0, 3, "kdtree", 1e-3), ("Gauss", 3.0, 20, "cobble", 1e-2)])
def test_round6_front_end_against_the_forms_it_replaced(kern, h, dim, clus, rtol):
    """30000 points through the round-6 front end -- clustering on the device, the filtered neighbour search, the row ID from Gram
    matrices -- against the forms they replaced in a second process (host clustering, heap search, TSQR + register QRCP:
    STRUMPACK_AMD_CLUSTER_HOST / HSSK_KNN_FILTER / STRUMPACK_AMD_ID_GRAM): the same permutation, the same tree, ranks equal
    (one off tolerated on a twentieth of the nodes), products equal to the compression tolerance; and sampled rows of the kernel
    matrix itself."""
    import json
    import os
    import subprocess
    import sys
    import numpy as np
    import kernel_cases as KC
    code = (
        "import sys, json, numpy as np; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "from strumpack_amd import _loader, capi, dist as sdist\n"
        "L = capi.load(_loader.lib_path()); n, d = 30000, %d\n"
        "X = np.random.default_rng(77).random((n, d))\n"
        "o = capi.StructuredMatrix.options(L, rel_tol=%g, abs_tol=1e-10, leaf_size=256, max_rank=50000)\n"
        "H, Xp, perm = sdist.from_kernel(L, X, o, kernel=%r, h=%g, lam=2.5, clustering=%r, neighbors=64)\n"
        "b = np.linspace(-1, 1, n); y = H.mult(b)[:, 0]; H.factor(); x = H.solve(b)[:, 0]\n"
        "res = float(np.linalg.norm(H.mult(x)[:, 0] - b) / np.linalg.norm(b))\n"
        "np.savez(sys.argv[1], perm=perm, info=H.node_info(), y=y, Xp=Xp, res=res)\n"
    ) % (os.path.dirname(os.path.abspath(__file__)), os.path.dirname(os.path.dirname(os.path.abspath(__file__))), dim, rtol, kern, h, clus)
    out = []
    for mode, env in (("new", {}), ("old", {"STRUMPACK_AMD_CLUSTER_HOST": "1", "HSSK_KNN_FILTER": "0", "STRUMPACK_AMD_ID_GRAM": "0"})):
        f = "/tmp/r06_front_%s_%s_%d.npz" % (mode, kern, dim)
        r = subprocess.run([sys.executable, "-c", code, f], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        out.append(np.load(f))
    new, old = out
    assert np.array_equal(new["perm"], old["perm"]), "device clustering differs from the host form"
    assert np.array_equal(new["info"][:, [0, 1, 5]], old["info"][:, [0, 1, 5]]), "tree"
    dr = np.abs(new["info"][:, 3:5] - old["info"][:, 3:5])
    assert dr.max() <= 1 and (dr > 0).mean() <= 0.05, (int(dr.max()), float((dr > 0).mean()))
    assert np.linalg.norm(new["y"] - old["y"]) <= 10 * rtol * np.linalg.norm(old["y"])
    assert float(new["res"]) <= 1e-10 and float(old["res"]) <= 1e-10
    # sampled rows of the kernel matrix against the compressed one
    n = len(new["y"])
    I = np.random.default_rng(3).integers(0, n, 16)
    KI = KC.kernel_np(new["Xp"], I, np.arange(n), {"Gauss": 0, "Laplace": 1}[kern], h, 2.5)
    b = np.linspace(-1, 1, n)
    err = np.linalg.norm(new["y"][I] - KI @ b) / np.linalg.norm(KI @ b)
    assert err <= 1e2 * rtol, err


def test_full_size_properties_100k():
    """BASELINE configs[3] size (N = 100000 points in R^8, Gauss kernel, h = 1.3, lambda = 3.11) through size-independent
    properties: sampled rows of K against the compressed matrix, symmetry (V = U, B10 = B01^T), linearity, ULV residual."""
    import ctypes as C
    import numpy as np
    import kernel_cases as KC
    from strumpack_amd import capi, dist as sdist
    L = capi.load(_loader.lib_path())
    n, d = 100000, 8
    rng = np.random.default_rng(2025)
    X = rng.random((n, d))
    o = capi.StructuredMatrix.options(L, rel_tol=1e-2, abs_tol=1e-8, leaf_size=256, max_rank=50000)
    H, Xp, perm = sdist.from_kernel(L, X, o, kernel="Gauss", h=1.3, lam=3.11, clustering="cobble", neighbors=64)
    assert H.is_compressed() and H.levels() >= 9
    assert np.array_equal(Xp, X[perm - 1])
    assert 20 <= H.rank() <= 200, H.rank()
    # sampled rows: (K x)_I exactly vs (H x)_I
    x = rng.standard_normal((n, 2))
    I = rng.integers(0, n, 24)
    KI = KC.kernel_np(Xp, I, np.arange(n), 0, 1.3, 3.11)
    Hx = H.mult(x)
    err = np.linalg.norm(Hx[I] - KI @ x) / np.linalg.norm(KI @ x)
    assert err <= 1e2 * 1e-2 and err < 5e-2, err
    # symmetry and linearity
    assert np.allclose(H.mult(x, "T"), Hx, rtol=1e-10, atol=1e-10)
    y = rng.standard_normal((n, 2))
    assert np.allclose(H.mult(2 * x - 3 * y), 2 * Hx - 3 * H.mult(y), atol=1e-8)
    # ULV
    H.factor()
    b = rng.standard_normal((n, 3))
    w = H.solve(b)
    res = np.linalg.norm(H.mult(w) - b) / np.linalg.norm(b)
    assert res <= 1e-10, res
    H.destroy()
