"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy + LAPACK via scipy) of the reference's HSS
hot path: randomized compression -> hierarchical apply -> ULV factor -> ULV solve.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.  The
product (strumpack_amd/, HIP kernels behind the C-ABI in include/) never does and has no CPU
fallback.

Pinning: this oracle is checked in tests/test_oracle.py against (i) the committed fixtures in
tests/golden/ that were produced by the reference itself (oracle/_ref, built by oracle/ref/Makefile
from /root/reference; generator script tests/golden/make_golden.py), (ii) the reference's own test
assertions (test/test_HSS_seq.cpp:38-39,148-152,247-250) over its CTest sweep
(test/CMakeLists.txt:57-159) and (iii) the known first draws of the default random generator.

The algorithm is restated *level-synchronously* (all nodes of equal height are processed together,
children before parents) -- mathematically identical to the reference's post-order recursion because
siblings are independent (cf. the reference's own level-wise variant,
HSS/HSSMatrix.compress_stable.hpp:234-277).  This is the schedule the HIP engine uses, one
variable-size batched launch per step and height.

Reference files followed (all under /root/reference/src):
  HSS/HSSMatrix.cpp:60-70            tree construction (bisection m/2 | m-m/2 while m > leaf)
  HSS/HSSMatrix.compress_stable.hpp  adaptive stable compression (default algorithm)
  HSS/HSSMatrix.compress.hpp         original compression, local samples, reduce
  HSS/HSSBasisID.hpp                 interpolative basis U = P [I; E]
  dense/DenseMatrix.cpp:693-790      LQ, orthogonalize, ID_row / ID_column_GEQP3
  dense/lapack/dgeqp3tol.f           tolerance-truncated QRCP
  HSS/HSSMatrix.apply.hpp            mat-vec
  HSS/HSSMatrix.factor.hpp           ULV factorization
  HSS/HSSMatrix.solve.hpp            ULV solve
  misc/RandomWrapper.hpp             minstd_rand + normal_distribution sketching matrix
"""
import math

import numpy as np
import scipy.linalg as sla

UNTOUCHED, PARTIAL, COMPRESSED = 0, 1, 2


# ----------------------------------------------------------------------------------------------
# Random numbers: misc/RandomWrapper.hpp:128-191,238-241 -- std::minstd_rand(seed 0) feeding
# libstdc++'s std::normal_distribution<double> (Marsaglia polar method, bits/random.tcc).
# ----------------------------------------------------------------------------------------------
class MinstdNormal:
    """Bit-exact restatement of libstdc++ 11 `normal_distribution<double>(minstd_rand)`.

    minstd_rand: x <- 48271 x mod (2^31-1); seed 0 is mapped to 1 (bits/random.h
    linear_congruential_engine::seed).  generate_canonical<double,53> with a 31-bit-range engine
    draws k=2 values: sum = (x1-1) + (x2-1)*R, R = 2147483646, result = sum / R^2 (clamped below 1).
    normal_distribution: polar method, returns saved*... second value on the next call.
    """

    A, M = 48271, 2147483647

    def __init__(self, seed=0):
        s = seed % self.M
        self.x = 1 if s == 0 else s
        self.saved = None

    def _next(self):
        self.x = (self.x * self.A) % self.M
        return self.x

    def _canonical(self):
        R = 2147483646.0
        s = float(self._next() - 1)
        s += float(self._next() - 1) * R
        r = s / (R * R)
        if r >= 1.0:
            r = math.nextafter(1.0, 0.0)
        return r

    def get(self):
        if self.saved is not None:
            v = self.saved
            self.saved = None
            return v
        while True:
            x = 2.0 * self._canonical() - 1.0
            y = 2.0 * self._canonical() - 1.0
            r2 = x * x + y * y
            if not (r2 > 1.0 or r2 == 0.0):
                break
        mult = math.sqrt(-2.0 * math.log(r2) / r2)
        self.saved = x * mult
        return y * mult

    def matrix(self, rows, cols):
        """DenseMatrix::random: column-major serial fill (dense/DenseMatrix.cpp:172-181)."""
        out = np.empty((rows, cols), order="F")
        flat = out.reshape(-1, order="F")
        for i in range(rows * cols):
            flat[i] = self.get()
        return flat.reshape((rows, cols), order="F")


class SJLTGenerator:
    """The SJLT sketching matrix of --hss_compression_sketch SJLT (HSS/HSSMatrix.sketch.hpp:260-460): every ROW of an
    n x cols block gets nnz entries +-1 (SJLT_to_dense, :573-596, keeps them unscaled).  CHUNK (createSJLTCRS_Chunks,
    :419-441): the columns are cut into nnz chunks of cols // nnz, one nonzero at a uniform position of each chunk,
    sign by a fair coin.  PERM (createSJLTCRS, :316-341): the first nnz entries of a random permutation of the
    columns.  The first block (d0 + dd columns) uses nnz0, later blocks nnz (compress_stable.hpp:62-76).  The
    reference seeds its engine from the clock (:266-270), so only the distribution is restated, with numpy's
    generator."""

    def __init__(self, nnz0=4, nnz=4, algo="chunk", seed=0):
        self.nnz0, self.nnz, self.algo = nnz0, nnz, algo
        self.rng = np.random.default_rng(seed)
        self.first = True

    def matrix(self, rows, cols):
        nnz = min(self.nnz0 if self.first else self.nnz, cols)
        self.first = False
        R = np.zeros((rows, cols), order="F")
        sign = self.rng.integers(0, 2, size=(rows, nnz)) * 2.0 - 1.0
        if self.algo == "chunk":
            chunk = cols // nnz
            pos = self.rng.integers(0, chunk, size=(rows, nnz)) + chunk * np.arange(nnz)[None, :]
        else:
            pos = np.argsort(self.rng.random((rows, cols)), axis=1)[:, :nnz]
        R[np.arange(rows)[:, None], pos] = sign
        return R


# ----------------------------------------------------------------------------------------------
# Test-problem generators of test/test_HSS_seq.cpp:69-105
# ----------------------------------------------------------------------------------------------
def toeplitz(n, kind="T"):
    i = np.arange(n)
    A = 1.0 / (1.0 + np.abs(i[:, None] - i[None, :]))
    if kind == "U":
        A = np.triu(A)
    return np.asfortranarray(A)


def test_matrix(kind, n):
    """'T'/'U' Toeplitz (test_HSS_seq.cpp:69-91); 'L' = I + (1/n) U V^T where U and V are both
    filled by DenseMatrix::random(), i.e. each from a fresh seed-0 generator (:92-105), so U == V."""
    if kind in ("T", "U"):
        return toeplitz(n, kind)
    k = max(1, int(0.3 * n))
    U = MinstdNormal(0).matrix(n, k)
    return np.asfortranarray(np.eye(n) + (U @ U.T) / n)


# ----------------------------------------------------------------------------------------------
# Options: HSS/HSSOptions.hpp:465-490 (HSSOptions defaults)
# ----------------------------------------------------------------------------------------------
class Options:
    def __init__(self, rel_tol=1e-2, abs_tol=1e-8, leaf_size=512, d0=128, dd=64, p=10,
                 max_rank=50000, algorithm="stable"):
        self.rel_tol, self.abs_tol, self.leaf_size = rel_tol, abs_tol, leaf_size
        self.d0, self.dd, self.p, self.max_rank, self.algorithm = d0, dd, p, max_rank, algorithm


# ----------------------------------------------------------------------------------------------
# Tree: HSS/HSSMatrix.cpp:60-70
# ----------------------------------------------------------------------------------------------
class Node:
    __slots__ = ("lo", "m", "lvl", "ch", "parent", "height", "Ustate", "Vstate", "D", "B01",
                 "B10", "UE", "Uperm", "VE", "Vperm", "Jr", "Jc", "Ir", "Ic", "Ur_max", "Vr_max",
                 "Qr", "Qc", "ulv", "idx", "cols")

    def __init__(self, lo, m, lvl):
        self.lo, self.m, self.lvl = lo, m, lvl
        self.ch, self.parent, self.height = [], None, 0
        self.Ustate = self.Vstate = UNTOUCHED
        self.D = self.B01 = self.B10 = None
        self.UE = self.VE = None          # E blocks ((rows-rank) x rank)
        self.Uperm = self.Vperm = None    # 0-based: row k of [I;E] is local row perm[k]
        self.Jr = self.Jc = None          # local skeleton rows (= perm[:rank])
        self.Ir = self.Ic = None          # global skeleton indices
        self.Ur_max = self.Vr_max = 0.0
        self.Qr = self.Qc = None
        self.ulv = None

    @property
    def leaf(self):
        return not self.ch

    @property
    def rU(self):
        return 0 if self.UE is None else self.UE.shape[1]

    @property
    def rV(self):
        return 0 if self.VE is None else self.VE.shape[1]

    @property
    def Urows(self):
        return self.m if self.leaf else self.ch[0].rU + self.ch[1].rU

    @property
    def Vrows(self):
        return self.m if self.leaf else self.ch[0].rV + self.ch[1].rV

    @property
    def compressed(self):
        return self.Ustate == COMPRESSED and self.Vstate == COMPRESSED

    @property
    def untouched(self):
        return self.Ustate == UNTOUCHED and self.Vstate == UNTOUCHED


def build_tree(n, leaf_size):
    nodes = []

    def rec(lo, m, lvl, parent):
        nd = Node(lo, m, lvl)
        nd.parent = parent
        nd.idx = len(nodes)
        nodes.append(nd)
        if m > leaf_size:
            nd.ch = [rec(lo, m // 2, lvl + 1, nd), rec(lo + m // 2, m - m // 2, lvl + 1, nd)]
            nd.height = 1 + max(c.height for c in nd.ch)
        return nd

    root = rec(0, n, 0, None)
    return root, nodes  # nodes in pre-order


def build_tree_preorder(rows, is_leaf):
    """Tree from a pre-order node table (rows per node, leaf flag): the cluster trees of the kernel front end are
    not bisection trees (binary_tree_clustering, clustering/Clustering.hpp:143-168)."""
    nodes = []
    pos = [0]

    def rec(lo, lvl, parent):
        k = pos[0]
        pos[0] += 1
        nd = Node(lo, int(rows[k]), lvl)
        nd.parent = parent
        nd.idx = len(nodes)
        nodes.append(nd)
        if not is_leaf[k]:
            a = rec(lo, lvl + 1, nd)
            b = rec(lo + a.m, lvl + 1, nd)
            nd.ch = [a, b]
            nd.height = 1 + max(a.height, b.height)
        return nd

    root = rec(0, 0, None)
    assert pos[0] == len(rows) and root.m == sum(r for r, l in zip(rows, is_leaf) if l)
    return root, nodes


def kernel_function(ktype, h, p=1):
    """kernel::GaussKernel / LaplaceKernel / ANOVAKernel::eval_kernel_function (kernel/Kernel.hpp:333-399), blockwise:
    returns f(XI, XJ) -> |I| x |J| for point blocks (rows = points)."""
    def f(xi, xj):
        df = xi[:, None, :] - xj[None, :, :]
        if ktype == 0:
            return np.exp(-(df ** 2).sum(-1) / (2.0 * h * h))
        if ktype == 1:
            return np.exp(-np.abs(df).sum(-1) / h)
        t = np.exp(-(df ** 2) / (2.0 * h * h))
        Kss = [(t ** (j + 1)).sum(-1) for j in range(p)]
        Kpp = [np.ones(t.shape[:2])]
        for i in range(1, p + 1):
            Kpp.append(sum((-1.0) ** (s_ + 1) * Kpp[i - s_] * Kss[s_ - 1] for s_ in range(1, i + 1)) / i)
        return Kpp[p]
    return f


# ----------------------------------------------------------------------------------------------
# Dense kernels on the path
# ----------------------------------------------------------------------------------------------
def id_row(S, rtol, atol, max_rank):
    """Row interpolative decomposition, dense/DenseMatrix.cpp:746-790 + dgeqp3tol.f:203-232.

    S (m x d).  Returns E ((m-r) x r), perm (m,), with  S[perm] ~= [I; E] S[perm[:r]].
    QRCP of S^T stops at the first c with |R_cc|/|R_00| <= rtol or |R_cc| <= atol.
    """
    m, d = S.shape
    if m == 0 or d == 0:
        return np.zeros((m, 0)), np.arange(m)
    R, jpvt = sla.qr(S.T, mode="r", pivoting=True)
    k = min(m, d)
    diag = np.abs(np.diag(R)[:k])
    rank = k
    for c in range(k):
        # 0/0 = NaN compares false like the Fortran test, then |R_cc| <= atol decides
        ratio = diag[c] / diag[0] if diag[0] != 0.0 else float("nan")
        if ratio <= rtol or diag[c] <= atol:
            rank = c
            break
    rank = min(rank, max_rank)
    X = sla.solve_triangular(R[:rank, :rank], R[:rank, rank:], lower=False) if rank else \
        np.zeros((0, m))
    return np.ascontiguousarray(X.T), np.asarray(jpvt, dtype=np.int64)


def basis_apply(E, perm, b):
    """U b with U = P [I; E]  (HSSBasisID::apply, HSS/HSSBasisID.hpp:155-186)."""
    r = E.shape[1]
    c = np.empty((len(perm), b.shape[1]))
    c[perm[:r]] = b
    c[perm[r:]] = E @ b
    return c


def basis_applyC(E, perm, b):
    """U^H b  (HSSBasisID::applyC, HSS/HSSBasisID.hpp:189-203)."""
    r = E.shape[1]
    pb = b[perm]
    return pb[:r] + E.T @ pb[r:]


def basis_dense(E, perm):
    r = E.shape[1]
    return basis_apply(E, perm, np.eye(r))


# ----------------------------------------------------------------------------------------------
# The HSS matrix
# ----------------------------------------------------------------------------------------------
class HSSMatrix:
    """Restatement of strumpack::HSS::HSSMatrix<double> (HSS/HSSMatrix.hpp:79-711)."""

    def __init__(self, A=None, opts=None, n=None, Amult=None, Aelem=None, rgen=None):
        self.opts = opts or Options()
        if A is not None:
            A = np.asarray(A)
            n = A.shape[0]
            # AFunctor, HSS/HSSExtra.hpp:231-248
            Amult = lambda Rr, Rc: (A @ Rr, A.T @ Rc)
            Aelem = lambda I, J: A[np.ix_(I, J)]
        self.n = n
        self.root, self.nodes = build_tree(n, self.opts.leaf_size)
        self.by_height = {}
        for nd in self.nodes:
            self.by_height.setdefault(nd.height, []).append(nd)
        self.rounds = 0
        self.d_final = 0
        if Amult is not None:
            self.compress(Amult, Aelem, rgen or MinstdNormal(0))

    # -- kernel matrices: compression from coordinates, no random sketch --------------------------
    @classmethod
    def from_kernel(cls, X, kfun, lam, tree_rows, tree_leaf, ann, opts):
        """HSSMatrix::compress_recursive_ann / compute_local_samples_ann / compute_U_V_bases_ann
        (HSS/HSSMatrix.compress_kernel.hpp:84-293) for ONE neighbour count (the caller doubles it on failure, :75).
        X: n x d points in cluster order; kfun from kernel_function(); ann: n x k neighbour ids (cluster order)."""
        H = cls(n=X.shape[0], opts=opts)
        H.root, H.nodes = build_tree_preorder(tree_rows, tree_leaf)
        H.by_height = {}
        for nd in H.nodes:
            H.by_height.setdefault(nd.height, []).append(nd)

        def K(I, J):
            I, J = np.asarray(I, dtype=np.int64), np.asarray(J, dtype=np.int64)
            return kfun(X[I], X[J]) + lam * (I[:, None] == J[None, :])

        o = opts
        for h in sorted(H.by_height):
            for nd in H.by_height[h]:
                lo, hi = nd.lo, nd.lo + nd.m
                if nd.leaf:
                    I = np.arange(lo, hi)
                    nd.D = K(I, I)
                    ids = ann[lo:hi].ravel()
                else:
                    a, b = nd.ch
                    if not (a.compressed and b.compressed):
                        continue
                    nd.B01 = K(a.Ir, b.Ic)
                    nd.B10 = nd.B01.T.copy()
                    I = np.concatenate([a.Ir, b.Ir])
                    ids = np.concatenate([a.cols, b.cols])
                if nd.lvl == 0:
                    nd.Ustate = nd.Vstate = COMPRESSED
                    continue
                ids = ids[(ids >= 0) & ((ids < lo) | (ids >= hi))]
                nd.cols = np.unique(ids)                       # sorted, duplicates dropped (:197-210)
                S = K(I, nd.cols)
                E, perm = id_row(S, o.rel_tol / nd.lvl, o.abs_tol / nd.lvl, o.max_rank)
                d, r = len(nd.cols), E.shape[1]
                if not (d >= nd.m or d >= o.max_rank or r + o.p < d):   # :262-272
                    continue
                H._set_basis(nd, "U", E, perm)
                H._set_basis(nd, "V", E, perm)
        return H

    # -- introspection (HSS/HSSMatrix.cpp:197-331) ---------------------------------------------
    def is_compressed(self):
        return self.root.compressed

    def levels(self):
        return 1 + self.root.height

    def rank(self):
        return max(max(nd.rU, nd.rV) for nd in self.nodes)

    def nonzeros_payload(self):
        """Stored scalars + permutation entries (without the reference's sizeof(*this) terms)."""
        t = 0
        for nd in self.nodes:
            for M in (nd.D, nd.B01, nd.B10, nd.UE, nd.VE):
                if M is not None:
                    t += M.size
            for P in (nd.Uperm, nd.Vperm):
                if P is not None:
                    t += len(P)
        return t

    # -- compression ---------------------------------------------------------------------------
    def compress(self, Amult, Aelem, rgen):
        if self.opts.algorithm == "original":
            self._compress_original(Amult, Aelem, rgen)
        else:
            self._compress_stable(Amult, Aelem, rgen)

    def _compress_stable(self, Amult, Aelem, rgen):
        """compress_stable(Amult, Aelem, opts), HSS/HSSMatrix.compress_stable.hpp:100-163."""
        o, n = self.opts, self.n
        d, dd = o.d0, o.dd
        Rr = Rc = Sr = Sc = np.zeros((n, 0), order="F")
        while not self.is_compressed():
            c = 0 if d == o.d0 else d
            dnew = d + dd if d == o.d0 else dd
            Rnew = rgen.matrix(n, dnew)
            Srn, Scn = Amult(Rnew, Rnew)
            # DenseMatrix::resize is content-preserving (dense/DenseMatrix.cpp:229-244): columns
            # [0,c) keep the in-place processed samples of the earlier rounds.
            Rr = np.asfortranarray(np.hstack([Rr[:, :c], Rnew]))
            Rc = np.asfortranarray(np.hstack([Rc[:, :c], Rnew]))
            Sr = np.asfortranarray(np.hstack([Sr[:, :c], Srn]))
            Sc = np.asfortranarray(np.hstack([Sc[:, :c], Scn]))
            self.rounds += 1
            for h in sorted(self.by_height):
                for nd in self.by_height[h]:
                    self._compress_node_stable(nd, Rr, Rc, Sr, Sc, Aelem, d, dd)
            self.d_final = d + dd
            if not self.is_compressed():
                d += dd
                dd = min(dd, o.max_rank - d)
                if dd <= 0:
                    raise RuntimeError("max_rank reached without convergence")

    def _extract_blocks(self, nd, Aelem):
        """D / B01 / B10 extraction, compress_stable.hpp:171-182,204-217."""
        if nd.leaf:
            I = np.arange(nd.lo, nd.lo + nd.m)
            nd.D = np.array(Aelem(I, I), order="F")
        else:
            c0, c1 = nd.ch
            nd.B01 = np.array(Aelem(c0.Ir, c1.Ic), order="F").reshape(c0.rU, c1.rV)
            nd.B10 = np.array(Aelem(c1.Ir, c0.Ic), order="F").reshape(c1.rU, c0.rV)

    def _compress_node_stable(self, nd, Rr, Rc, Sr, Sc, Aelem, d, dd):
        """One node of compress_recursive_stable, compress_stable.hpp:165-232."""
        if not nd.leaf and not (nd.ch[0].compressed and nd.ch[1].compressed):
            return
        if nd.untouched:
            self._extract_blocks(nd, Aelem)
        if nd.lvl == 0:
            nd.Ustate = nd.Vstate = COMPRESSED
            return
        if nd.untouched:
            self._local_samples(nd, Rr, Rc, Sr, Sc, 0, d + dd)
        else:
            self._local_samples(nd, Rr, Rc, Sr, Sc, d, dd)
        if not nd.compressed:
            self._basis_stable(nd, Sr, d, dd, "U")
            self._basis_stable(nd, Sc, d, dd, "V")
            if nd.compressed:
                self._reduce_samples(nd, Rr, Rc, 0, d + dd)
        else:
            self._reduce_samples(nd, Rr, Rc, d, dd)

    def _local_samples(self, nd, Rr, Rc, Sr, Sc, c0, dc):
        """compute_local_samples, HSS/HSSMatrix.compress.hpp:524-629."""
        cs = slice(c0, c0 + dc)
        o = nd.lo
        if nd.leaf:
            Sr[o:o + nd.m, cs] -= nd.D @ Rr[o:o + nd.m, cs]
            Sc[o:o + nd.m, cs] -= nd.D.T @ Rc[o:o + nd.m, cs]
        else:
            a, b = nd.ch
            t0 = Sr[a.lo + a.Jr, cs].copy()
            t1 = Sr[b.lo + b.Jr, cs].copy()
            Sr[o:o + a.rU, cs] = t0 - nd.B01 @ Rr[b.lo:b.lo + b.rV, cs]
            Sr[o + a.rU:o + a.rU + b.rU, cs] = t1 - nd.B10 @ Rr[a.lo:a.lo + a.rV, cs]
            t0 = Sc[a.lo + a.Jc, cs].copy()
            t1 = Sc[b.lo + b.Jc, cs].copy()
            Sc[o:o + a.rV, cs] = t0 - nd.B10.T @ Rc[b.lo:b.lo + b.rU, cs]
            Sc[o + a.rV:o + a.rV + b.rV, cs] = t1 - nd.B01.T @ Rc[a.lo:a.lo + a.rU, cs]

    def _update_orthogonal_basis(self, nd, which, S, d, dd, untouched):
        """update_orthogonal_basis, compress_stable.hpp:390-442.  True => rank is resolved."""
        o = self.opts
        m = S.shape[0]
        if d >= m:
            return True
        Q = getattr(nd, "Qr" if which == "U" else "Qc")
        Qn = np.zeros((m, d + dd), order="F")
        if Q is not None:
            Qn[:, :Q.shape[1]] = Q
        Q = Qn
        Q[:, d:d + dd] = S[:, d:d + dd]
        if untouched:
            c2 = slice(0, min(d, m))
            Q[:, :d] = S[:, :d]
        else:
            c2 = slice(d - dd, d - dd + min(dd, m - (d - dd)))
        c12 = slice(0, min(d, m))
        # DenseMatrix::orthogonalize (dense/DenseMatrix.cpp:721-744): geqrf + orgqr
        Q2 = Q[:, c2]
        minmn = min(Q2.shape)
        qq, rr = sla.qr(Q2[:, :minmn], mode="economic")
        dg = np.abs(np.diag(rr))
        r_max, r_min = dg.max(), dg.min()
        Q2[:, :minmn] = qq
        Q2[:, minmn:] = 0.0
        if untouched:
            setattr(nd, "Ur_max" if which == "U" else "Vr_max", r_max)
        r_max_0 = nd.Ur_max if which == "U" else nd.Vr_max
        setattr(nd, "Qr" if which == "U" else "Qc", Q)
        atol, rtol = o.abs_tol / nd.lvl, o.rel_tol / nd.lvl
        if abs(r_min) < atol or abs(r_min / r_max_0) < rtol:
            return True
        Q12 = Q[:, c12]
        Q3 = Q[:, d:d + dd]
        pc = min(dd, o.p)
        S3norm = np.linalg.norm(Q3[:, :pc])
        for _ in range(2):  # iterated classical Gram-Schmidt
            Q3 -= Q12 @ (Q12.T @ Q3)
        Q3norm = np.linalg.norm(Q3[:, :pc])
        return (Q3norm / math.sqrt(float(dd)) < atol) or (Q3norm / S3norm < rtol)

    def _basis_stable(self, nd, S, d, dd, which):
        """compute_{U,V}_basis_stable, compress_stable.hpp:280-348."""
        o = self.opts
        state = nd.Ustate if which == "U" else nd.Vstate
        if state == COMPRESSED:
            return
        rows = nd.Urows if which == "U" else nd.Vrows
        lS = S[nd.lo:nd.lo + rows, :d + dd]
        if (d + dd >= o.max_rank or d + dd >= rows or
                self._update_orthogonal_basis(nd, which, lS, d, dd, state == UNTOUCHED)):
            setattr(nd, "Qr" if which == "U" else "Qc", None)
            E, perm = id_row(lS, o.rel_tol / nd.lvl, o.abs_tol / nd.lvl, o.max_rank)
            self._set_basis(nd, which, E, perm)
        else:
            if which == "U":
                nd.Ustate = PARTIAL
            else:
                nd.Vstate = PARTIAL

    def _set_basis(self, nd, which, E, perm):
        r = E.shape[1]
        J = perm[:r].copy()
        if nd.leaf:
            I = nd.lo + J
        else:
            a, b = nd.ch
            ia = a.Ir if which == "U" else a.Ic
            ib = b.Ir if which == "U" else b.Ic
            r0 = len(ia)
            I = np.array([ia[j] if j < r0 else ib[j - r0] for j in J], dtype=np.int64)
        if which == "U":
            nd.UE, nd.Uperm, nd.Jr, nd.Ir, nd.Ustate = E, perm, J, I, COMPRESSED
        else:
            nd.VE, nd.Vperm, nd.Jc, nd.Ic, nd.Vstate = E, perm, J, I, COMPRESSED

    def _reduce_samples(self, nd, Rr, Rc, c0, dc):
        """reduce_local_samples, HSS/HSSMatrix.compress.hpp:689-724."""
        cs = slice(c0, c0 + dc)
        o = nd.lo
        if nd.leaf:
            wr, wc = Rr[o:o + nd.m, cs], Rc[o:o + nd.m, cs]
        else:
            a, b = nd.ch
            wr = np.vstack([Rr[a.lo:a.lo + a.rV, cs], Rr[b.lo:b.lo + b.rV, cs]])
            wc = np.vstack([Rc[a.lo:a.lo + a.rU, cs], Rc[b.lo:b.lo + b.rU, cs]])
        Rr[o:o + nd.rV, cs] = basis_applyC(nd.VE, nd.Vperm, wr)
        Rc[o:o + nd.rU, cs] = basis_applyC(nd.UE, nd.Uperm, wc)

    def _compress_original(self, Amult, Aelem, rgen):
        """compress_original, HSS/HSSMatrix.compress.hpp:100-165,300-368,631-687."""
        o, n = self.opts, self.n
        d_old, d = 0, o.d0 + o.p
        Rr = Rc = Sr = Sc = np.zeros((n, 0), order="F")
        while not self.is_compressed():
            Rnew = rgen.matrix(n, d - d_old)
            Srn, Scn = Amult(Rnew, Rnew)
            Rr = np.asfortranarray(np.hstack([Rr, Rnew]))
            Rc = np.asfortranarray(np.hstack([Rc, Rnew]))
            Sr = np.asfortranarray(np.hstack([Sr, Srn]))
            Sc = np.asfortranarray(np.hstack([Sc, Scn]))
            self.rounds += 1
            dd = d - d_old
            for h in sorted(self.by_height):
                for nd in self.by_height[h]:
                    if not nd.leaf and not (nd.ch[0].compressed and nd.ch[1].compressed):
                        continue
                    if nd.untouched:
                        self._extract_blocks(nd, Aelem)
                    if nd.lvl == 0:
                        nd.Ustate = nd.Vstate = COMPRESSED
                        continue
                    if nd.untouched:
                        self._local_samples(nd, Rr, Rc, Sr, Sc, 0, d)
                    else:
                        self._local_samples(nd, Rr, Rc, Sr, Sc, d - dd, dd)
                    if not nd.compressed:
                        rt, at = o.rel_tol / nd.lvl, o.abs_tol / nd.lvl
                        EU, pU = id_row(Sr[nd.lo:nd.lo + nd.Urows, :d], rt, at, o.max_rank)
                        EV, pV = id_row(Sc[nd.lo:nd.lo + nd.Vrows, :d], rt, at, o.max_rank)
                        if d - o.p >= o.max_rank or (EU.shape[1] < d - o.p and
                                                     EV.shape[1] < d - o.p):
                            self._set_basis(nd, "U", EU, pU)
                            self._set_basis(nd, "V", EV, pV)
                            self._reduce_samples(nd, Rr, Rc, 0, d)
                        else:
                            nd.Ustate = nd.Vstate = PARTIAL
                    else:
                        self._reduce_samples(nd, Rr, Rc, d - dd, dd)
            self.d_final = d
            if not self.is_compressed():
                d_old = d
                d = 2 * (d_old - o.p) + o.p

    # -- apply: HSS/HSSMatrix.apply.hpp:55-220 ---------------------------------------------------
    def subtree(self, top):
        """Nodes of the sub-tree of `top` in pre-order (a child of an HSS matrix is an HSS matrix, HSSMatrix.hpp:194)."""
        out, stack = [], [top]
        while stack:
            nd = stack.pop()
            out.append(nd)
            stack.extend(reversed(nd.ch))
        return out

    def mult(self, x, trans="N", beta=0.0, y=None, top=None):
        """op(H) x; top = a node: the diagonal block of that node (child(c)->apply, HSSMatrix.apply.hpp:55-220)."""
        if top is not None and top is not self.root:
            return self._mult_sub(top, x, trans)
        x = np.asarray(x, dtype=np.float64).reshape(self.n, -1)
        T = trans in ("T", "C", "t", "c")
        tmp1, tmp2 = {}, {}
        for h in sorted(self.by_height):          # apply_fwd / applyT_fwd (up-sweep)
            for nd in self.by_height[h]:
                if nd.lvl == 0:
                    continue
                E, perm = (nd.UE, nd.Uperm) if T else (nd.VE, nd.Vperm)
                if nd.leaf:
                    b = x[nd.lo:nd.lo + nd.m]
                else:
                    b = np.vstack([tmp1[nd.ch[0].idx], tmp1[nd.ch[1].idx]])
                tmp1[nd.idx] = basis_applyC(E, perm, b)
        out = np.zeros_like(x) if y is None else beta * np.asarray(y, dtype=np.float64).reshape(
            self.n, -1)
        for nd in self.nodes:                     # pre-order = parents first (down-sweep)
            E, perm = (nd.VE, nd.Vperm) if T else (nd.UE, nd.Uperm)
            r = 0 if E is None else E.shape[1]
            if nd.leaf:
                D = nd.D.T if T else nd.D
                out[nd.lo:nd.lo + nd.m] += D @ x[nd.lo:nd.lo + nd.m]
                if r and nd.lvl != 0:
                    out[nd.lo:nd.lo + nd.m] += basis_apply(E, perm, tmp2[nd.idx])
            else:
                a, b = nd.ch
                if T:
                    t0 = nd.B10.T @ tmp1[b.idx]
                    t1 = nd.B01.T @ tmp1[a.idx]
                    ra = a.rV
                else:
                    t0 = nd.B01 @ tmp1[b.idx]
                    t1 = nd.B10 @ tmp1[a.idx]
                    ra = a.rU
                if nd.lvl != 0 and r:
                    t = basis_apply(E, perm, tmp2[nd.idx])
                    t0 = t0 + t[:ra]
                    t1 = t1 + t[ra:]
                tmp2[a.idx], tmp2[b.idx] = t0, t1
        return out

    def _mult_sub(self, top, x, trans):
        T = trans in ("T", "C", "t", "c")
        x = np.asarray(x, dtype=np.float64).reshape(top.m, -1)
        nodes = self.subtree(top)
        tmp1, tmp2 = {}, {}
        for nd in sorted(nodes, key=lambda q: q.height):
            if nd is top:
                continue
            E, perm = (nd.UE, nd.Uperm) if T else (nd.VE, nd.Vperm)
            b = x[nd.lo - top.lo:nd.lo - top.lo + nd.m] if nd.leaf else np.vstack([tmp1[nd.ch[0].idx], tmp1[nd.ch[1].idx]])
            tmp1[nd.idx] = basis_applyC(E, perm, b)
        out = np.zeros_like(x)
        for nd in nodes:
            E, perm = (nd.VE, nd.Vperm) if T else (nd.UE, nd.Uperm)
            r = 0 if (E is None or nd is top) else E.shape[1]
            lo = nd.lo - top.lo
            if nd.leaf:
                out[lo:lo + nd.m] += (nd.D.T if T else nd.D) @ x[lo:lo + nd.m]
                if r:
                    out[lo:lo + nd.m] += basis_apply(E, perm, tmp2[nd.idx])
            else:
                a, b = nd.ch
                t0 = (nd.B10.T if T else nd.B01) @ tmp1[b.idx]
                t1 = (nd.B01.T if T else nd.B10) @ tmp1[a.idx]
                if r:
                    t = basis_apply(E, perm, tmp2[nd.idx])
                    ra = a.rV if T else a.rU
                    t0, t1 = t0 + t[:ra], t1 + t[ra:]
                tmp2[a.idx], tmp2[b.idx] = t0, t1
        return out

    def dense(self):
        return self.mult(np.eye(self.n))

    # -- Schur complement of the (0,0) block: HSS/HSSMatrix.Schur.hpp, factor.hpp:43-49 ---------------------------
    def apply_UV_big(self, top, Uop=None, Vop=None):
        """Theta = Ubig Uop, Phi = Vbig Vop over the sub-tree of `top` (apply_UV_big, Schur.hpp:254-323)."""
        res = []
        for op, useU in ((Uop, True), (Vop, False)):
            if op is None:
                res.append(None)
                continue
            out = np.zeros((top.m, op.shape[1]))
            cur = {top.idx: op}
            for nd in self.subtree(top):
                E, perm = (nd.UE, nd.Uperm) if useU else (nd.VE, nd.Vperm)
                t = basis_apply(E, perm, cur[nd.idx]) if E is not None and E.shape[1] and op.shape[1] else \
                    np.zeros(((nd.Urows if useU else nd.Vrows), op.shape[1]))
                if nd.leaf:
                    out[nd.lo - top.lo:nd.lo - top.lo + nd.m] = t
                else:
                    a, b = nd.ch
                    ra = a.rU if useU else a.rV
                    cur[a.idx], cur[b.idx] = t[:ra], t[ra:]
            res.append(out)
        return res

    def apply_UtVt_big(self, top, A):
        """(Ubig^T A, Vbig^T A) over the sub-tree of `top` (apply_UtVt_big, Schur.hpp:223-252)."""
        res = []
        for useU in (True, False):
            tmp = {}
            for nd in sorted(self.subtree(top), key=lambda q: q.height):
                E, perm = (nd.UE, nd.Uperm) if useU else (nd.VE, nd.Vperm)
                b = A[nd.lo - top.lo:nd.lo - top.lo + nd.m] if nd.leaf else np.vstack([tmp[nd.ch[0].idx], tmp[nd.ch[1].idx]])
                tmp[nd.idx] = basis_applyC(E, perm, b)
            res.append(tmp[top.idx])
        return res

    def partial_factor(self):
        """ULV of child(0) alone, as the root of its sub-tree, keeping Vhat (partial_factor, factor.hpp:43-49, :96-104,
        :113-114)."""
        self.factor(top=self.root.ch[0], partial=True)

    def schur_update(self):
        """-> Theta = U1big B10, DUB01 = D00^{-1} U0 B01, Phi = (D00^{-1} U0 B01 V1big^T)^T, Vhat  (Schur_update,
        Schur.hpp:40-59; Vhat = child(0)->ULV().Vhat(), HSSExtra.hpp:191)."""
        c0, c1 = self.root.ch
        UB = basis_apply(c0.UE, c0.Uperm, self.root.B01) if c0.rU else np.zeros((c0.Urows, self.root.B01.shape[1]))
        DUB01 = sla.lu_solve(c0.ulv["LU"], UB) if UB.size else UB
        Theta, Phi = self.apply_UV_big(c1, self.root.B10, DUB01.T.copy())
        return Theta, DUB01, Phi, c0.ulv["Vhat"]

    def schur_product_direct(self, Theta, DUB01, Phi, Vhat, R):
        """Sr = S R, Sc = S^T R with S = H11 - Theta Vhat^T Phi^T (Schur_product_direct, Schur.hpp:60-143)."""
        c1 = self.root.ch[1]
        U1tR, V1tR = self.apply_UtVt_big(c1, R)
        Sr = self.mult(R, "N", top=c1) - Theta @ ((Vhat.T @ DUB01) @ V1tR)
        Sc = self.mult(R, "T", top=c1) - Phi @ (Vhat @ (self.root.B10.T @ U1tR))
        return Sr, Sc

    def shift(self, sigma):
        """HSSMatrix::shift, HSS/HSSMatrix.cpp:359-365 (ULV factors become stale)."""
        for nd in self.nodes:
            if nd.leaf:
                nd.D[np.diag_indices(nd.m)] += sigma
            nd.ulv = None

    # -- ULV factorization: HSS/HSSMatrix.factor.hpp:51-147 -------------------------------------
    def factor(self, top=None, partial=False):
        work = {}
        top = top or self.root
        inside = {nd.idx for nd in self.subtree(top)}
        for h in sorted(self.by_height):
            for nd in self.by_height[h]:
                if nd.idx not in inside:
                    continue
                f = {}
                isroot = nd is top
                if not nd.leaf:
                    a, b = nd.ch
                    Dt0, Vt10 = work.pop(a.idx)
                    Dt1, Vt11 = work.pop(b.idx)
                    u = a.rU + b.rU
                    Dh = np.zeros((u, u))
                    Dh[:a.rU, :a.rU] = Dt0
                    Dh[a.rU:, a.rU:] = Dt1
                    Dh[:a.rU, a.rU:] = nd.B01 @ Vt11.T
                    Dh[a.rU:, :a.rU] = nd.B10 @ Vt10.T
                    if not isroot or partial:
                        V = basis_dense(nd.VE, nd.Vperm)
                        Vh = np.vstack([Vt10 @ V[:a.rV], Vt11 @ V[a.rV:]])
                else:
                    Dh = nd.D.copy()
                    if not isroot or partial:
                        Vh = basis_dense(nd.VE, nd.Vperm)
                if isroot:
                    f["LU"] = sla.lu_factor(Dh) if Dh.size else None
                    if partial:
                        f["Vhat"] = Vh
                else:
                    m, r = len(nd.Uperm), nd.rU
                    PD = Dh[nd.Uperm]                                # P^T D
                    if m > r:
                        W1 = PD[:r].copy()
                        W0 = PD[r:] - nd.UE @ W1
                        # DenseMatrix::LQ (dense/DenseMatrix.cpp:693-719): W0 = [L 0] Q
                        qt, rt = sla.qr(W0.T, mode="full")
                        Q = qt.T
                        L = rt[:m - r].T
                        f.update(W1=W1, L=np.ascontiguousarray(L), Q=np.ascontiguousarray(Q),
                                 Vt0=Q[:m - r] @ Vh)
                        work[nd.idx] = (W1 @ Q[m - r:].T, Q[m - r:] @ Vh)
                    else:
                        work[nd.idx] = (PD, Vh)
                nd.ulv = f

    # -- ULV solve: HSS/HSSMatrix.solve.hpp:69-238 ----------------------------------------------
    def solve(self, b):
        b = np.array(b, dtype=np.float64).reshape(self.n, -1)
        ft1, y, z, xs = {}, {}, {}, {}
        nrhs = b.shape[1]
        for h in sorted(self.by_height):          # solve_fwd
            for nd in self.by_height[h]:
                if nd.leaf:
                    f = b[nd.lo:nd.lo + nd.m].copy()
                else:
                    c0, c1 = nd.ch
                    f0 = ft1[c0.idx] - nd.B01 @ z[c1.idx]
                    f1 = ft1[c1.idx] - nd.B10 @ z[c0.idx]
                    for c, which in ((c0, 0), (c1, 1)):
                        mc, rc = len(c.Uperm), c.rU
                        if mc > rc:
                            t = c.ulv["W1"] @ (c.ulv["Q"][:mc - rc].T @ y[c.idx])
                            if which == 0:
                                f0 = f0 - t
                            else:
                                f1 = f1 - t
                    f = np.vstack([f0, f1])
                if nd.lvl == 0:
                    xs[nd.idx] = sla.lu_solve(nd.ulv["LU"], f) if f.size else f
                    continue
                f = f[nd.Uperm]
                m, r = len(nd.Uperm), nd.rU
                ft1[nd.idx] = f[:r]
                zc = None if nd.leaf else basis_applyC(
                    nd.VE, nd.Vperm, np.vstack([z[nd.ch[0].idx], z[nd.ch[1].idx]]))
                if m > r:
                    yy = sla.solve_triangular(nd.ulv["L"], f[r:] - nd.UE @ f[:r], lower=True)
                    y[nd.idx] = yy
                    zz = nd.ulv["Vt0"].T @ yy
                    z[nd.idx] = zz if zc is None else zc + zz
                else:
                    y[nd.idx] = np.zeros((0, nrhs))
                    z[nd.idx] = np.zeros((nd.rV, nrhs)) if zc is None else zc
        out = np.empty_like(b)
        for nd in self.nodes:                     # solve_bwd (pre-order)
            x = xs[nd.idx]
            if nd.leaf:
                out[nd.lo:nd.lo + nd.m] = x
                continue
            off = 0
            for c in nd.ch:
                mc, rc = len(c.Uperm), c.rU
                xc = x[off:off + rc]
                off += rc
                if mc > rc:
                    xs[c.idx] = c.ulv["Q"].T @ np.vstack([y[c.idx], xc])
                else:
                    xs[c.idx] = xc
        return out
