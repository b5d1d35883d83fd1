"""TEST INFRASTRUCTURE: frontal matrices of the 3D 7-point Poisson problem (BASELINE configs[4]: test_sparse_seq on a
k^3 grid with --sp_compression BLR), in closed form, for the BLR partial-factorization tests and bench.py --workload blr_front.

A front of a nested-dissection factorization with planar separators: the separator is an n x n plane of an n x n x L slab
whose two end planes belong to ancestors' separators (the front's update part) and whose lateral faces are the domain
boundary; everything between the planes has been eliminated.  F = A_SS - A_SI A_II^{-1} A_IS over S = separator + update
planes is exact and independent of the elimination order below, so it is what the multifrontal driver
(sparse/fronts/FrontBLR.cpp:405-431: extract_front + extend_add of the children's contribution blocks) assembles:

    F11 = T - G_pp(left) - G_11(right)       F12 = [-G_p1(left), -G_1p(right)]       F21 = F12^T
    F22 = blockdiag(-G_11(left), -G_pp(right))   (children's contributions only, as in a front before its parent assembles)

with T the plane's 5-point block (diagonal 6) and G = tridiag_p(-I, T, -I)^{-1} the inverse of a chain of p eliminated planes.
All blocks are functions of T, i.e. diagonal in the 2D sine basis: per eigenvalue lambda = 2 cosh(theta) of T the chain
inverse is g_11 = g_pp = sinh(p theta) / sinh((p+1) theta), g_1p = sinh(theta) / sinh((p+1) theta).

Separator nodes are ordered by recursive coordinate bisection down to `leaf` nodes per tile (the reference partitions the
separator's graph the same way, sparse/fronts/FrontBLR.cpp:622-640); the update part is cut into consecutive tiles of `leaf`
(FrontBLR.cpp:660-664)."""
import numpy as np


def plane_order(n, leaf, ny=None):
    """recursive coordinate bisection of the n x ny plane (ny = n by default): permutation (new -> old, old = ix * ny + iy) and
    the tile sizes"""
    ny = n if ny is None else ny
    perm, tiles, boxes = [], [], []

    def rec(x0, x1, y0, y1):
        cnt = (x1 - x0) * (y1 - y0)
        if cnt <= leaf or cnt <= 1:
            xs, ys = np.meshgrid(np.arange(x0, x1), np.arange(y0, y1), indexing="ij")
            perm.extend((xs * ny + ys).ravel().tolist())
            tiles.append(cnt)
            boxes.append((x0, x1, y0, y1))
            return
        if x1 - x0 >= y1 - y0:
            xm = x0 + (x1 - x0) // 2
            rec(x0, xm, y0, y1)
            rec(xm, x1, y0, y1)
        else:
            ym = y0 + (y1 - y0) // 2
            rec(x0, x1, y0, ym)
            rec(x0, x1, ym, y1)

    rec(0, n, 0, ny)
    return np.array(perm), tiles, boxes


def _chain(lmb, p):
    th = np.arccosh(lmb / 2.0)
    den = 1.0 - np.exp(-2.0 * (p + 1) * th)
    g11 = np.exp(-th) * (1.0 - np.exp(-2.0 * p * th)) / den
    g1p = np.exp(-p * th) * (1.0 - np.exp(-2.0 * th)) / den
    return g11, g1p


def poisson_front(n, p_left, p_right, leaf, upd="both", unsym=False, matmul=None, ny=None):
    """-> dict(F11, F12, F21, F22 (Fortran order), tiles1, tiles2, boxes); separator = an n x ny plane (ny = n by default),
    dsep = n ny, dupd = n ny per update plane.
    unsym: rows and columns scaled by two different smooth positive diagonals (same rank structure, F21 != F12^T).
    matmul(A, B): optional replacement of A @ B (bench.py hands in a torch product on the device for large planes)."""
    mm = matmul or (lambda a, b: a @ b)
    ny = n if ny is None else ny

    def sine(m):
        k = np.arange(1, m + 1)
        return np.sqrt(2.0 / (m + 1)) * np.sin(np.outer(k, k) * np.pi / (m + 1)), 2.0 * np.cos(k * np.pi / (m + 1))

    Sx, cx = sine(n)
    Sy, cy = sine(ny)
    Q = np.kron(Sx, Sy)
    lmb = (6.0 - cx[:, None] - cy[None, :]).ravel()
    perm, tiles1, boxes = plane_order(n, leaf, ny)
    Qp = np.ascontiguousarray(Q[perm, :])
    fun = lambda f: mm(Qp * f[None, :], Qp.T)
    gl11, gl1p = _chain(lmb, p_left)
    gr11, gr1p = _chain(lmb, p_right)
    F11 = fun(lmb - gl11 - gr11)
    sides = {"both": ("L", "R"), "left": ("L",), "right": ("R",), "none": ()}[upd]
    blocks12, blocks22 = [], []
    for s in sides:
        blocks12.append(fun(-(gl1p if s == "L" else gr1p)))
        blocks22.append(fun(-(gl11 if s == "L" else gr11)))
    ds = n * ny
    du = ds * len(sides)
    F12 = np.concatenate(blocks12, axis=1) if sides else np.zeros((ds, 0))
    F22 = np.zeros((du, du))
    for q, b in enumerate(blocks22):
        F22[q * ds:(q + 1) * ds, q * ds:(q + 1) * ds] = b
    F21 = F12.T.copy()
    if unsym:
        t = np.linspace(0.0, 1.0, ds + du)
        dr, dc = 1.0 + 0.5 * np.sin(7.0 * t) ** 2, 1.0 / (1.0 + 0.7 * t)
        F11 = dr[:ds, None] * F11 * dc[None, :ds]
        F12 = dr[:ds, None] * F12 * dc[None, ds:]
        F21 = dr[ds:, None] * F21 * dc[None, :ds]
        F22 = dr[ds:, None] * F22 * dc[None, ds:]
    tiles2 = [leaf] * (du // leaf) + ([du % leaf] if du % leaf else [])
    f = np.asfortranarray
    return dict(F11=f(F11), F12=f(F12), F21=f(F21), F22=f(F22), tiles1=tiles1, tiles2=tiles2, boxes=boxes, n=n, ny=ny)


def strong_admissibility(boxes):
    """tiles whose patches touch (share an edge or a corner) are not admissible -- the pattern the reference derives
    from the separator's graph with --blr_admissibility strong (sparse/fronts/FrontBLR.cpp:647-649)"""
    nt = len(boxes)
    adm = np.ones((nt, nt), dtype=bool)
    for i, (a0, a1, b0, b1) in enumerate(boxes):
        for j, (c0, c1, d0, d1) in enumerate(boxes):
            if a0 <= c1 and c0 <= a1 and b0 <= d1 and d0 <= b1:
                adm[i, j] = False
    return adm


def dense_schur(fr):
    """F22 - F21 F11^{-1} F12 and F11 by dense algebra (the exact answers the compressed factorization approximates)"""
    X = np.linalg.solve(fr["F11"], fr["F12"]) if fr["F12"].shape[1] else fr["F12"]
    return fr["F22"] - fr["F21"] @ X


def poisson_front_device(torch, nx, ny, p_left, p_right, leaf, upd="both", device=None):
    """The same front for an nx x ny separator plane, built and LEFT on the GPU (torch tensors; setup plumbing for fronts of the
    200^3 problem's own size, where the blocks are tens of GB): dsep = nx ny, dupd = nx ny per update plane.
    -> dict: F11 (ds x ds), F12cm, F21cm, F22 as tensors whose MEMORY is the column-major block (every block is symmetric or a
    row of symmetric blocks, so row-major storage of the transpose is the column-major block), tiles1, tiles2, boxes, norm."""
    dev, dt = device if device is not None else torch.device("cuda", torch.cuda.current_device()), torch.float64

    def sine(n):
        k = torch.arange(1, n + 1, dtype=dt, device=dev)
        return np.sqrt(2.0 / (n + 1)) * torch.sin(torch.outer(k, k) * (np.pi / (n + 1))), 2.0 * torch.cos(k * (np.pi / (n + 1)))

    Sx, cx = sine(nx)
    Sy, cy = sine(ny)
    lmb = (6.0 - cx[:, None] - cy[None, :]).reshape(-1)
    perm, tiles1, boxes = plane_order(nx, leaf, ny)
    Q = torch.kron(Sx, Sy)[torch.as_tensor(perm, device=dev)]
    del Sx, Sy

    def chain(p):
        th = torch.acosh(lmb / 2.0)
        den = 1.0 - torch.exp(-2.0 * (p + 1) * th)
        return torch.exp(-th) * (1.0 - torch.exp(-2.0 * p * th)) / den, torch.exp(-p * th) * (1.0 - torch.exp(-2.0 * th)) / den

    fun = lambda f: (Q * f[None, :]) @ Q.T
    gl11, gl1p = chain(p_left)
    gr11, gr1p = chain(p_right)
    F11 = fun(lmb - gl11 - gr11)
    sides = {"both": ("L", "R"), "left": ("L",), "right": ("R",), "none": ()}[upd]
    ds = nx * ny
    du = ds * len(sides)
    out = dict(F11=F11, tiles1=tiles1, boxes=boxes, nx=nx, ny=ny, ds=ds, du=du)
    n2 = float(torch.linalg.norm(F11)) ** 2
    if sides:
        b12 = [fun(-(gl1p if s_ == "L" else gr1p)) for s_ in sides]
        # column-major ds x du block [B_0 | B_1] == row-major (du x ds) stack of the (symmetric) blocks; F21 = F12^T column-major
        # (du x ds) == row-major (ds x du) array [B_0 | B_1]
        out["F12cm"] = torch.cat(b12, dim=0).contiguous()
        out["F21cm"] = torch.cat(b12, dim=1).contiguous()
        n2 += 2.0 * sum(float(torch.linalg.norm(b)) ** 2 for b in b12)
        del b12
        F22 = torch.zeros((du, du), dtype=dt, device=dev)
        for q, s_ in enumerate(sides):
            F22[q * ds:(q + 1) * ds, q * ds:(q + 1) * ds] = fun(-(gl11 if s_ == "L" else gr11))
        out["F22"] = F22
    del Q
    if dev.type == "cuda":
        torch.cuda.empty_cache()
    out["norm"] = float(np.sqrt(n2))
    out["tiles2"] = [leaf] * (du // leaf) + ([du % leaf] if du % leaf else [])
    return out
