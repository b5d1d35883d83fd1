#!/usr/bin/env python
"""bench.py -- the reference's headline workload on MI355X.

One "step" = one full pass of the HSS hot path on BASELINE.json's headline configuration
(configs[2]): randomized HSS compression + ULV factorization + ULV solve of the 100000 x 100000
double-precision Toeplitz matrix A(i,j) = 1/(1+|i-j|) (test/test_HSS_seq.cpp:75-78), leaf 256,
rel_tol 1e-4, d0+dd = 128+64 samples, matrix A already resident in HBM when the clock starts.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: under a launcher -- python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ... -- the ranks are the
    launcher's; without one this process starts the N ranks itself.  WORLD_SIZE and --gpus must agree.)

Prints ONE JSON line (rank 0).  `value` = algorithmic GFLOP/s of compress+factor+solve (flop model of
SURVEY.md section 8(d): 4 N^2 d for the sketch + the per-node terms) over the max-over-ranks step
time.  `roofline` describes the dominant kernel (the sketch DGEMM, FP64 MFMA bound) with the launch
duration measured by HIP events on the launch stream; `cpu_baseline` is the reference's own CPU HSS
(oracle/_ref, built from /root/reference by oracle/ref/Makefile) timed on this host on a bounded
sample (N = 32768 = BASELINE configs[1], same options) -- the only place anything under oracle/ is used here.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP64_MFMA_TFLOPS = 78.6  # gfx950 FP64 matrix peak (MI355X spec; 256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=3)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--size", dest="n", type=int, default=100000)
    p.add_argument("--leaf", type=int, default=256)
    p.add_argument("--rel-tol", type=float, default=1e-4)
    p.add_argument("--nrhs", type=int, default=1)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--factor-ahead", action=argparse.BooleanOptionalAction, default=None,
                   help="toeplitz workload: tell the library that a factorization follows -- it enqueues every tree level's ULV factorization "
                   "on a second stream as soon as the compression has settled the level (SPXHSSOptions::factor_ahead).  Default: on for "
                   "N > 1 (a rank's subtree leaves the chip room beside the compression: 18.5 -> 17.7 ms per step in the 8-rank model, "
                   "profiles/r04_scale_model.json), off at N = 1 (no gain: both phases want the whole chip, DESIGN.md section 2)")
    p.add_argument("--symmetric", action="store_true", help="toeplitz workload, SECONDARY line: declare the operand symmetric (SPXHSSOptions::symmetric_operand = 2, "
                   "checked on a sample): A^T R = A R, the second sketch GEMM is a copy; `value` counts the executed flops")
    p.add_argument("--cpu-n", type=int, default=32768)
    p.add_argument("--sketch", choices=["gaussian", "sjlt"], default="gaussian",
                   help="gaussian = the reference's default (BASELINE's metric is quoted on it); sjlt = its "
                        "--hss_compression_sketch SJLT option (nnz = 4), a separate, HBM-bound workload")
    p.add_argument("--front-n", type=int, default=64, help="blr_front: the separator is an n x n plane (dsep = n^2, dupd = 2 n^2)")
    p.add_argument("--front-ny", type=int, default=0, help="blr_front: separator = a front-n x front-ny plane (0: square); "
                   "the 200^3 problem's root front is 200 x 200 with --front-upd none, its second-level fronts 200 x 100 with both update planes")
    p.add_argument("--front-upd", choices=["both", "none"], default="both", help="blr_front: update planes of the front")
    p.add_argument("--front-device", action="store_true", help="blr_front: build the front on the GPU and leave it there (forced above 30000 rows: "
                   "the host generator and its dense checks are for the fixture-sized fronts)")
    p.add_argument("--front-leaf", type=int, default=256, help="blr_front: tile size (the reference's BLR default)")
    p.add_argument("--front-lra", choices=["rrqr", "aca"], default="rrqr",
                   help="blr_front: tile compression (--blr_low_rank_algorithm; the reference's default and BASELINE's: RRQR)")
    p.add_argument("--operand", choices=["resident", "generated"], default="resident",
                   help="toeplitz workload: 'resident' = A stored in HBM before the clock starts (BASELINE's configuration); 'generated' = A "
                        "is the library's Toeplitz formula, evaluated inside the sketch kernel and never stored (SPX_d_struct_from_generator)")
    p.add_argument("--workload", choices=["toeplitz", "kernel", "host", "blr_front"], default="toeplitz",
                   help="toeplitz = BASELINE configs[2] (headline, default); kernel = configs[3]: Gaussian-kernel matrix over "
                        "synthetic points in R^8 (kernel ridge regression fit), reported as a secondary line; host = the headline "
                        "matrix resident in HOST memory, through the reference's own entry point SP_d_struct_from_dense "
                        "(the drop-in call: the operand crosses PCIe inside the step), reported as a secondary line; blr_front = "
                        "BASELINE configs[4]'s kernel: BLR partial factorization (batched LU) of an exact frontal matrix of the 3D "
                        "7-point Poisson problem, operands resident in HBM, reported as a secondary line")
    return p.parse_args()


def kernel_workload(a, torch, dist, world, rank, local):
    """BASELINE configs[3]: N x N Gaussian-kernel matrix (examples/dense/KernelRegression.cpp: h = 1.3, lambda = 3.11),
    points uniform in [0,1)^8, HSS tree sharded by subtree over the ranks.  One step = clustering +
    neighbour search + compression from coordinates + ULV factor + solve for the regression weights."""
    import numpy as np
    from strumpack_amd import _loader, capi, dist as sdist
    L = capi.load(_loader.lib_path())
    n = a.n
    rng = np.random.default_rng(2025)
    X = rng.random((n, 8))
    y = np.sign((X - 0.5) @ rng.standard_normal(8)).reshape(-1, 1)
    opts = capi.StructuredMatrix.options(L, rel_tol=1e-2, abs_tol=1e-8, leaf_size=a.leaf, max_rank=50000)
    exch = sdist.make_exchange(L, world, rank) if world > 1 else None

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        H, Xp, perm = sdist.from_kernel(L, X, opts, kernel="Gauss", h=1.3, lam=3.11, clustering="cobble", neighbors=64, exchange_cb=exch)
        H.factor()
        w = H.solve(y[perm - 1])
        return H, w, perm

    H = None
    for _ in range(a.warmup):
        if H is not None:
            H.destroy()
        H, w, perm = step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        if H is not None:
            H.destroy()
        H, w, perm = step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    st = H.stats()
    resid = float(np.linalg.norm(H.mult(w) - y[perm - 1]) / np.linalg.norm(y))
    out = {"metric": "hss_kernel_fit_points_per_s", "value": n / (elapsed / a.steps), "unit": "points/s", "n_gpus": world,
           "steps": a.steps, "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True,
           "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": "BASELINE configs[3]: %dx%d Gaussian-kernel matrix (h=1.3, lambda=3.11) over uniform points in R^8, "
                                  "cobble clustering, leaf=%d, rel_tol=1e-2, 64 nearest neighbours: cluster + compress + ULV factor + solve"
                                  % (n, n, a.leaf), "n": n, "leaf": a.leaf},
           "phases_s": {"compress": st["t_compress"], "neighbours": st["t_random"], "column_sets": st["t_sketch"],
                        "blocks_id": st["t_tree"], "factor": st["t_factor"], "solve": st["t_solve"]},
           "hss": {"rank": H.rank(), "levels": H.levels(), "memory_MB": H.memory() / 1e6, "neighbours": int(st["d_final"])},
           "checks": {"solve_resid_H": resid}}
    # a dominant kernel of this workload: the exact nearest-neighbour search -- since round 6 a filter on the FP32 matrix cores
    # (knn2_scan_kernel: d2 - threshold of all pairs as a product with K = d + 3, v_mfma_f32_32x32x2_f32) with exact FP64 selection
    # of what passes; HIP events on the engine's stream around its launches (hssk_watch_*).  Priced against the dense FP32 MFMA
    # peak (MI355X_MICROARCH.md: 157.3 TFLOP/s); the Gram products of the row IDs (gram_kernel, FP64 MFMA) take as long.
    if st["sketch_launches"] > 0 and st["sketch_kernel_ms"] > 0:
        dim = X.shape[1]
        fl = 2.0 * (dim + 3) * float(n) * float(n)
        ach = fl * st["sketch_launches"] / (st["sketch_kernel_ms"] * 1e-3) * 1e-12
        out["roofline"] = {"kernel": "knn2_scan_kernel (exact %d nearest neighbours of every point: all pairs' squared distances minus the query's threshold on the FP32 matrix cores, "
                                     "v_mfma_f32_32x32x2_f32, K = d + 3; candidates that pass are re-evaluated in FP64 and the k smallest keys kept)" % int(st["d_final"]),
                           "bound": "mfma", "achieved": ach, "peak": 157.3, "unit": "TFLOP/s", "frac": ach / 157.3, "traffic": None,
                           "avg_launch_ms": st["sketch_kernel_ms"] / st["sketch_launches"], "launches_per_step": int(st["sketch_launches"]),
                           "flops_per_launch": fl,
                           "note": "2 (d + 3) N^2 flops per search; the launch includes the compactions of the candidate lists (3.3 per query: exact keys + bisection), "
                                   "which take about as long as the products (matrix pipe busy 0.41 in the scan alone: gpurun_out/r06knn_pmc)"}
    # dominant PHASE of the step: the row IDs of the sampled blocks -- TSQR of the d x m panels in the register QR kernels
    # (chunk QRs, then pairwise merges of triangles), the truncated QRCP of the m x m triangles, the kernel evaluations;
    # flops counted by the engine (Householder counts of the chunks and merges + the ID's), time = the phase on the host clock
    if st["t_tree"] > 0:
        fl = st["f_ortho"] + st["f_id"]
        ach = fl / st["t_tree"] * 1e-12
        out["phase_roofline"] = {"phase": "blocks + ID (hssk_kernel_eval_vbatched, gram_kernel, pchol_id_kernel)", "bound": "mfma",
                                 "achieved": ach, "peak": 78.6, "unit": "TFLOP/s", "frac": ach / 78.6, "ms": st["t_tree"] * 1e3,
                                 "flops": {"gram": st["f_ortho"], "id": st["f_id"]},
                                 "note": "row IDs of the tall sample panels from their Gram matrices: W^T W (triangles) on the FP64 matrix cores, then a diagonally "
                                         "pivoted Cholesky factorization in `rank` steps (DESIGN.md 8); the kernel evaluations (one exp per entry) are a quarter of the phase"}
    if rank == 0:
        print(json.dumps(out))
    H.destroy()


def host_workload(a, L, hk):
    """The drop-in call on the headline matrix: A lives in (pageable) host memory and goes through
    SP_d_struct_from_dense-with-options; the engine streams it through the device in column blocks, uploads overlapped
    with the sketch GEMMs.  One step = construct + factor + solve; the bound of the step is the PCIe upload of 8 N^2 bytes."""
    import numpy as np
    from strumpack_amd import capi
    n = a.n
    try:
        import psutil
        avail = psutil.virtual_memory().available
        while 8.0 * n * n * 1.25 > avail and n > 8192:
            n = int(n * 0.8) // 1024 * 1024
    except Exception:
        pass
    A = np.empty((n, n), order="F")
    slab = 4096
    dS = hk.empty((n, slab))
    for c0 in range(0, n, slab):   # generated on the device, brought to the host slab by slab (setup, untimed)
        w = min(slab, n - c0)
        hk.check(hk.lib.hssk_fill_toeplitz_block(hk.ctx, dS.ptr, n, w, n, 0, c0, b"T"))
        A[:, c0:c0 + w] = dS.get()[:, :w]
    del dS
    opts = capi.StructuredMatrix.options(L, rel_tol=a.rel_tol, abs_tol=1e-8, leaf_size=a.leaf, max_rank=50000)
    hopts = capi.StructuredMatrix.hss_options(L, random_engine="philox")
    b = np.random.default_rng(7).standard_normal((n, a.nrhs))

    def step():
        H = capi.StructuredMatrix.from_dense(L, A, opts, hopts)
        H.factor()
        x = H.solve(b)
        return H, x

    H = None
    for _ in range(a.warmup):
        if H is not None:
            H.destroy()
        H, x = step()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        if H is not None:
            H.destroy()
        H, x = step()
    elapsed = (time.perf_counter() - t0) / a.steps
    st = H.stats()
    f_total = st["f_sketch"] + st["f_local"] + st["f_reduce"] + st["f_id"] + st["f_ortho"] + st["f_ulv"] + st["f_solve"]
    resid = float(np.linalg.norm(H.mult(x) - b) / np.linalg.norm(b))
    gbs = 8.0 * n * n / st["t_sketch"] * 1e-9
    # what the link of this box delivers to a plain pinned-buffer copy (2 GB, best of 3), next to the nominal 63 GB/s
    link = None
    try:
        import torch
        hp = torch.empty(1 << 28, dtype=torch.float64, pin_memory=True)
        dv = torch.empty(1 << 28, dtype=torch.float64, device="cuda")
        for _ in range(3):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            dv.copy_(hp, non_blocking=True)
            torch.cuda.synchronize()
            r = 8.0 * (1 << 28) / (time.perf_counter() - t1) * 1e-9
            link = r if link is None else max(link, r)
        del hp, dv
    except Exception:
        pass
    out = {"metric": "hss_compress_ulv_factor_solve_gflops", "value": f_total / elapsed * 1e-9, "unit": "GFLOP/s", "n_gpus": 1,
           "steps": a.steps, "warmup": a.warmup, "ms_per_step": elapsed * 1e3, "higher_is_better": True, "scaling": "strong",
           "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": "BASELINE configs[2] matrix (%dx%d double Toeplitz, leaf=%d, rel_tol=%g) resident in pageable HOST memory, "
                                  "through the reference's entry point SP_d_struct_from_dense (+ HSS options): streamed column blocks, "
                                  "uploads overlapped with the sketch; compress + ULV factor + solve (nrhs=%d)" % (n, n, a.leaf, a.rel_tol, a.nrhs),
                      "n": n, "leaf": a.leaf, "rel_tol": a.rel_tol, "nrhs": a.nrhs, "operand": "host"},
           "phases_s": {"compress": st["t_compress"], "sketch_incl_upload": st["t_sketch"], "tree": st["t_tree"], "factor": st["t_factor"], "solve": st["t_solve"]},
           "hss": {"rank": H.rank(), "levels": H.levels(), "memory_MB": H.memory() / 1e6},
           "checks": {"solve_resid_H": resid},
           "roofline": {"kernel": "host -> device stream of A (column blocks on a copy stream, the sketch GEMMs of a block run under the next upload)", "bound": "pcie",
                        "achieved": gbs, "peak": 63.0, "unit": "GB/s", "frac": gbs / 63.0, "traffic": None,
                        "bytes": 8.0 * n * n, "link_measured_GBps": link, "frac_of_measured_link": (gbs / link if link else None),
                        "step_over_ideal_upload": (elapsed / (8.0 * n * n / (link * 1e9)) if link else None),
                        "note": "PCIe gen5 x16 ~63 GB/s per direction (link_measured_GBps: a 2 GB pinned hipMemcpy on this box); the two sketch GEMMs of a block (2.4 ms per 1.6 GB) hide behind its upload"}}
    print(json.dumps(out))
    H.destroy()


def blr_front_workload(a, L, hk, torch):
    """BASELINE configs[4]'s device kernel: the reference factors a 3D Poisson problem with BLR-compressed fronts; the
    multifrontal driver (METIS ordering, assembly tree) is out of scope, the per-front work
    BLRMatrix::construct_and_partial_factor (BLR/BLRMatrix.cpp:740, GPU precedent BLRMatrix.GPU.cpp:71-262) is what runs
    here, on an exact front of that problem (tests/blr_fronts.py: separator = an n x n plane, update part = the two
    planes that bound the eliminated slab), operands in HBM.  One step = partial factorization + forward / backward
    solve phase of the front with one right-hand side."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import blr_fronts as BF
    from strumpack_amd import capi
    n, leaf = a.front_n, a.front_leaf
    ny = a.front_ny or n
    if a.front_device or a.front_ny or a.front_upd != "both" or n * ny * (3 if a.front_upd == "both" else 1) > 30000:
        return blr_front_device_workload(a, L, hk, torch, BF, n, ny, leaf)

    def mm(A, B):   # setup only: the closed-form blocks are products with the 2D sine basis
        return (torch.from_numpy(np.ascontiguousarray(A)).cuda() @ torch.from_numpy(np.ascontiguousarray(B)).cuda()).cpu().numpy()
    fr = BF.poisson_front(n, 8, 8, leaf, matmul=mm)
    torch.cuda.empty_cache()
    ds, du = fr["F11"].shape[0], fr["F12"].shape[1]
    nF = float(np.sqrt(sum(np.linalg.norm(fr[k]) ** 2 for k in ("F11", "F12", "F21"))))
    rtol, atol = 1e-4, 1e-12 * nF      # BLROptions defaults; abs_tol scaled by the front's norm (FrontBLR.cpp:424-429)
    o = capi.StructuredMatrix.options(L, rel_tol=rtol, abs_tol=atol, type=capi.SP_TYPE_BLR)
    if L.SPX_blr_low_rank_algorithm(1 if a.front_lra == "aca" else 0):
        raise SystemExit("SPX_blr_low_rank_algorithm failed")
    d = {k: hk.array(fr[k]) for k in ("F11", "F12", "F21", "F22")}
    rng = np.random.default_rng(5)
    b, bu = rng.standard_normal((ds, 1)), rng.standard_normal((du, 1))
    hk.sync()

    def step():
        F = capi.BLRFront.factor_device(L, ds, du, d["F11"].ptr, ds, d["F12"].ptr, ds, d["F21"].ptr, du, d["F22"].ptr, du,
                                        fr["tiles1"], fr["tiles2"], o)
        ys, yu = F.forward(b, bu)
        x = F.backward(ys, np.zeros_like(bu))
        return F, x

    F = None
    for _ in range(a.warmup):
        if F is not None:
            F.destroy()
        F, x = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sts = []
    for _ in range(a.steps):
        if F is not None:
            F.destroy()
        F, x = step()
        sts.append(F.stats())
    torch.cuda.synchronize()
    elapsed = (time.perf_counter() - t0) / a.steps
    st = sts[-1]
    med = lambda k: sorted(s_[k] for s_ in sts)[len(sts) // 2]
    # phases on the device clock: extra steps OUTSIDE the timed region with the stopwatches on (one stream then: the timed
    # steps run the diagonal tile's LU next to the compression on a second stream)
    L.SPX_d_blr_front_time_phases(1)
    ph = []
    for _ in range(3):
        F.destroy()
        F, x = step()
        ph.append(F.stats())
    L.SPX_d_blr_front_time_phases(0)
    pmed = lambda k: sorted(s_[k] for s_ in ph)[len(ph) // 2]
    S = F.schur()
    Sx = BF.dense_schur(fr)
    err = lambda p, q: float(np.linalg.norm(p - q) / np.linalg.norm(q))
    rk = F.tile_ranks()
    lr = rk[rk >= 0]
    ms_schur = pmed("ms_schur")
    ach = st["f_schur"] / (ms_schur * 1e-3) * 1e-12 if ms_schur > 0 else 0.0
    gbs = st["b_schur"] / (ms_schur * 1e-3) * 1e-9 if ms_schur > 0 else 0.0
    # which roof bounds the phase: at tile ranks r the update A -= T V^T moves 16 bytes per 2 r flops
    hbm_bound = st["b_schur"] / 8000e9 > st["f_schur"] / (PEAK_FP64_MFMA_TFLOPS * 1e12)
    out = {"metric": "blr_front_partial_factor_gflops", "value": st["f_total"] / elapsed * 1e-9, "unit": "GFLOP/s", "n_gpus": 1,
           "steps": a.steps, "warmup": a.warmup, "ms_per_step": elapsed * 1e3, "higher_is_better": True, "scaling": "strong",
           "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": "BASELINE configs[4] kernel: BLR partial factorization (RL, " + a.front_lra.upper() + " tiles, rel_tol 1e-4, tiles of %d) of an exact "
                                  "3D 7-point Poisson front: separator %dx%d plane (dsep=%d), update part dupd=%d, operands in HBM; "
                                  "+ forward / backward solve phase, 1 rhs" % (leaf, n, n, ds, du),
                      "dsep": ds, "dupd": du, "tiles": [len(fr["tiles1"]), len(fr["tiles2"])], "leaf": leaf, "rel_tol": rtol},
           "phases_ms": {"factor_wall": med("t_factor") * 1e3, "one_stream_device_clock": {"lu_diag": pmed("ms_lu"), "compress_tiles": pmed("ms_compress"),
                                                                                       "trsm": pmed("ms_trsm"), "schur_gemm": ms_schur}},
           "flops": {"schur_gemm": st["f_schur"], "total": st["f_total"]},
           "blr": {"max_rank": int(st["max_rank"]), "mean_rank": float(lr.mean()) if lr.size else 0.0,
                   "nnz": [st["nnz11"], st["nnz12"], st["nnz21"]], "dense_nnz": [ds * ds, ds * du, du * ds]},
           "checks": {"schur_err_vs_dense": err(S, Sx), "B11_solve_resid": err(fr["F11"] @ x, b)},
           "roofline": blr_roofline({"lu_diag": pmed("ms_lu"), "compress_tiles": pmed("ms_compress"), "trsm": pmed("ms_trsm"), "schur_gemm": ms_schur},
                                    st, fr["tiles1"], ms_schur, ach, gbs, hbm_bound)}
    out["roofline"]["note"] = "phases: HIP events on the engine's stream around the launches of each kind in every block step (hssk_watch_*), summed over the step"
    if not a.no_cpu_baseline:
        try:
            from oracle import ref_lib as R
            if not R.available():
                raise RuntimeError("oracle/_ref not built")
            ref = R.blr_front(fr["F11"], fr["F12"], fr["F21"], fr["F22"], fr["tiles1"], fr["tiles2"], rtol, atol)
            out["cpu_baseline"] = {"value": st["f_total"] / ref["stats"][0] * 1e-9, "unit": "GFLOP/s", "cores": os.cpu_count(), "kind": "reference",
                                   "sample": "STRUMPACK v8.0.0 BLRMatrix::construct_and_partial_factor (CPU, MKL + OpenMP tasks) on the same front: "
                                             "%.3f s, max rank %d (same flop count as the GPU line)" % (ref["stats"][0], int(ref["stats"][4]))}
        except Exception as e:
            out["cpu_baseline"] = {"error": str(e)[:200]}
    print(json.dumps(out))
    F.destroy()


def blr_roofline(phases, st, tiles1, ms_schur, ach, gbs, hbm_bound):
    """`roofline` of a BLR front line: that of the LARGEST of the four phases on the one-stream device clock (diagonal LUs, tile
    compression, triangular solves, Schur GEMMs)."""
    # the line's roofline is that of the LARGEST of the four phases on the one-stream device clock
    sizes1 = [int(t_) for t_ in tiles1]
    f_lu = sum(2.0 / 3.0 * t_ ** 3 for t_ in sizes1)
    dominant = max(phases, key=lambda k_: phases[k_])
    if dominant == "schur_gemm":
        roof = {"kernel": "gemm_vbatched_kernel (Schur updates of the trailing array, deferred over blocks of block steps; v_mfma_f64_16x16x4_f64)",
                "bound": "hbm" if hbm_bound else "mfma", "achieved": gbs if hbm_bound else ach, "peak": 8000.0 if hbm_bound else PEAK_FP64_MFMA_TFLOPS,
                "unit": "GB/s" if hbm_bound else "TFLOP/s", "frac": (gbs / 8000.0) if hbm_bound else (ach / PEAK_FP64_MFMA_TFLOPS), "traffic": None,
                "phase_ms": ms_schur, "flops_per_step": st["f_schur"], "bytes_per_step": st["b_schur"], "tflops": ach, "mfma_frac": ach / PEAK_FP64_MFMA_TFLOPS}
    elif dominant == "lu_diag":
        # LU with partial pivoting of the diagonal tiles, one after the other: a chain of elimination steps on one workgroup each
        # (getrf_quad_kernel up to 192 rows -- the tile in registers --, getrf_wg2_kernel beyond); priced against the FP64 roof of
        # the chip to say how far a serial chain sits from any throughput bound
        a_lu = f_lu / (phases["lu_diag"] * 1e-3) * 1e-12 if phases["lu_diag"] > 0 else 0.0
        roof = {"kernel": "getrf_quad_kernel / getrf_wg2_kernel + trtri_diag_kernel (LU of the %d diagonal tiles of the separator, a serial chain: "
                          "%.3f ms per tile)" % (len(sizes1), phases["lu_diag"] / max(len(sizes1), 1)),
                "bound": "mfma", "achieved": a_lu, "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": a_lu / PEAK_FP64_MFMA_TFLOPS,
                "traffic": None, "phase_ms": phases["lu_diag"], "flops_per_step": f_lu}
    else:
        # tile compression (truncated pivoted QR of every off-diagonal tile: a chain of dependent Householder steps per tile) or
        # the triangular solves with the diagonal tile: their flops against the FP64 rate say how far from any throughput bound
        # a latency chain sits
        f_rest = max(st["f_total"] - st["f_schur"] - f_lu, 0.0)
        what = ("id_group_kernel / id_reg_kernel (truncated pivoted QR of the tiles of a block row and column: latency chain of Householder steps)"
                if dominant == "compress_tiles" else "trsm_fused_kernel / laswp (triangular solves of a block row and column with the diagonal tile)")
        roof = {"kernel": what, "bound": "mfma", "achieved": f_rest / (phases[dominant] * 1e-3) * 1e-12 if phases[dominant] > 0 else 0.0,
                "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s", "traffic": None, "phase_ms": phases[dominant],
                "note": "flops = everything but the Schur GEMMs and the diagonal LUs (triangular solves and compression together: an upper bound of the phase's own)"}
        roof["frac"] = roof["achieved"] / PEAK_FP64_MFMA_TFLOPS
    roof["dominant_phase"] = dominant
    return roof


def blr_front_device_workload(a, L, hk, torch, BF, nx, ny, leaf):
    """Fronts of the 200^3 problem's own size (BASELINE configs[4]: root front 200 x 200, dsep 40000; second level 200 x 100,
    dsep 20000 + dupd 40000): built on the GPU and left there (tests/blr_fronts.py: poisson_front_device), checked there --
    B11 \\ b against F11, the Schur complement against sampled dense algebra (F22 R - F21 F11^{-1} F12 R with a dense solve).
    torch is setup / checking plumbing only; the timed step is the library's partial factorization + solve phases."""
    import numpy as np
    from strumpack_amd import capi, dist as sdist
    fr = BF.poisson_front_device(torch, nx, ny, 8, 8, leaf, upd=a.front_upd)
    ds, du = fr["ds"], fr["du"]
    rtol, atol = 1e-4, 1e-12 * fr["norm"]
    o = capi.StructuredMatrix.options(L, rel_tol=rtol, abs_tol=atol, type=capi.SP_TYPE_BLR)
    if L.SPX_blr_low_rank_algorithm(1 if a.front_lra == "aca" else 0):
        raise SystemExit("SPX_blr_low_rank_algorithm failed")
    ptr = lambda k: fr[k].data_ptr() if k in fr else None
    rng = np.random.default_rng(5)
    b, bu = rng.standard_normal((ds, 1)), (rng.standard_normal((du, 1)) if du else None)
    torch.cuda.synchronize()

    walls = {"factor_call": [], "solve_phases": [], "destroy": []}

    def step():
        t_a = time.perf_counter()
        F = capi.BLRFront.factor_device(L, ds, du, ptr("F11"), ds, ptr("F12cm"), ds, ptr("F21cm"), max(du, 1), ptr("F22"), max(du, 1),
                                        fr["tiles1"], fr["tiles2"], o)
        t_b = time.perf_counter()
        ys, yu = F.forward(b, bu)
        x = F.backward(ys, np.zeros_like(bu) if du else None)
        walls["factor_call"].append((t_b - t_a) * 1e3)
        walls["solve_phases"].append((time.perf_counter() - t_b) * 1e3)
        return F, x

    F = None
    for _ in range(a.warmup):
        if F is not None:
            F.destroy()
        F, x = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sts = []
    for _ in range(a.steps):
        if F is not None:
            t_d = time.perf_counter()
            F.destroy()
            walls["destroy"].append((time.perf_counter() - t_d) * 1e3)
        F, x = step()
        sts.append(F.stats())
    torch.cuda.synchronize()
    elapsed = (time.perf_counter() - t0) / a.steps
    st = sts[-1]
    walls_out = {k_: [round(v_, 2) for v_ in vs[-a.steps:]] for k_, vs in walls.items()}   # the timed steps, call by call (host clock)
    med = lambda k: sorted(s_[k] for s_ in sts)[len(sts) // 2]
    # phases on the device clock: ONE extra step outside the timed region with the stopwatches on
    L.SPX_d_blr_front_time_phases(1)
    F.destroy()
    F, x = step()
    ph = F.stats()
    L.SPX_d_blr_front_time_phases(0)
    # ---- checks, on the device
    dev = fr["F11"].device
    xt = torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    bt = torch.from_numpy(b).to(dev)
    resid = float(torch.linalg.norm(fr["F11"] @ xt - bt) / torch.linalg.norm(bt))     # (F11 symmetric: row-major == column-major)
    schur_err = None
    if du:
        R = torch.from_numpy(rng.standard_normal((du, 8))).to(dev)
        sp, ld = F.schur_device()
        St = sdist._tensor(sp, ld * du, True).view(du, ld)[:, :du]        # St[j, i] = S(i, j)
        SR = St.t() @ R
        F12 = fr["F12cm"].t()      # the (du x ds) row-major stack of symmetric blocks IS the column-major ds x du block
        F21 = fr["F21cm"].t()      # the (ds x du) row-major array IS the column-major du x ds block
        ref = fr["F22"] @ R - F21 @ torch.linalg.solve(fr["F11"], F12 @ R)
        schur_err = float(torch.linalg.norm(SR - ref) / torch.linalg.norm(ref))
        del St, SR, ref, F12, F21
    rk = F.tile_ranks()
    lr = rk[rk >= 0]
    phases = {"lu_diag": ph["ms_lu"], "compress_tiles": ph["ms_compress"], "trsm": ph["ms_trsm"], "schur_gemm": ph["ms_schur"]}
    ms_schur = ph["ms_schur"]
    ach = st["f_schur"] / (ms_schur * 1e-3) * 1e-12 if ms_schur > 0 else 0.0
    gbs = st["b_schur"] / (ms_schur * 1e-3) * 1e-9 if ms_schur > 0 else 0.0
    hbm_bound = st["b_schur"] / 8000e9 > st["f_schur"] / (PEAK_FP64_MFMA_TFLOPS * 1e12)
    roof = blr_roofline(phases, st, fr["tiles1"], ms_schur, ach, gbs, hbm_bound)
    out = {"metric": "blr_front_partial_factor_gflops", "value": st["f_total"] / elapsed * 1e-9, "unit": "GFLOP/s", "n_gpus": 1,
           "steps": a.steps, "warmup": a.warmup, "ms_per_step": elapsed * 1e3, "higher_is_better": True, "scaling": "strong",
           "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": "BASELINE configs[4] kernel at the 200^3 problem's own front sizes: BLR partial factorization (RL with look-ahead, "
                                  + a.front_lra.upper() + " tiles, rel_tol 1e-4, tiles of %d) of an exact 3D 7-point Poisson front: separator %dx%d plane "
                                  "(dsep=%d), update part dupd=%d, built and resident in HBM; + forward / backward solve phase, 1 rhs" % (leaf, nx, ny, ds, du),
                      "dsep": ds, "dupd": du, "tiles": [len(fr["tiles1"]), len(fr["tiles2"])], "leaf": leaf, "rel_tol": rtol,
                      "lookahead": os.environ.get("STRUMPACK_AMD_BLR_LOOKAHEAD", "sqrt(block rows left), 4..24")},
           "phases_ms": {"factor_wall": med("t_factor") * 1e3, "one_stream_device_clock": phases, "calls_of_the_timed_steps": walls_out},
           "flops": {"schur_gemm": st["f_schur"], "total": st["f_total"]},
           "blr": {"max_rank": int(st["max_rank"]), "mean_rank": float(lr.mean()) if lr.size else 0.0,
                   "nnz": [st["nnz11"], st["nnz12"], st["nnz21"]], "dense_nnz": [ds * ds, ds * du, du * ds]},
           "checks": {"schur_err_vs_sampled_dense": schur_err, "B11_solve_resid": resid},
           "roofline": roof,
           "cpu_baseline": {"skipped": "the reference's CPU routine needs minutes to hours on a front of this size (87 s on the dsep 4096 front, "
                                       "profiles/r03_bench_blr_front_n1.json); its line is the fixture-sized front's"}}
    print(json.dumps(out))
    F.destroy()


def measure_traffic(argv_extra, kernel_match, out_dir=None):
    """HBM traffic of the dominant kernel from counter passes of THIS command: rocprofv3 --kernel-trace --pmc <counter> in
    separate runs (FETCH_SIZE and WRITE_SIZE do not fit one pass, MI355X_MICROARCH.md), one step each, parsed per kernel.
    gfx950 correction of the guide: FETCH_SIZE (KB) tallies the 128-byte requests of a wide streaming read at 64 B -> x 2.
    Returns bytes per launch of the kernels whose name matches the regular expression `kernel_match` (mean over those launches), or None when
    rocprofv3 is not available / a pass fails -- never a constant from another run."""
    import csv
    import glob
    import re
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if not exe or os.environ.get("STRUMPACK_AMD_BENCH_INNER") or os.environ.get("STRUMPACK_AMD_BENCH_NO_PMC"):
        return None
    res = {}
    tmp = out_dir or tempfile.mkdtemp(prefix="bench_pmc_", dir="/tmp")
    env = dict(os.environ, STRUMPACK_AMD_BENCH_INNER="1", TMPDIR="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            cmd = [exe, "--kernel-trace", "--pmc", counter, "-d", d, "-o", "p", "--output-format", "csv", "--", sys.executable,
                   os.path.abspath(__file__), "--steps", "1", "--warmup", "0", "--no-cpu-baseline"] + argv_extra
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=240)
            fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
            if r.returncode != 0 or not fs:
                return None
            vals = [float(row["Counter_Value"]) for row in csv.DictReader(open(fs[0])) if re.search(kernel_match, row["Kernel_Name"])]
            if not vals:
                return None
            res[counter] = sum(vals) / len(vals) * 1024.0     # KB -> bytes per launch
    except Exception:
        return None
    finally:
        if not out_dir:
            shutil.rmtree(tmp, ignore_errors=True)
    return {"read_bytes": 2.0 * res["FETCH_SIZE"], "write_bytes": res["WRITE_SIZE"], "bytes": 2.0 * res["FETCH_SIZE"] + res["WRITE_SIZE"]}


def _ref_sample(R, n, leaf, rel_tol):
    r = R.bench_toeplitz(n, leaf=leaf, rel_tol=rel_tol, abs_tol=1e-8, nrhs=1)
    fl = R.flops(reset=True)
    # same flop model as the GPU number: 4 N^2 d for the sketch + the reference's own counters
    flops = 4.0 * n * n * 192 + fl["update_sample"] + fl["reduce_sample"] + fl["ID"] + fl["QR"] + \
        fl["ortho"] + fl["ULV_factor"] + fl["hss_solve"]
    t = r["compress_s"] + r["factor_s"] + r["solve_s"]
    return flops / t * 1e-9, ("STRUMPACK v8.0.0 CPU HSS (MKL, OpenMP), Toeplitz N=%d leaf=%d rel_tol=%g: compress %.3fs "
                              "factor %.3fs solve %.4fs, rank %d" % (n, leaf, rel_tol, r["compress_s"], r["factor_s"], r["solve_s"], r["rank"]))


def cpu_baseline(n_small, leaf, rel_tol, n_full=100000):
    """Reference CPU HSS (oracle/_ref) timed on this host: on the headline workload itself (N = n_full, the dense operand
    is 8 N^2 bytes of host memory: ~20 s of compression on a 256-core host) when the host has the memory, and on the
    N = n_small sample (BASELINE configs[1]) as a second point."""
    try:
        from oracle import ref_lib as R
        if not R.available():
            raise RuntimeError("oracle/_ref not built")
        out = dict(unit="GFLOP/s", cores=os.cpu_count(), kind="reference")
        big = None
        try:
            import psutil
            if psutil.virtual_memory().available > 8.0 * n_full * n_full * 1.6 and not os.environ.get("STRUMPACK_AMD_BENCH_SMALL_CPU"):
                big = _ref_sample(R, n_full, leaf, rel_tol)
        except Exception as e:   # e.g. allocation failure: keep the small sample
            out["full_size_error"] = str(e)[:200]
        small = _ref_sample(R, n_small, leaf, rel_tol)
        if big:
            out.update(value=big[0], sample=big[1], second_sample={"value": small[0], "sample": small[1]})
        else:
            out.update(value=small[0], sample=small[1] + " (host memory too small for the dense N=%d operand)" % n_full)
        return out
    except Exception as e:  # reference library unavailable on this host: time the numpy/LAPACK port instead
        import numpy as np
        from oracle import hss_oracle as O
        n = min(n, 8192)
        A = O.toeplitz(n)
        t0 = time.time()
        H = O.HSSMatrix(A, O.Options(rel_tol=rel_tol, abs_tol=1e-8, leaf_size=leaf), rgen=type(
            "G", (), {"matrix": staticmethod(lambda r, c: np.asfortranarray(np.random.default_rng(0).standard_normal((r, c))))})())
        H.factor()
        H.solve(np.ones(n))
        t = time.time() - t0
        return dict(value=4.0 * n * n * 192 / t * 1e-9, unit="GFLOP/s", cores=os.cpu_count(), kind="port",
                    sample="numpy/LAPACK oracle, Toeplitz N=%d (reference library unavailable: %s)" % (n, e))


def spawn_ranks(a):
    """`python bench.py --gpus N` without a launcher: this process starts the N ranks itself (one per GPU, RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* in their environment, rendezvous on 127.0.0.1), relays rank 0's JSON line and exits non-zero if any
    rank does -- so that both spellings of the driver's multi-GPU command measure N ranks."""
    import socket
    import subprocess
    import tempfile
    # the rendezvous port: the socket that found it stays open (SO_REUSEADDR) until the ranks have been started, so that no
    # other process of the box is handed the same number in between
    sk = socket.socket()
    sk.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    procs = []
    out0f = tempfile.TemporaryFile(mode="w+")   # (rank 0's stdout: a file, so that no pipe has to be drained while all ranks are watched)
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), LOCAL_WORLD_SIZE=str(a.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), STRUMPACK_AMD_BENCH_SPAWNED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=out0f if r == 0 else subprocess.DEVNULL, text=True))
    sk.close()
    # every rank is watched: one that dies (bad device, RCCL set-up) leaves the others in the rendezvous or in a collective until
    # torch's own time-out, minutes later -- they are given a moment, then exactly the processes started here are stopped
    first_bad = None
    while any(q.poll() is None for q in procs):
        if first_bad is None and any(q.poll() not in (None, 0) for q in procs):
            first_bad = time.time()
        if first_bad is not None and time.time() - first_bad > 5:
            for q in procs:
                if q.poll() is None:
                    q.kill()
        time.sleep(0.05)
    rcs = [q.wait() for q in procs]
    out0f.seek(0)
    out0 = out0f.read()
    out0f.close()
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    if bad:
        sys.stdout.write("".join(ln + "\n" for ln in (out0 or "").splitlines() if not ln.startswith("{")))
        print("bench: ranks failed (rank, exit code): %s" % bad, file=sys.stderr)
        raise SystemExit(next(rc for _, rc in bad if rc) if any(rc and rc > 0 for _, rc in bad) else 1)
    sys.stdout.write(out0 or "")
    sys.stdout.flush()


def main():
    a = parse()
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        return spawn_ranks(a)
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) != a.gpus:
        if int(os.environ.get("RANK", "0")) == 0:
            print("bench: --gpus %d but the launcher started WORLD_SIZE=%s ranks; the two must agree" % (a.gpus, os.environ["WORLD_SIZE"]),
                  file=sys.stderr)
        raise SystemExit(2)
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # TEST HARNESS ONLY (tests/test_dist_cpu.py): STRUMPACK_AMD_BENCH_DRYRUN_LIB names the CPU emulator build of the library;
    # the run then exercises this file's argument / rendezvous / sharding / reduction path with gloo on a tiny matrix and
    # prints its line with "dry_run": true and no value -- it measures nothing and is never what the driver runs.
    dry = os.environ.get("STRUMPACK_AMD_BENCH_DRYRUN_LIB")
    if not dry and not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    if os.environ.get("STRUMPACK_AMD_SHARE_GPU") or dry:  # functional check only: several ranks on one GPU
        local = 0
    if dry:
        os.environ["STRUMPACK_AMD_BACKEND"] = "gloo"
        os.environ["STRUMPACK_AMD_BENCH_NO_PMC"] = "1"
        os.environ.setdefault("STRUMPACK_AMD_BENCH_COMM", "auto")   # (no RCCL on the emulator: the agreed fall-back is what is exercised)
        torch.cuda.synchronize = lambda *a_, **k_: None
    else:
        torch.cuda.set_device(local)
    os.environ["STRUMPACK_AMD_DEVICE"] = str(local)
    if world > 1:
        dist.init_process_group(os.environ.get("STRUMPACK_AMD_BACKEND", "nccl"),
                                **({"device_id": torch.device("cuda", local)} if os.environ.get("STRUMPACK_AMD_BACKEND", "nccl") == "nccl" else {}))
    if a.workload == "kernel":
        kernel_workload(a, torch, dist, world, rank, local)
        if world > 1:
            dist.destroy_process_group()
        return
    from strumpack_amd import _loader, capi, dist as sdist
    from strumpack_amd import hssk as K
    libpath = dry or _loader.lib_path()
    L = capi.load(libpath)
    hk = K.Hssk(libpath, device=local)
    n = a.n
    if a.workload == "host":
        if world > 1:
            raise SystemExit("--workload host is the single-GPU drop-in call")
        host_workload(a, L, hk)
        return
    if a.workload == "blr_front":
        if world > 1:
            raise SystemExit("--workload blr_front: fronts are independent, run one per GPU (replicas only)")
        blr_front_workload(a, L, hk, torch)
        return

    opts = capi.StructuredMatrix.options(L, rel_tol=a.rel_tol, abs_tol=1e-8, leaf_size=a.leaf, max_rank=50000)
    # --factor-ahead: at N = 1 no gain (the leaf level's factorization and the first inner levels of the compression both
    # want the whole chip); on a rank's subtree of N > 1 ranks it hides most of the factorization
    if a.factor_ahead is None:
        a.factor_ahead = world > 1
    hopts = capi.StructuredMatrix.hss_options(L, random_engine="philox", sketch=a.sketch, factor_ahead=a.factor_ahead,
                                              symmetric=2 if a.symmetric else 0)
    # ---- process group.  Default for N > 1: the library's own RCCL communicator (collectives on the engine's stream) and
    # a SHARDED operand -- every rank generates only its row block and its column block of A (2 x 80 GB / N), never the
    # whole matrix.  STRUMPACK_AMD_BENCH_COMM=torch selects the round-1 path (replicated A, torch.distributed callback).
    comm_mode = "single"
    comm = exch = None
    if world > 1:
        comm_mode = os.environ.get("STRUMPACK_AMD_BENCH_COMM", "rccl")
        auto_comm = comm_mode == "auto"
        if auto_comm:
            comm_mode = "rccl"
        if comm_mode == "rccl" and a.sketch == "gaussian":
            # set the native path up and TRY it on a small matrix; every rank must succeed (agreement through
            # torch.distributed), otherwise all ranks take the torch callback path together
            ok, why = 1.0, ""
            import numpy as np

            def trial(ho):
                nt = 4096 * world
                ot = capi.StructuredMatrix.options(L, rel_tol=1e-4, abs_tol=1e-8, leaf_size=256)
                lo_t, hi_t = sdist.shard_range(L, nt, ot, world, rank)
                tr, tc = hk.empty((hi_t - lo_t, nt)), hk.empty((nt, hi_t - lo_t))
                hk.check(hk.lib.hssk_fill_toeplitz_block(hk.ctx, tr.ptr, hi_t - lo_t, nt, hi_t - lo_t, lo_t, 0, b"T"))
                hk.check(hk.lib.hssk_fill_toeplitz_block(hk.ctx, tc.ptr, nt, hi_t - lo_t, nt, 0, lo_t, b"T"))
                hk.sync()
                Ht = sdist.from_blocks_device(L, tr.ptr, hi_t - lo_t, tc.ptr, nt, nt, ot, ho, comm=comm)
                Ht.factor()
                bt = np.random.default_rng(1).standard_normal((nt, 1))
                xt = Ht.solve(bt)
                good = np.linalg.norm(Ht.mult(xt) - bt) <= 1e-10 * np.linalg.norm(bt)
                Ht.destroy()
                if not good:
                    raise RuntimeError("trial solve residual too large")

            try:
                comm = sdist.NativeComm(L)
                if L.SPX_comm_selftest(comm.h):
                    raise RuntimeError("SPX_comm_selftest failed")
                sdist.shard_range(L, n, opts, world, rank)
                trial(hopts)
            except Exception as e:   # e.g. world not a power of two, RCCL not loadable
                ok, why = 0.0, str(e)[:200]
            t = torch.tensor([ok], device="cuda" if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            if t.item() < 1.0 and a.factor_ahead and comm is not None:
                # the factorization behind the compression is an option of the run, not part of the path under test: if the trial
                # failed with it on some rank, all ranks try once more without it (and the line says so)
                a.factor_ahead = False
                hopts = capi.StructuredMatrix.hss_options(L, random_engine="philox", sketch=a.sketch, factor_ahead=False,
                                                          symmetric=2 if a.symmetric else 0)
                ok2, why2 = 1.0, why
                try:
                    trial(hopts)
                except Exception as e:
                    ok2, why2 = 0.0, str(e)[:200]
                t = torch.tensor([ok2], device="cuda" if dist.get_backend() == "nccl" else "cpu")
                dist.all_reduce(t, op=dist.ReduceOp.MIN)
                why = why2
                if t.item() >= 1.0 and rank == 0:
                    print("bench: trial run with factor_ahead failed on some rank (%s); continuing without it" % why, file=sys.stderr)
            if t.item() < 1.0:
                # No silent fallback: a line that says "rccl" must have run on RCCL with one rank per GPU.  The torch
                # callback path is taken only when asked for (STRUMPACK_AMD_BENCH_COMM=torch, or =auto: RCCL if it works).
                if not auto_comm:
                    if rank == 0:
                        print("bench: native RCCL / sharded operand unavailable on some rank (%s); refusing to fall back "
                              "(STRUMPACK_AMD_BENCH_COMM=auto or =torch selects the torch callback path)" % why, file=sys.stderr)
                    dist.destroy_process_group()
                    raise SystemExit(3)
                if rank == 0:
                    print("bench: native RCCL / sharded operand unavailable on some rank (%s); using the torch callback path" % why, file=sys.stderr)
                comm, comm_mode = None, "torch"
        else:
            comm_mode = "torch"
        if comm is None:
            exch = sdist.make_exchange(L, world, rank)
    # ---- inputs resident in HBM before the clock starts: dense A (column-major) and the rhs
    generated = a.operand == "generated"
    if generated:
        if a.sketch != "gaussian":
            raise SystemExit("--operand generated: the SJLT sketch streams a stored matrix")
    elif comm is not None:
        lo, hi = sdist.shard_range(L, n, opts, world, rank)
        dAr = hk.empty((hi - lo, n))
        dAc = hk.empty((n, hi - lo))
        hk.check(hk.lib.hssk_fill_toeplitz_block(hk.ctx, dAr.ptr, hi - lo, n, hi - lo, lo, 0, b"T"))
        hk.check(hk.lib.hssk_fill_toeplitz_block(hk.ctx, dAc.ptr, n, hi - lo, n, 0, lo, b"T"))
    else:
        dA = hk.empty((n, n))
        hk.check(hk.lib.hssk_fill_toeplitz(hk.ctx, dA.ptr, n, n, b"T"))
    dB = hk.empty((n, a.nrhs))
    dX = hk.empty((n, a.nrhs))
    hk.check(hk.lib.hssk_randn(hk.ctx, dB.ptr, n, a.nrhs, n, 0, a.nrhs, 7))
    hk.sync()

    def step():
        if generated:   # the operand is the library's Toeplitz formula: evaluated inside the sketch kernel on every rank
            H = sdist.from_generator(L, n, 1, opts, hopts, comm=comm, exchange_cb=exch)
        elif comm is not None:
            H = sdist.from_blocks_device(L, dAr.ptr, hi - lo, dAc.ptr, n, n, opts, hopts, comm=comm)
        else:
            H = sdist.from_dense_device(L, dA.ptr, n, n, opts, hopts, exch)
        H.factor()
        hk.check(hk.lib.hssk_memcpy_d2d(hk.ctx, dX.ptr, dB.ptr, 8 * n * a.nrhs))
        hk.sync()
        H.solve_device(dX.ptr, a.nrhs)
        return H

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    H = None
    for _ in range(a.warmup):
        if H is not None:
            H.destroy()
        H = step()
    barrier()
    t0 = time.perf_counter()
    stats = []
    for _ in range(a.steps):
        if H is not None:
            H.destroy()
        H = step()
        stats.append(H.stats())
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device="cuda" if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / a.steps * 1e3

    # ---- correctness gates reported with the number
    import numpy as np
    Xh = dX.get()
    Bh = dB.get()
    HX = H.mult(Xh)
    resid = float(np.linalg.norm(HX - Bh) / np.linalg.norm(Bh))
    rng = np.random.default_rng(0)
    nc = 64  # compression error sampled on 64 random columns
    cols = rng.integers(0, n, nc)
    E = np.zeros((n, nc))
    E[cols, np.arange(nc)] = 1.0
    i = np.arange(n)
    Ac = 1.0 / (1.0 + np.abs(i[:, None] - cols[None, :]))
    comp_err = float(np.linalg.norm(H.mult(E) - Ac) / np.linalg.norm(Ac))
    ax_resid = float(np.linalg.norm(Ac.T @ Xh[:, 0] - Bh[cols, 0]) / np.linalg.norm(Bh[cols, 0]))

    # ---- apply / solve sweeps on their own (HBM-bound: every D, E, B resp. ULV block is read once per call)
    reps = 10
    dY = hk.empty((n, a.nrhs))

    def timed(fn):
        ts = []
        for _ in range(reps):
            barrier()
            t1 = time.perf_counter()
            fn()
            barrier()
            ts.append((time.perf_counter() - t1) * 1e3)
        if os.environ.get("STRUMPACK_AMD_BENCH_DEBUG"):
            print(ts, file=sys.stderr)
        return sorted(ts)[len(ts) // 2]

    # the same calls on the DEVICE clock: HIP events on the engine's stream around the launches of one call (hssk_watch_*), i.e.
    # without the host's launch and synchronisation latencies that the wall-clock figure of a 0.1 ms call carries
    mctx = L.SPX_d_struct_hssk_ctx(H.h)
    hk.lib.hssk_watch_read_ms.restype = C.c_double
    hk.lib.hssk_watch_read_ms.argtypes = [C.c_void_p, C.c_int, C.c_void_p]

    def timed_device(fn):
        if dry or not mctx:
            return None
        vals = []
        for _ in range(reps):
            barrier()
            if hk.lib.hssk_watch_start(C.c_void_p(mctx), 6):
                return None
            fn()
            hk.lib.hssk_watch_stop(C.c_void_p(mctx), 6)
            ms = hk.lib.hssk_watch_read_ms(C.c_void_p(mctx), 6, None)
            if ms <= 0:
                return None
            vals.append(ms)
        return sorted(vals)[len(vals) // 2]

    apply_ms = timed(lambda: H.mult_device(dB.ptr, dY.ptr, a.nrhs))
    apply_dev_ms = timed_device(lambda: H.mult_device(dB.ptr, dY.ptr, a.nrhs))

    def one_solve():
        H.solve_device(dY.ptr, a.nrhs)
    hk.check(hk.lib.hssk_memcpy_d2d(hk.ctx, dY.ptr, dB.ptr, 8 * n * a.nrhs))
    solve_ms = timed(one_solve)
    solve_dev_ms = timed_device(one_solve)

    st = stats[-1]
    st2 = H.stats()
    f_total = st["f_sketch"] + st["f_local"] + st["f_reduce"] + st["f_id"] + st["f_ortho"] + st["f_ulv"] + st["f_solve"]
    value = f_total / (ms_per_step * 1e-3) * 1e-9
    # dominant kernel: the sketch DGEMM.  Algorithmic flops per launch = 2 d N^2 (4 N^2 d
    # for the Sr and Sc launches together), per-rank share when the sketch is sharded.
    d = int(st["d_final"])
    launches = max(st["sketch_launches"], 1)
    avg_ms = st["sketch_kernel_ms"] / launches
    # flops of the timed launches themselves (the main grid of each sketch GEMM: whole rounds of the 512
    # workgroup slots; the short tail / edge launches are separate kernels outside the event bracket)
    flops_per_launch = st["sketch_kernel_flops"] / launches
    ach = flops_per_launch / (avg_ms * 1e-3) * 1e-12 if avg_ms > 0 else 0.0
    # HBM bytes of the dominant kernel's launches, from counter passes of this very command (None without rocprofv3)
    traffic = tsrc = None
    if rank == 0 and world == 1 and a.sketch == "gaussian" and not os.environ.get("STRUMPACK_AMD_BENCH_INNER"):
        extra = ["--size", str(n), "--leaf", str(a.leaf), "--rel-tol", str(a.rel_tol), "--nrhs", str(a.nrhs), "--operand", a.operand]
        if a.factor_ahead:
            extra.append("--factor-ahead")
        if a.symmetric:
            extra.append("--symmetric")
        # the MAIN launches of both sketch products: sketch_kernel<rows / 64, transposed?, group 0>
        tmain = measure_traffic(extra, r"sketch_kernel<\d, (true|false), 0, (true|false)>")
        if tmain is not None:
            traffic = tmain["bytes"]
            tsrc = ("rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE passes of this command (one step each), main launches of "
                    "sketch_kernel<3, ., 0>: read %.1f GB (FETCH_SIZE x 2, the guide's gfx950 correction) + written %.2f GB per launch"
                    % (tmain["read_bytes"] * 1e-9, tmain["write_bytes"] * 1e-9))
    out = {
        "metric": "hss_compress_ulv_factor_solve_gflops", "value": value, "unit": "GFLOP/s", "n_gpus": world,
        "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "BASELINE configs[2]: %dx%d double Toeplitz HSS, randomized compression "
                               "(d0+dd=128+64, %s), leaf=%d, rel_tol=%g, compress + ULV factor + solve (nrhs=%d)"
                               % (n, n, "Philox samples" if a.sketch == "gaussian" else "SJLT sketch, nnz=4: NOT the configuration of BASELINE's metric",
                                  a.leaf, a.rel_tol, a.nrhs),
                   "sketch": a.sketch, "n": n, "leaf": a.leaf, "rel_tol": a.rel_tol, "nrhs": a.nrhs,
                   "factor_ahead": a.factor_ahead,
                   "symmetric_hint": ("operand declared symmetric and checked on a sample: ONE sketch product, the other a copy -- NOT the general "
                                      "two-product path of the headline line; value counts executed flops" if a.symmetric else False),
                   "operand": ("generated: A is the library's Toeplitz formula, its tiles evaluated inside the sketch kernel -- never stored "
                               "(SPX_d_struct_from_generator; bitwise the compression of the stored matrix); NOT BASELINE's configuration, which holds A in HBM"
                               if generated else "resident in HBM before the clock starts"),
                   "parallelism": "1 GPU" if world == 1 else "HSS tree partitioned by subtree over %d GPUs (sketch rows, compression, ULV, sweeps local; RCCL all-gathers of the cut-level blocks; top %d nodes replicated)" % (world, world - 1),
                   "comm": {"single": "none", "rccl": "native RCCL communicator inside the library, collectives on the engine's stream; operand " + ("generated on every rank" if generated else "sharded (row block + column block per rank)"),
                            "torch": "torch.distributed all-gather callback; operand replicated on every rank"}[comm_mode],
                   "rccl_nranks": world if comm_mode == "rccl" else 0},
        # per-phase wall times: the median over the timed steps (a single step's host-side phases are noisy)
        "phases_s": {name: sorted(s_[key] for s_ in stats)[len(stats) // 2]
                     for name, key in (("compress", "t_compress"), ("sketch", "t_sketch"), ("random", "t_random"),
                                       ("tree", "t_tree"), ("factor", "t_factor"), ("solve", "t_solve"))},
        "flops": {"sketch": st["f_sketch"], "local": st["f_local"], "reduce": st["f_reduce"], "id": st["f_id"],
                  "ortho": st["f_ortho"], "ulv": st["f_ulv"], "solve": st["f_solve"]},
        "hss": {"rank": H.rank(), "levels": H.levels(), "memory_MB": H.memory() / 1e6, "rounds": int(st["rounds"]), "d": d},
        "checks": {"solve_resid_H": resid, "compress_err_sampled": comp_err, "Ax_minus_b_sampled": ax_resid},
        # bytes = the blocks each sweep reads, once (engine-side count: D, X, B for the mat-vec; X, R~, WQ, Vt0, B, Q~ for
        # the solve) + the vectors in and out
        # ms / GBps: wall clock of the call between two host synchronisations (what a caller sees); device_ms / device_GBps: the
        # launches of the call on the device clock (what the kernels do)
        "sweeps": {"apply": {"ms": apply_ms, "bytes": st2["b_mult"] + 16.0 * n * a.nrhs,
                             "GBps": (st2["b_mult"] + 16.0 * n * a.nrhs) / (apply_ms * 1e-3) * 1e-9, "bound": "hbm (8000 GB/s); one launch, %d dependent levels" % H.levels(),
                             "device_ms": apply_dev_ms, "device_GBps": ((st2["b_mult"] + 16.0 * n * a.nrhs) / (apply_dev_ms * 1e-3) * 1e-9) if apply_dev_ms else None},
                   "solve": {"ms": solve_ms, "bytes": st2["b_solve"] + 16.0 * n * a.nrhs,
                             "GBps": (st2["b_solve"] + 16.0 * n * a.nrhs) / (solve_ms * 1e-3) * 1e-9, "bound": "hbm (8000 GB/s); two launches",
                             "device_ms": solve_dev_ms, "device_GBps": ((st2["b_solve"] + 16.0 * n * a.nrhs) / (solve_dev_ms * 1e-3) * 1e-9) if solve_dev_ms else None}},
        "roofline": {"kernel": "sketch_kernel<3> (sketch S^T = R^T op(A): 192 x 128 tiles on 8 waves, operands by LDS DMA, v_mfma_f64_16x16x4_f64)", "bound": "mfma",
                     "achieved": ach, "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_FP64_MFMA_TFLOPS,
                     "traffic": traffic,
                     "traffic_source": tsrc, "algorithmic_bytes_per_launch": ((16.0 * d * n if generated else 8.0 * st["sketch_kernel_flops"] / launches / (2.0 * d) + 8.0 * d * n) if d else None),
                     "avg_launch_ms": avg_ms, "launches_per_step": launches, "flops_per_launch": flops_per_launch},
    }
    if dry:
        out["dry_run"] = True
        out["value"] = None
    if a.sketch == "sjlt":
        # the flop model counts what is executed (2 nnz flops per element and product), so GFLOP/s is not comparable
        # with the Gaussian run: compare ms_per_step
        out["note"] = "SJLT sketch: value counts the executed flops (4 N^2 nnz for the sketch); compare ms_per_step with the Gaussian run"
    if a.sketch == "sjlt" and st["sketch_kernel_bytes"] > 0 and avg_ms > 0:
        # the SJLT products stream A once: algorithmic bytes per launch = 8 N^2 (per-rank share when sharded)
        bpl = st["sketch_kernel_bytes"] / launches
        gbs = bpl / (avg_ms * 1e-3) * 1e-9
        out["roofline"] = {"kernel": "sjlt_n_kernel / sjlt_t_kernel (S^T = (op(A) R)^T, R with 4 entries +-1 per row)",
                           "bound": "hbm", "achieved": gbs, "peak": 8000.0, "unit": "GB/s", "frac": gbs / 8000.0,
                           "traffic": None, "avg_launch_ms": avg_ms, "launches_per_step": launches, "bytes_per_launch": bpl}
        tfs = os.path.join(ROOT, "profiles", "r02_pmc_sjlt_traffic.json")
        if os.path.exists(tfs) and n == 100000 and world == 1:
            out["roofline"]["traffic"] = json.load(open(tfs)).get("hbm_read_bytes_per_launch")
    if world > 1:
        # diagnostic step with synchronised collectives: host-visible time of the compression's collectives
        os.environ["STRUMPACK_AMD_TIME_COMM"] = "1"
        H.destroy()
        H = step()
        out["comm_s"] = {"compress_collectives_synchronised": H.stats()["t_comm"], "note": "one extra step outside the timed region, every collective bracketed by stream synchronisations"}
        del os.environ["STRUMPACK_AMD_TIME_COMM"]
    if rank == 0:
        if not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(a.cpu_n, a.leaf, a.rel_tol)
        elif not a.no_cpu_baseline:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    H.destroy()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
