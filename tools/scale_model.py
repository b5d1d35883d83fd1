"""Per-rank cost model of the multi-GPU path, measured on ONE MI355X.

For G = 2, 4, 8 ranks of the headline problem (N = 100000 Toeplitz, leaf 256, rel_tol 1e-4, sharded operand):
  1. record: the G ranks run as G host threads of this process on the one GPU (every rank its own engine, stream and
     operand shard); the all-gather callback (SPXAllGatherFn) exchanges the ranks' slots through host memory and keeps
     rank 0's view of every exchange -- a functionally exact G-rank run, not a timed one (the ranks share the GPU);
  2. replay: rank 0 ALONE repeats the same construct / factor / solve with a loop-back callback that fills the other
     ranks' slots from the recording -- its kernels have the GPU to themselves, so the phase times are what one GPU of a
     G-GPU node spends; the time inside the callbacks (host round trips of the replay) is measured and left out.
Output (JSON on stdout, committed as profiles/r03_scale_model.json): per G the rank's step time, its phases, the number and
size of the collectives of each phase, and the predicted step time with a per-collective latency for small RCCL all-gathers
over xGMI (the `lat_us` argument; 8 ranks, < 1 MB: 20 us assumed -- to be replaced by the driver's SCALE run).

    python tools/scale_model.py [--size 100000] [--ranks 2,4,8] [--lat-us 20]
"""
import argparse
import ctypes as C
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401  (its HIP runtime must be the first one loaded)
from strumpack_amd import _loader, capi, dist as sdist  # noqa: E402
from strumpack_amd import hssk as K  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=100000)
    ap.add_argument("--leaf", type=int, default=256)
    ap.add_argument("--ranks", default="2,4,8")
    ap.add_argument("--lat-us", type=float, default=20.0)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--lib", default=None, help="library to load (tests: the emulator build)")
    ap.add_argument("--cold", action="store_true", help="no products in front of the replayed step (the rounds 3 - 5 form of the model)")
    ap.add_argument("--factor-ahead", action="store_true", help="SPXHSSOptions::factor_ahead: the rank's own levels are factored on a second stream behind the compression")
    a = ap.parse_args()
    libpath = a.lib or _loader.lib_path()
    L = capi.load(libpath)
    L.hssk_is_device_pointer.argtypes = [C.c_void_p]
    n = a.size
    opts = capi.StructuredMatrix.options(L, rel_tol=1e-4, abs_tol=1e-8, leaf_size=a.leaf, max_rank=50000)
    hopts = capi.StructuredMatrix.hss_options(L, random_engine="philox", factor_ahead=a.factor_ahead)
    hk = K.Hssk(libpath)
    hk_lock = threading.Lock()   # (the helper context is shared by the rank threads)
    # single-GPU reference step (whole matrix on one GPU)
    out = {"n": n, "leaf": a.leaf, "assumed_collective_latency_us": a.lat_us, "factor_ahead": a.factor_ahead, "ranks": {}}

    def d2h(ptr, nbytes):
        buf = np.empty(nbytes, dtype=np.uint8)
        with hk_lock:
            hk.check(hk.lib.hssk_memcpy_d2h(hk.ctx, buf.ctypes.data, ptr, nbytes))
        return buf

    def h2d(ptr, buf):
        with hk_lock:
            hk.check(hk.lib.hssk_memcpy_h2d(hk.ctx, ptr, buf.ctypes.data, buf.nbytes))
            hk.sync()

    for G in [int(x) for x in a.ranks.split(",")]:
        shards = []
        for g in range(G):
            lo, hi = sdist.shard_range(L, n, opts, G, g)
            dAr, dAc = hk.empty((hi - lo, n)), hk.empty((n, hi - lo))
            hk.check(hk.lib.hssk_fill_toeplitz_block(hk.ctx, dAr.ptr, hi - lo, n, hi - lo, lo, 0, b"T"))
            hk.check(hk.lib.hssk_fill_toeplitz_block(hk.ctx, dAc.ptr, n, hi - lo, n, 0, lo, b"T"))
            shards.append((lo, hi, dAr, dAc))
        dB = hk.empty((n, 1))
        hk.check(hk.lib.hssk_randn(hk.ctx, dB.ptr, n, 1, n, 0, 1, 7))
        hk.sync()
        # ---- 1. record: G threads, exchange through host memory, rank 0 keeps every gathered buffer
        barrier = threading.Barrier(G)
        slots = {}
        record = []
        lock = threading.Lock()
        errors = []

        def make_cb(rank):
            calls = [0]

            def cb(user, dbuf, bpr):
                try:
                    bpr_ = int(bpr)
                    mine = d2h(int(dbuf) + rank * bpr_, bpr_) if L.hssk_is_device_pointer(dbuf) else \
                        np.frombuffer((C.c_uint8 * bpr_).from_address(int(dbuf) + rank * bpr_), dtype=np.uint8).copy()
                    with lock:
                        slots[(calls[0], rank)] = mine
                    barrier.wait()
                    full = np.concatenate([slots[(calls[0], r)] for r in range(G)])
                    if L.hssk_is_device_pointer(dbuf):
                        h2d(int(dbuf), full)
                    else:
                        C.memmove(int(dbuf), full.ctypes.data, full.nbytes)
                    if rank == 0:
                        record.append((bpr_, bool(L.hssk_is_device_pointer(dbuf)), full))
                    barrier.wait()
                    calls[0] += 1
                except BaseException as e:   # an exception cannot cross the C caller
                    errors.append(repr(e))
                    os._exit(3)
            return capi.ALLGATHER_CB(cb)

        phase_marks = {}

        def run_rank(rank, cb, timed=None):
            lo, hi, dAr, dAc = shards[rank]
            dX = hk.empty((n, 1))
            t0 = time.perf_counter()
            H = sdist.from_blocks_device(L, dAr.ptr, hi - lo, dAc.ptr, n, n, opts, hopts, exchange_cb=cb, world=G, rank=rank)
            t1 = time.perf_counter()
            if rank == 0:
                phase_marks["construct_end"] = len(record)
            H.factor()
            t2 = time.perf_counter()
            if rank == 0:
                phase_marks["factor_end"] = len(record)
            hk.check(hk.lib.hssk_memcpy_d2d(hk.ctx, dX.ptr, dB.ptr, 8 * n))
            hk.sync()
            t3 = time.perf_counter()
            H.solve_device(dX.ptr, 1)
            t4 = time.perf_counter()
            st = H.stats()
            x = dX.get() if rank == 0 else None
            info = dict(rank=H.rank(), levels=H.levels())
            H.destroy()
            if timed is not None:
                timed.append(dict(construct=t1 - t0, factor=t2 - t1, solve=t4 - t3, stats=st, info=info, x=x))

        cbs = [make_cb(r) for r in range(G)]
        res0 = []
        threads = [threading.Thread(target=run_rank, args=(r, cbs[r], res0 if r == 0 else None)) for r in range(G)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        marks = dict(phase_marks)
        x_ref = res0[0]["x"]
        # ---- 2. replay: rank 0 alone, the other ranks' slots from the recording; callback time measured and left out
        best = None
        dW, dRw = hk.empty((192, shards[0][1] - shards[0][0])), hk.empty((192, n))   # operands of the warm-up products
        for _ in range(a.steps):
            pos = [0]
            cbtime = [0.0]

            def replay(user, dbuf, bpr):
                t0 = time.perf_counter()
                bpr_, on_dev, full = record[pos[0]]
                assert bpr_ == int(bpr), "the replay diverged from the recording"
                if on_dev:
                    h2d(int(dbuf), full)
                else:
                    C.memmove(int(dbuf), full.ctypes.data, full.nbytes)
                pos[0] += 1
                cbtime[0] += time.perf_counter() - t0
            timed = []
            cbt = {}
            rcb = capi.ALLGATHER_CB(replay)
            phase_marks.clear()
            # (run_rank reads len(record) for the marks: keep them from the recording)
            lo, hi, dAr, dAc = shards[0]
            dX = hk.empty((n, 1))
            if not a.cold:
                # a rank of a G-GPU run goes from step to step without pause; here the device has been idle for ~100 ms of
                # host work (destroy, set-up) and the first product after that measures 7.0 - 7.7 ms instead of 6.4 (round 6):
                # two products on the shard right in front of the clock keep the model to what a busy rank sees
                for _ in range(2):
                    hk.check(hk.lib.hssk_dgemm(hk.ctx, 0, 192, hi - lo, n, 1.0, dRw.ptr, 192, dAc.ptr, n, 0.0, dW.ptr, 192))
                hk.sync()
            t0 = time.perf_counter()
            H = sdist.from_blocks_device(L, dAr.ptr, hi - lo, dAc.ptr, n, n, opts, hopts, exchange_cb=rcb, world=G, rank=0)
            t1 = time.perf_counter(); cbt["construct"] = cbtime[0]
            H.factor()
            t2 = time.perf_counter(); cbt["factor"] = cbtime[0] - cbt["construct"]
            hk.check(hk.lib.hssk_memcpy_d2d(hk.ctx, dX.ptr, dB.ptr, 8 * n))
            hk.sync()
            t3 = time.perf_counter()
            H.solve_device(dX.ptr, 1)
            t4 = time.perf_counter(); cbt["solve"] = cbtime[0] - cbt["construct"] - cbt["factor"]
            st = H.stats()
            x = dX.get()
            H.destroy()
            assert pos[0] == len(record), "the replay used %d of %d recorded exchanges" % (pos[0], len(record))
            lo0, hi0 = shards[0][0], shards[0][1]
            assert np.allclose(x[lo0:hi0], x_ref[lo0:hi0], rtol=1e-9, atol=1e-12), "replayed rank differs from the recorded run"
            cur = dict(construct_ms=(t1 - t0 - cbt["construct"]) * 1e3, factor_ms=(t2 - t1 - cbt["factor"]) * 1e3,
                       solve_ms=(t4 - t3 - cbt["solve"]) * 1e3, callback_ms={k: v * 1e3 for k, v in cbt.items()},
                       sketch_ms=st["t_sketch"] * 1e3, tree_ms=st["t_tree"] * 1e3)
            cur["step_ms"] = cur["construct_ms"] + cur["factor_ms"] + cur["solve_ms"]
            if best is None or cur["step_ms"] < best["step_ms"]:
                best = cur
        ncol = {"construct": marks["construct_end"], "factor": marks["factor_end"] - marks["construct_end"],
                "solve": len(record) - marks["factor_end"]}
        byts = {"construct": sum(r[0] for r in record[:marks["construct_end"]]),
                "factor": sum(r[0] for r in record[marks["construct_end"]:marks["factor_end"]]),
                "solve": sum(r[0] for r in record[marks["factor_end"]:])}
        best.update(collectives=ncol, bytes_per_rank=byts, hss=res0[0]["info"],
                    predicted_step_ms=best["step_ms"] + sum(ncol.values()) * a.lat_us * 1e-3)
        out["ranks"][str(G)] = best
        dW.free()
        dRw.free()
        for (_, _, dAr, dAc) in shards:
            dAr.free()
            dAc.free()
        print("G=%d: rank step %.2f ms (construct %.2f [sketch %.2f, tree %.2f], factor %.2f, solve %.2f), %d collectives, predicted %.2f ms"
              % (G, best["step_ms"], best["construct_ms"], best["sketch_ms"], best["tree_ms"], best["factor_ms"], best["solve_ms"],
                 sum(ncol.values()), best["predicted_step_ms"]), file=sys.stderr, flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
