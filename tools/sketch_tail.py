"""Where the main launch of the sketch GEMM loses its last per cent: the per-workgroup trace (start / end tick of the 100 MHz
counter, hardware id, tile) of hssk_dgemm on the headline shape (192 x N x N, N = 1e5, A resident) -- workgroup durations by
K-slice, the moment every CU runs out of work against the end of the launch, and the idle share of the CU-time that is.
usage: sketch_tail.py [n] [splits, e.g. "0,6,8"]"""
import ctypes as C
import os
import sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from strumpack_amd import _loader  # noqa: E402
from strumpack_amd import hssk as K  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
d = 192
hk = K.Hssk(_loader.lib_path())
dA = hk.empty((n, n))
hk.check(hk.lib.hssk_fill_toeplitz(hk.ctx, dA.ptr, n, n, b"T"))
dR = hk.empty((d, n))
hk.check(hk.lib.hssk_randn(hk.ctx, dR.ptr, d, n, d, 0, n, 1))
dS = hk.empty((d, n))
for sp in (sys.argv[2] if len(sys.argv) > 2 else "0").split(","):
    if sp not in ("0", ""):
        os.environ["HSSK_DGEMM_SPLIT"] = sp
    else:
        os.environ.pop("HSSK_DGEMM_SPLIT", None)
    for tb in (1, 0):
        for rep in range(2):
            hk.check(hk.lib.hssk_dgemm(hk.ctx, tb, d, n, n, 1.0, dR.ptr, d, dA.ptr, n, 0.0, dS.ptr, d))
            hk.sync()
        ms = hk.lib.hssk_last_dgemm_ms(hk.ctx)
        buf = np.zeros((1 << 16, 4), dtype=np.int64)
        nw = hk.lib.hssk_last_dgemm_trace(hk.ctx, buf.ctypes.data, buf.shape[0])
        t = buf[:nw]
        t0, t1 = t[:, 0].min(), t[:, 1].max()
        dur = (t[:, 1] - t[:, 0]) / 100.0          # us
        total = (t1 - t0) / 100.0
        hw = ((t[:, 2] >> 32) << 8) | ((t[:, 2] >> 8) & 0xff)   # XCC id | SE / SH / CU bits of HW_ID: one key per CU
        cus = np.unique(hw)
        last = np.array([t[hw == c, 1].max() for c in cus])
        first = np.array([t[hw == c, 0].min() for c in cus])
        busy = np.array([dur[hw == c].sum() for c in cus])
        idle_tail = ((t1 - last) / 100.0).sum() / (len(cus) * total)
        idle_head = ((first - t0) / 100.0).sum() / (len(cus) * total)
        idle_all = 1.0 - busy.sum() / (len(cus) * total)
        per_cu = np.array([(hw == c).sum() for c in cus])
        print("split %s transB %d: main %.3f ms (events) trace span %.1f us, %d workgroups on %d hw ids; duration us min %.0f p10 %.0f med %.0f p90 %.0f max %.0f;"
              " workgroups per CU %d..%d" % (sp, tb, ms, total, nw, len(cus), dur.min(), np.percentile(dur, 10), np.median(dur), np.percentile(dur, 90), dur.max(),
                                               per_cu.min(), per_cu.max()))
        print("    idle CU-time: head %.4f tail %.4f all %.4f ; last-workgroup end before launch end us: med %.0f p90 %.0f max %.0f"
              % (idle_head, idle_tail, idle_all, np.median((t1 - last) / 100.0), np.percentile((t1 - last) / 100.0, 90), ((t1 - last) / 100.0).max()), flush=True)
