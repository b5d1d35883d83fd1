"""Per-workgroup timeline of the sketch GEMM main launch (hssk_last_dgemm_trace): durations, idle gaps per CU slot."""
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
for tb in (1, 0, 1, 0):
    hk.check(hk.lib.hssk_dgemm(hk.ctx, tb, d, n, n, 1.0, dR.ptr, d, dA.ptr, n, 0.0, dS.ptr, d))
    hk.sync()
    ms = hk.lib.hssk_last_dgemm_ms(hk.ctx)
    buf = np.zeros((8192, 4), dtype=np.int64)
    nw = hk.lib.hssk_last_dgemm_trace(hk.ctx, buf.ctypes.data, 8192)
    r = buf[:nw]
    t0 = r[:, 0].min()
    st, en = (r[:, 0] - t0) / 100.0, (r[:, 1] - t0) / 100.0   # us
    dur = en - st
    hw = r[:, 2]
    xcc = (hw >> 32) & 0xF
    cu = ((hw >> 8) & 0xF) | (((hw >> 13) & 0x7) << 4) | (((hw >> 12) & 1) << 7)   # cu_id, se_id, sh_id
    print("transB=%d  main launch %.2f ms, %d workgroups; span by trace %.2f ms" % (tb, ms, nw, en.max() / 1e3))
    print("  WG duration us: min %.0f  p10 %.0f  median %.0f  p90 %.0f  max %.0f" % (dur.min(), *np.percentile(dur, [10, 50, 90]), dur.max()))
    for x in range(8):
        m = xcc == x
        if m.any():
            print("  XCC %d: %4d WGs, CUs used %3d, mean dur %.0f us, last end %.2f ms" % (x, m.sum(), len(set(cu[m].tolist())), dur[m].mean(), en[m].max() / 1e3))
    # start-time rounds
    order = np.argsort(st)
    print("  start times (ms) of WG #0, #511, #512, #1023, #1024, last:", [round(float(st[order[i]]) / 1e3, 2) for i in (0, min(511, nw - 1), min(512, nw - 1), min(1023, nw - 1), min(1024, nw - 1), nw - 1)])
    busy = dur.sum() / 512.0 / 1e3
    print("  sum(dur)/512 slots = %.2f ms  -> slot occupancy %.1f%%" % (busy, busy / (en.max() / 1e3) * 100))
