"""Per-step wall times of the BLR front bench step: factor, forward, backward, destroy; spin-limit status of the ID exchange
(python tools/probes/blr_step_times.py)"""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np, torch
import blr_fronts as BF
from strumpack_amd import capi, hssk as K, _loader
L = capi.load(_loader.lib_path()); hk = K.Hssk(_loader.lib_path())
def mm(A, B):
    return (torch.from_numpy(np.ascontiguousarray(A)).cuda() @ torch.from_numpy(np.ascontiguousarray(B)).cuda()).cpu().numpy()
fr = BF.poisson_front(64, 8, 8, 256, matmul=mm)
torch.cuda.empty_cache()
ds, du = fr["F11"].shape[0], fr["F12"].shape[1]
nF = float(np.sqrt(sum(np.linalg.norm(fr[k]) ** 2 for k in ("F11", "F12", "F21"))))
o = capi.StructuredMatrix.options(L, rel_tol=1e-4, abs_tol=1e-12 * nF, type=capi.SP_TYPE_BLR)
d = {k: hk.array(fr[k]) for k in ("F11", "F12", "F21", "F22")}
rng = np.random.default_rng(5)
b, bu = rng.standard_normal((ds, 1)), rng.standard_normal((du, 1))
hk.sync()
F = None
for it in range(7):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if F is not None: F.destroy()
    torch.cuda.synchronize(); t1 = time.perf_counter()
    F = capi.BLRFront.factor_device(L, ds, du, d["F11"].ptr, ds, d["F12"].ptr, ds, d["F21"].ptr, du, d["F22"].ptr, du, fr["tiles1"], fr["tiles2"], o)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    ys, yu = F.forward(b, bu)
    torch.cuda.synchronize(); t3 = time.perf_counter()
    x = F.backward(ys, np.zeros_like(bu))
    torch.cuda.synchronize(); t4 = time.perf_counter()
    print("step %d: destroy %.2f  factor %.2f  forward %.2f  backward %.2f ms   factor_wall %.2f" % (it, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, F.stats().get("t_factor", -1e-3) * 1e3), flush=True)
