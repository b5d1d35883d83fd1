import sys, os, subprocess, json
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
code = r'''
import sys, numpy as np
sys.path.insert(0,'/root/repo')
from strumpack_amd import _loader, capi, dist as sdist
L = capi.load(_loader.lib_path())
kind = sys.argv[2]
r = np.random.default_rng(5)
n, d = 50000, 8
if kind == "mixture":
    cen = r.random((20, d)) * 4
    X = cen[r.integers(0, 20, n)] + 0.15 * r.standard_normal((n, d))
elif kind == "dups":
    X = r.random((n, d)); idx = r.integers(0, n, n // 20); X[r.permutation(n)[: n // 20]] = X[idx]
elif kind == "offset":
    X = 1e3 + r.random((n, d))
elif kind == "line":
    t = r.random(n); X = np.outer(t, np.ones(d)) + 1e-3 * r.standard_normal((n, d))
o = capi.StructuredMatrix.options(L, rel_tol=1e-2, abs_tol=1e-10, leaf_size=256, max_rank=50000)
import time
t0 = time.perf_counter()
H, Xp, perm = sdist.from_kernel(L, X, o, kernel="Gauss", h=1.0, lam=2.5, clustering="cobble", neighbors=64)
t1 = time.perf_counter()
b = np.linspace(-1, 1, n); y = H.mult(b)[:, 0]; H.factor(); x = H.solve(b)[:, 0]
res = float(np.linalg.norm(H.mult(x)[:, 0] - b) / np.linalg.norm(b))
np.savez(sys.argv[1], perm=perm, info=H.node_info(), y=y, res=res, t=t1 - t0, rank=H.rank())
'''
for kind in ("mixture", "dups", "offset", "line"):
    out = []
    for mode, env in (("new", {"HSSK_KNN_DEBUG": "1"}), ("old", {"STRUMPACK_AMD_CLUSTER_HOST": "1", "HSSK_KNN_FILTER": "0", "STRUMPACK_AMD_ID_GRAM": "0"})):
        f = "/tmp/rob_%s_%s.npz" % (mode, kind)
        r = subprocess.run([sys.executable, "-c", code, f, kind], env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
        if r.returncode: print(kind, mode, "FAILED", r.stderr[-600:]); out.append(None); continue
        msg = [l for l in r.stderr.splitlines() if "hssk_knn filtered" in l]
        out.append((np.load(f), msg[-1] if msg else ""))
    if None in out: continue
    (new, msg), (old, _) = out
    dr = np.abs(new["info"][:, 3:5] - old["info"][:, 3:5])
    print(kind, "perm equal", np.array_equal(new["perm"], old["perm"]), "rank", int(new["rank"]), int(old["rank"]), "rank diffs max", int(dr.max()), "frac", round(float((dr > 0).mean()), 3),
          "y rel diff %.2e" % (np.linalg.norm(new["y"] - old["y"]) / np.linalg.norm(old["y"])), "res %.1e %.1e" % (float(new["res"]), float(old["res"])),
          "compress s new %.3f old %.3f" % (float(new["t"]), float(old["t"])), "|", msg[18:110])
