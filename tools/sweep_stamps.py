import sys, ctypes as C, numpy as np
sys.path.insert(0, "/root/repo")
from strumpack_amd import capi, hssk as K, _loader
L = capi.load(_loader.lib_path())
hk = K.Hssk(_loader.lib_path())
n = 100000
dA = hk.empty((n, n)); hk.check(hk.lib.hssk_fill_toeplitz(hk.ctx, dA.ptr, n, n, b"T"))
opts = capi.StructuredMatrix.options(L, rel_tol=1e-4, abs_tol=1e-8, leaf_size=256, max_rank=50000)
hopts = capi.StructuredMatrix.hss_options(L, random_engine="philox")
H = capi.StructuredMatrix.from_dense_device(L, dA.ptr, n, n, opts, hopts)
H.factor()
b = np.random.default_rng(1).standard_normal((n, 1))
for _ in range(3): x = H.solve(b)
st = np.zeros(3 * 1023, dtype=np.int64)
lib = C.CDLL(_loader.lib_path())
lib.hssk_debug_stamps(st.ctypes.data_as(C.c_void_p), 3 * 1023)
st = st.reshape(-1, 3).astype(np.float64) / 100.0
t0 = st[:, 0].min()
st -= t0
# workgroups by level: 512 leaves, 256, 128, ...
lo = 0; cnt = 512; lvl = 0
while cnt >= 1 and lo < 1023:
    s = st[lo:lo + cnt]
    print("level %d (%4d nodes): start %.1f..%.1f  deps arrived %.1f..%.1f  end %.1f..%.1f   (compute after deps: mean %.2f us)" %
          (lvl, cnt, s[:, 0].min(), s[:, 0].max(), s[:, 1].min(), s[:, 1].max(), s[:, 2].min(), s[:, 2].max(), (s[:, 2] - s[:, 1]).mean()))
    lo += cnt; cnt //= 2; lvl += 1
