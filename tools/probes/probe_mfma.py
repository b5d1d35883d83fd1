"""On-box FP64 MFMA roof probe: TFLOP/s, cycles per MFMA, effective shader clock."""
import ctypes as C
import json
import sys

sys.path.insert(0, ".")
from strumpack_amd import _loader  # noqa: E402
from strumpack_amd import hssk as K  # noqa: E402

hk = K.Hssk(_loader.lib_path())
res = []
for zero in (0, 1):
    for w in (1, 2):
        out = (C.c_double * 3)()
        for _ in range(2):
            hk.check(hk.lib.hssk_mfma_f64_probe(hk.ctx, 5000, w, zero, out))
        res.append(dict(zero_data=zero, waves_per_simd=w, tflops=out[0], cycles_per_mfma=out[1], clock_ghz=out[2]))
        print(res[-1], flush=True)
json.dump(res, open("gpurun_out/probe_mfma.json", "w"), indent=1)
