"""GPU probe of id_group_kernel: small batches first, spin-limit status after each (python tools/probes/idgroup_probe.py)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
import kernel_cases as KC
from strumpack_amd import hssk as K, _loader
hk = K.Hssk(_loader.lib_path())
for probs, seed in (([(256, 256, 1e-4, 1e-12, 129, 13)] * 1, 1), ([(256, 256, 1e-4, 1e-12, 129, 13)] * 8, 2), ([(256, 256, 1e-4, 1e-12, 129, 40)] * 64, 3),
                    ([(256, 256, 1e-4, 1e-12, 129, 13)] * 150, 4), ([(192, 391, 1e-4, 1e-10, 50000, 41)] * 8, 5)):
    t0 = time.time()
    try:
        KC.case_id(hk, probs, seed=seed)
        res = "ok"
    except AssertionError as e:
        res = "FAILED " + str(e)[:80]
    st = hk.lib.hssk_sweep_status(hk.ctx)
    print(len(probs), probs[0][:2], res, "spin-limit status", st, "%.2f s" % (time.time() - t0), flush=True)
    if st or res != "ok":
        break
