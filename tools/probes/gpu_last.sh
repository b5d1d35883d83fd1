#!/bin/bash
# short end-of-round check: the tests touched since the last full run + a bench line
cd /root/repo; mkdir -p gpurun_out/$1; O=gpurun_out/$1
timeout 75 python -m pytest tests/test_kernels_gpu.py tests/test_hss_gpu.py -m gpu -q -x -k "dgemm or expand or contract or two_threads or ctest_case or concurrent or streamed or float_and or api_sem" 2>&1 | tail -4
timeout 30 python bench.py --no-cpu-baseline --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - $O <<'PY'
import json,sys
O=sys.argv[1]
try:
    d=json.loads(open(O+'/bench.json').read().strip().splitlines()[-1])
    print(round(d["ms_per_step"],2), {k: round(v*1e3,3) for k,v in d["phases_s"].items()}, d.get("sweeps"))
except Exception as e: print("bench failed", e)
PY
