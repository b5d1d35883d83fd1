#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/$1; O=gpurun_out/$1
timeout 300 python -m pytest tests/test_kernels_gpu.py tests/test_kernel_gpu.py -m gpu -q -x 2>&1 | tail -4
timeout 100 python tools/knn_only.py
timeout 200 python bench.py --workload kernel --no-cpu-baseline > $O/bench_kernel_n1.json 2> $O/bench_kernel.err; echo "kernel rc=$?"
python - $O <<'PY'
import json,sys
O=sys.argv[1]
d=json.loads(open(O+'/bench_kernel_n1.json').read().strip().splitlines()[-1])
print(round(d["ms_per_step"],2), {k: round(v*1e3,2) for k,v in d["phases_s"].items()})
PY
