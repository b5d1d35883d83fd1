#!/bin/bash
# full GPU tier, default bench line, nrhs = 64 and kernel-matrix workloads, PMC passes over knn_kernel
cd /root/repo; mkdir -p gpurun_out/$1; O=gpurun_out/$1
( time timeout 1200 python -m pytest tests -m gpu -q -x --durations=8 > $O/pytest.log 2>&1 ) 2>&1 | grep real; tail -14 $O/pytest.log
timeout 400 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"
timeout 200 python bench.py --no-cpu-baseline --steps 3 --nrhs 64 > $O/bench_nrhs64.json 2> $O/bench_nrhs64.err; echo "nrhs64 rc=$?"
timeout 200 python bench.py --workload kernel --no-cpu-baseline > $O/bench_kernel_n1.json 2> $O/bench_kernel.err; echo "kernel rc=$?"
python - $O <<'PY'
import json,sys
O=sys.argv[1]
for f in ["bench_n1","bench_nrhs64","bench_kernel_n1"]:
    try:
        d=json.loads(open(O+'/'+f+'.json').read().strip().splitlines()[-1])
        print(f, round(d["ms_per_step"],2), {k: round(v*1e3,2) for k,v in d["phases_s"].items()}, d.get("roofline",{}).get("frac"), d.get("sweeps"))
    except Exception as e: print(f, "failed", e)
PY
bash tools/pmc_knn.sh 2>&1 | tail -8
