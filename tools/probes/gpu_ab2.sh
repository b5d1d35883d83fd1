#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/$1; O=gpurun_out/$1
timeout 600 python bench.py --workload host --steps 2 > $O/host100k.json 2> $O/host100k.err; echo "host rc=$?"
python bench.py --workload kernel --no-cpu-baseline > $O/bench_kernel_n1.json 2> $O/bench_kernel.err; echo "kernel rc=$?"
HSSK_QR_BLOCKED=1 python bench.py --no-cpu-baseline --steps 3 > $O/bench_blockedqr.json 2> $O/bench_blockedqr.err; echo "blockedqr rc=$?"
python bench.py --no-cpu-baseline --steps 3 --leaf 512 > $O/bench_leaf512.json 2> $O/bench_leaf512.err; echo "leaf512 rc=$?"
python bench.py --no-cpu-baseline --steps 3 --nrhs 64 > $O/bench_nrhs64.json 2> $O/bench_nrhs64.err; echo "nrhs64 rc=$?"
python - $O <<'PY'
import json,sys
O=sys.argv[1]
for f in ["host100k","bench_kernel_n1","bench_blockedqr","bench_leaf512","bench_nrhs64"]:
    try:
        d=json.loads(open(O+'/'+f+'.json').read().strip().splitlines()[-1])
        print(f, round(d["ms_per_step"],2), {k: round(v*1e3,2) for k,v in d["phases_s"].items()}, d.get("roofline",{}).get("frac"))
    except Exception as e: print(f, "failed", e)
PY
