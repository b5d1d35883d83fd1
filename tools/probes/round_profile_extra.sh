#!/bin/bash
# the other bench lines of a round: host-resident operand, leaf 512, 64 right-hand sides
tag=${1:-rXX}; out=/root/repo/gpurun_out/$tag; mkdir -p $out; cd /root/repo
timeout 600 python bench.py --workload host --steps 2 > $out/bench_host_n1.json 2> $out/bench_host.err
timeout 300 python bench.py --no-cpu-baseline --steps 3 --leaf 512 > $out/bench_leaf512_n1.json 2> $out/bench_leaf512.err
timeout 300 python bench.py --no-cpu-baseline --steps 3 --nrhs 64 > $out/bench_nrhs64_n1.json 2> $out/bench_nrhs64.err
python - $out <<'PY'
import json,sys
O=sys.argv[1]
for f in ["bench_n1","bench_sjlt_n1","bench_kernel_n1","bench_host_n1","bench_leaf512_n1","bench_nrhs64_n1"]:
    try:
        d=json.loads(open(O+'/'+f+'.json').read().strip().splitlines()[-1])
        print(f, round(d["ms_per_step"],2), {k: round(v*1e3,2) for k,v in d["phases_s"].items()}, d.get("roofline",{}).get("frac"), d.get("roofline",{}).get("traffic"))
    except Exception as e: print(f, "failed", e)
PY
