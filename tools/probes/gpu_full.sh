#!/bin/bash
# full GPU tier + default bench + host-operand bench
cd /root/repo; mkdir -p gpurun_out/$1; O=gpurun_out/$1
( time timeout 1500 python -m pytest tests -m gpu -q -x --durations=8 > $O/pytest.log 2>&1 ) 2>&1 | grep real; echo "pytest rc=$?"; tail -14 $O/pytest.log
timeout 300 python bench.py --no-cpu-baseline > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"
timeout 600 python bench.py --workload host --steps 2 > $O/host100k.json 2> $O/host100k.err; echo "host rc=$?"
python - $O <<'PY'
import json,sys
O=sys.argv[1]
d=json.loads(open(O+'/bench_n1.json').read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], {k: round(v*1e3,3) for k,v in d["phases_s"].items()}); print("sweeps", d["sweeps"]["apply"]["ms"], d["sweeps"]["apply"]["GBps"], d["sweeps"]["solve"]["ms"], d["sweeps"]["solve"]["GBps"])
h=json.loads(open(O+'/host100k.json').read().strip().splitlines()[-1])
print("host ms", h["ms_per_step"], h["phases_s"], h["roofline"]["achieved"], h["roofline"]["frac"])
PY
