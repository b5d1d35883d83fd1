#!/bin/bash
# What the driver does at round end, in one GPU call: the GPU test tier, smoke(), the default bench line.
# usage (GPU box, repo root): bash tools/gpu_verify.sh <tag>
tag=${1:-verify}; out=/root/repo/gpurun_out/$tag; mkdir -p $out; cd /root/repo
timeout 1500 python -m pytest tests -x -q -m gpu > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $out/smoke.log
(time timeout 900 python bench.py) > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"; tail -4 $out/bench.err
python - $out/bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("ms %.2f value %.1f frac %.3f traffic %s cpu %s"%(d["ms_per_step"],d["value"],d["roofline"]["frac"],d["roofline"]["traffic"],(d.get("cpu_baseline") or {}).get("value")))
PY
