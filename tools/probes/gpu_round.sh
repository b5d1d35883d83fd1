#!/bin/bash
# One GPU call: GPU test tier, then the secondary bench lines of the round (multi-RHS sweeps, leaf 512, BLR front).
# usage (GPU box, repo root): bash tools/gpu_round.sh <tag>
tag=${1:-r}; out=/root/repo/gpurun_out/$tag; mkdir -p $out; cd /root/repo
timeout 1500 python -m pytest tests -m gpu -q -x --durations=8 > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -14 $out/pytest.log
for cfg in "n1:" "nrhs64:--nrhs 64" "leaf512:--leaf 512"; do
  name=${cfg%%:*}; args=${cfg#*:}
  timeout 300 python bench.py --no-cpu-baseline $args > $out/bench_$name.json 2> $out/bench_$name.err; echo "bench $name rc=$?"
  python - $out/bench_$name.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("  ms_per_step %.2f"%d["ms_per_step"], {k: round(v*1e3,3) for k,v in d["phases_s"].items()}, "sweeps", {k:(round(v["ms"],3), round(v["GBps"])) for k,v in d["sweeps"].items()}, "frac %.3f"%d["roofline"]["frac"], d["checks"])
except Exception as e: print("  parse failed", e)
PY
done
timeout 600 python bench.py --workload blr_front --no-cpu-baseline > $out/bench_blr_front.json 2> $out/bench_blr_front.err; echo "bench blr rc=$?"
python - $out/bench_blr_front.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("  blr ms_per_step %.2f"%d["ms_per_step"], d["phases_ms"], "roofline %.3f"%d["roofline"]["frac"], d["checks"])
except Exception as e: print("  parse failed", e)
PY
