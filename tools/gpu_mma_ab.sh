#!/bin/bash
# A/B of the many-right-hand-side sweep forms at nrhs = 64 (GPU box, repo root): bash tools/gpu_mma_ab.sh <tag>
tag=${1:-mma}; out=gpurun_out/$tag; mkdir -p $out
run() { name=$1; shift; env "$@" timeout 300 python bench.py --no-cpu-baseline --nrhs 64 --steps 2 --warmup 1 > $out/$name.json 2> $out/$name.err; python - $out/$name.json $name <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], "apply %.3f ms  solve %.3f ms  resid %.2e" % (d["sweeps"]["apply"]["ms"], d["sweeps"]["solve"]["ms"], d["checks"]["solve_resid_H"]))
except Exception as e: print(sys.argv[2], "failed", e)
PY
}
export STRUMPACK_AMD_BENCH_NO_PMC=1
run vector HSSK_SWEEP_MMA=0
run mma16 HSSK_SWEEP_MMA=16
run mma16_g1 HSSK_SWEEP_MMA=16 HSSK_SWEEP_MMA_GROUPS=1
run mma32 HSSK_SWEEP_MMA=16 HSSK_SWEEP_MMA_NC=32
run mma64 HSSK_SWEEP_MMA=16 HSSK_SWEEP_MMA_NC=64
timeout 600 python -m pytest tests/test_hss_gpu.py -x -q -k "multi_rhs" > $out/pytest.log 2>&1; tail -3 $out/pytest.log
