#!/bin/bash
# A/B of the many-right-hand-side sweep forms at nrhs = 64 (GPU box, repo root): bash tools/gpu_mma_ab.sh <tag>
tag=${1:-mma}; out=gpurun_out/$tag; mkdir -p $out
run() { name=$1; shift; env "$@" timeout 300 python bench.py --no-cpu-baseline --nrhs 64 --steps 2 --warmup 1 > $out/$name.json 2> $out/$name.err; python - $out/$name.json $name <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], "step %.2f ms  apply %.3f ms  solve %.3f ms  resid %.2e" % (d["ms_per_step"], d["sweeps"]["apply"]["ms"], d["sweeps"]["solve"]["ms"], d["checks"]["solve_resid_H"]))
except Exception as e: print(sys.argv[2], "failed", e)
PY
}
export STRUMPACK_AMD_BENCH_NO_PMC=1
run default A=1
run default_again A=1
run big32 HSSK_SWEEP_MMA_NC_BIG=32
run t256 HSSK_SWEEP_MMA_T_BIG=256
run t1024 HSSK_SWEEP_MMA_T_BIG=1024
run notall HSSK_GEMM_NO_TALL=1
timeout 600 python -m pytest tests/test_hss_gpu.py tests/test_kernels_gpu.py -x -q -k "multi_rhs or gemm" > $out/pytest.log 2>&1; tail -3 $out/pytest.log
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /root/repo/$out/kt -o kt --output-format csv -- python /root/repo/bench.py --no-cpu-baseline --nrhs 64 --steps 1 --warmup 0 > /root/repo/$out/bench_prof.json 2> /root/repo/$out/kt.err
python /root/repo/tools/trace_tail.py /root/repo/$out/kt > /dev/null 2>&1
