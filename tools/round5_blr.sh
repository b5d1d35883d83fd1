#!/bin/bash
# Round 5: the BLR front lines (root 200 x 200, second level 200 x 100, the fixture-sized dsep 4096 front) with their phases, the LU
# kernel tests and the front tests.  usage (GPU box, repo root): bash tools/round5_blr.sh <tag> [skip-tests]
tag=${1:-r05_blr}; out=/root/repo/gpurun_out/$tag; mkdir -p $out; cd /root/repo
export STRUMPACK_AMD_BENCH_NO_PMC=1
if [ -z "$2" ]; then
  timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "trsm_lu or random_shapes" > $out/pytest_lu.log 2>&1; echo "pytest lu rc=$?"; tail -2 $out/pytest_lu.log
  timeout 900 python -m pytest tests/test_blr_front_gpu.py -x -q -m gpu > $out/pytest_blr.log 2>&1; echo "pytest blr rc=$?"; tail -2 $out/pytest_blr.log
fi
timeout 300 python bench.py --no-cpu-baseline --workload blr_front --front-n 200 --front-upd none --steps 3 --warmup 1 > $out/bench_blr_front_200x200_root_n1.json 2> $out/b1.err; echo "root rc=$?"
timeout 400 python bench.py --no-cpu-baseline --workload blr_front --front-n 200 --front-ny 100 --steps 2 --warmup 1 > $out/bench_blr_front_200x100_n1.json 2> $out/b2.err; echo "200x100 rc=$?"
timeout 300 python bench.py --no-cpu-baseline --workload blr_front --front-n 64 --steps 5 --warmup 2 > $out/bench_blr_front_n1.json 2> $out/b3.err; echo "n64 rc=$?"
python - $out <<'PY'
import json,sys,glob
for f in sorted(glob.glob(sys.argv[1]+"/bench_blr_front*_n1.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        r=d["roofline"]
        print(f.split("/")[-1], "ms %.2f"%d["ms_per_step"], d["phases_ms"], d["checks"], "roof:", r.get("dominant_phase"), "%.4f"%r["frac"], r["unit"])
    except Exception as e: print(f, "failed", e)
PY
