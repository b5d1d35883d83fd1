#!/bin/bash
# A/B of the ID of panels beyond one workgroup's registers: several workgroups per panel (id_group_kernel) against the
# streaming kernel (HSSK_ID_NO_GROUP=1) -- BLR front and leaf-512 bench lines; usage (GPU box): bash tools/gpu_idgroup_ab.sh <tag>
tag=${1:-idg}; out=gpurun_out/$tag; mkdir -p $out
export STRUMPACK_AMD_BENCH_NO_PMC=1
timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "test_id" > $out/pytest.log 2>&1; tail -3 $out/pytest.log
for v in group stream; do
  if [ $v = stream ]; then export HSSK_ID_NO_GROUP=1; else unset HSSK_ID_NO_GROUP; fi
  timeout 400 python bench.py --workload blr_front --no-cpu-baseline > $out/blr_$v.json 2> $out/blr_$v.err
  timeout 300 python bench.py --no-cpu-baseline --leaf 512 > $out/l512_$v.json 2> $out/l512_$v.err
  python - $out $v <<'PY'
import json,sys
o,v=sys.argv[1],sys.argv[2]
try:
    d=json.loads(open(o+"/blr_%s.json"%v).read().strip().splitlines()[-1]); print(v,"blr_front ms %.2f"%d["ms_per_step"], d.get("phases_ms",{}).get("one_stream_device_clock"), d.get("checks"))
except Exception as e: print(v,"blr failed",e)
try:
    d=json.loads(open(o+"/l512_%s.json"%v).read().strip().splitlines()[-1]); print(v,"leaf512 ms %.2f"%d["ms_per_step"], {k:round(x*1e3,2) for k,x in d["phases_s"].items()}, d["hss"]["rank"], d["checks"]["solve_resid_H"])
except Exception as e: print(v,"l512 failed",e)
PY
done
