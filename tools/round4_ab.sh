#!/bin/bash
# The A/B runs behind the round-4 numbers of DESIGN.md sections 6 and 8 (GPU box, repo root):  bash tools/round4_ab.sh <tag>
#   leaf 512: blocked QR with two batched GEMMs per panel / fused block reflector, one panel per pass / groups of four
#   kernel matrices: column sets on the host / on the device
tag=${1:-r04_ab}; O=/root/repo/gpurun_out/$tag; mkdir -p $O; cd /root/repo
export STRUMPACK_AMD_BENCH_NO_PMC=1
for v in "gemm:HSSK_QR_WY=0" "fused_g1:HSSK_QR_GROUP=1" "fused_g4:HSSK_QR_GROUP=4" "formq_wy:HSSK_QR_FORMQ_WY=1"; do
  env ${v#*:} timeout 300 python bench.py --no-cpu-baseline --steps 5 --leaf 512 > $O/bench_leaf512_${v%%:*}.json 2> $O/bench_leaf512_${v%%:*}.err
done
for v in "host:STRUMPACK_AMD_KERNEL_HOST_SETS=1" "device:STRUMPACK_AMD_KERNEL_HOST_SETS=0"; do
  env ${v#*:} timeout 600 python bench.py --workload kernel --no-cpu-baseline --steps 4 > $O/bench_kernel_${v%%:*}.json 2> $O/bench_kernel_${v%%:*}.err
done
python - $O <<'PY'
import json,sys,glob
for f in sorted(glob.glob(sys.argv[1]+"/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], "ms %.2f"%d["ms_per_step"], {k:round(v*1e3,3) for k,v in d["phases_s"].items()})
    except Exception as e: print(f, "failed", e)
PY
