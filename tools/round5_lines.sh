#!/bin/bash
# Round 5: the bench lines without counter passes (headline, kernel matrix, leaf 512, 64 right-hand sides).
# usage (GPU box, repo root): bash tools/round5_lines.sh <tag>
tag=${1:-r05_lines}; out=/root/repo/gpurun_out/$tag; mkdir -p $out; cd /root/repo
export STRUMPACK_AMD_BENCH_NO_PMC=1
for cfg in "n1:--steps 10 --warmup 3" "kernel:--workload kernel --steps 4" "leaf512:--leaf 512" "nrhs64:--nrhs 64"; do
  timeout 300 python bench.py --no-cpu-baseline ${cfg#*:} > $out/bench_${cfg%%:*}.json 2> $out/bench_${cfg%%:*}.err; echo "${cfg%%:*} rc=$?"
done
python - $out <<'PY'
import json,sys,glob
for f in sorted(glob.glob(sys.argv[1]+"/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], "ms %.2f"%d["ms_per_step"], {k:round(v*1e3,3) for k,v in d.get("phases_s",{}).items()}, (d.get("phase_roofline") or {}).get("frac"), {k:(round(v["ms"],3), round(v["GBps"])) for k,v in (d.get("sweeps") or {}).items()})
    except Exception as e: print(f, "failed", e)
PY
