#!/bin/bash
# Round-end GPU call: the GPU test tier and smoke() (what the driver runs), the default bench line with its live counter passes,
# its kernel summary and launch-by-launch timeline.   usage (GPU box, repo root): bash tools/round4_final.sh <tag>
tag=${1:-r04_final}; out=/root/repo/gpurun_out/$tag; mkdir -p $out; cd /root/repo
timeout 1500 python -m pytest tests -x -q -m gpu > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $out/smoke.log
timeout 600 python bench.py > $out/bench_n1.json 2> $out/bench_n1.err; echo "bench n1 rc=$?"
cd /tmp; export TMPDIR=/tmp; export STRUMPACK_AMD_BENCH_NO_PMC=1
timeout 400 rocprofv3 --kernel-trace --stats -d $out/kt -o kt --output-format csv -- python /root/repo/bench.py --no-cpu-baseline > $out/bench_under_rocprof.json 2> $out/kt.err
python /root/repo/tools/trace_tail.py $out/kt > /dev/null 2>&1
cp $out/kt/kt_kernel_stats.csv $out/kernel_stats_bench_n100k.csv 2>/dev/null
[ -f $out/kt/trace_tail.txt ] && cp $out/kt/trace_tail.txt $out/trace_tail.txt
rm -rf $out/kt
cd /root/repo
unset STRUMPACK_AMD_BENCH_NO_PMC
# secondary lines of the round (no counter passes, no CPU baseline)
for cfg in "kernel:--workload kernel --steps 4" "leaf512:--leaf 512" "nrhs64:--nrhs 64" "generated:--operand generated" "symmetric:--symmetric"; do
  STRUMPACK_AMD_BENCH_NO_PMC=1 timeout 300 python bench.py --no-cpu-baseline ${cfg#*:} > $out/bench_${cfg%%:*}_n1.json 2> $out/bench_${cfg%%:*}.err; echo "${cfg%%:*} rc=$?"
done
python - $out <<'PY'
import json,sys,glob
for f in sorted(glob.glob(sys.argv[1]+"/bench_*_n1.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], "ms %.2f"%d["ms_per_step"], {k:round(v*1e3,3) for k,v in d.get("phases_s",{}).items()}, (d.get("phase_roofline") or {}).get("frac"), (d.get("sweeps") or {}))
    except Exception as e: print(f, "failed", e)
PY
python - $out/bench_n1.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d.get("roofline",{})
print("ms %.2f"%d["ms_per_step"], "value %.1f %s"%(d["value"],d["unit"]), "roofline", r.get("bound"), "%.3f"%r.get("frac",0), "traffic", r.get("traffic"), "cpu", (d.get("cpu_baseline") or {}).get("value"), d.get("phases_s"))
PY
