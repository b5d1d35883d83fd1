#!/bin/bash
# Round-end GPU call: the GPU test tier and smoke() (what the driver runs), then the bench lines whose kernels changed since the
# last refresh of profiles/ (default line with its live counter passes, leaf 512, BLR front, kernel matrices) and the rocprofv3
# kernel summaries of the default and BLR front lines.   usage (GPU box, repo root): bash tools/round3_final.sh <tag>
tag=${1:-r03_final3}; out=/root/repo/gpurun_out/$tag; mkdir -p $out; cd /root/repo
timeout 1500 python -m pytest tests -x -q -m gpu > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $out/smoke.log
timeout 600 python bench.py > $out/bench_n1.json 2> $out/bench_n1.err; echo "bench n1 rc=$?"
timeout 300 python bench.py --no-cpu-baseline --leaf 512 > $out/bench_leaf512_n1.json 2> $out/bench_leaf512.err
timeout 900 python bench.py --workload blr_front > $out/bench_blr_front_n1.json 2> $out/bench_blr_front.err; echo "bench blr rc=$?"
timeout 300 python bench.py --workload kernel --no-cpu-baseline > $out/bench_kernel_n1.json 2> $out/bench_kernel.err
cd /tmp; export TMPDIR=/tmp; export STRUMPACK_AMD_BENCH_NO_PMC=1
prof() {  # name, bench args
  timeout 400 rocprofv3 --kernel-trace --stats -d $out/kt_$1 -o kt --output-format csv -- python /root/repo/bench.py --no-cpu-baseline $2 > $out/bench_under_rocprof_$1.json 2> $out/kt_$1.err
  python /root/repo/tools/trace_tail.py $out/kt_$1 > /dev/null 2>&1
  cp $out/kt_$1/kt_kernel_stats.csv $out/kernel_stats_$1.csv 2>/dev/null
  [ -f $out/kt_$1/trace_tail.txt ] && cp $out/kt_$1/trace_tail.txt $out/trace_tail_$1.txt
}
prof bench_blr_front "--workload blr_front --steps 2 --warmup 1"
prof bench_n100k ""
cd /root/repo
for f in bench_n1 bench_leaf512_n1 bench_blr_front_n1 bench_kernel_n1; do
  python - $out/$f.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d.get("roofline",{})
    print(sys.argv[1].split("/")[-1], "ms %.2f"%d["ms_per_step"], "value %.1f %s"%(d["value"],d["unit"]), "roofline", r.get("bound"), "%.3f"%r.get("frac",0), "traffic", r.get("traffic"), "cpu", (d.get("cpu_baseline") or {}).get("value"), d.get("phases_s") or d.get("phases_ms"))
except Exception as e: print(sys.argv[1], "parse failed", e)
PY
done
