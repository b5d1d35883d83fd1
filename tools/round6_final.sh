#!/bin/bash
# Round-end GPU call (rounds 5 and 6): the GPU test tier and smoke() (what the driver runs), the default bench line with its live counter passes,
# its kernel summary and launch-by-launch timeline, the secondary lines, the BLR front lines and the 8-rank cost model.
#   usage (GPU box, repo root): bash tools/round6_final.sh <tag> [notests]
tag=${1:-r06_final}; out=/root/repo/gpurun_out/$tag; mkdir -p $out; cd /root/repo
if [ -z "$2" ]; then
  timeout 1500 python -m pytest tests -x -q -m gpu --durations=8 > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -14 $out/pytest.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $out/smoke.log
fi
timeout 600 python bench.py > $out/bench_n1.json 2> $out/bench_n1.err; echo "bench n1 rc=$?"
cd /tmp; export TMPDIR=/tmp; export STRUMPACK_AMD_BENCH_NO_PMC=1
timeout 400 rocprofv3 --kernel-trace --stats -d $out/kt -o kt --output-format csv -- python /root/repo/bench.py --no-cpu-baseline > $out/bench_under_rocprof.json 2> $out/kt.err
python /root/repo/tools/trace_tail.py $out/kt > /dev/null 2>&1
cp $out/kt/kt_kernel_stats.csv $out/kernel_stats_bench_n100k.csv 2>/dev/null
[ -f $out/kt/trace_tail.txt ] && cp $out/kt/trace_tail.txt $out/tail_timeline.txt
rm -rf $out/kt
timeout 400 rocprofv3 --kernel-trace --stats -d $out/ktk -o kt --output-format csv -- python /root/repo/bench.py --no-cpu-baseline --workload kernel --steps 4 > $out/bench_kernel_under_rocprof.json 2> $out/ktk.err
cp $out/ktk/kt_kernel_stats.csv $out/kernel_stats_bench_kernel.csv 2>/dev/null; rm -rf $out/ktk
timeout 400 rocprofv3 --kernel-trace --stats -d $out/ktb -o kt --output-format csv -- python /root/repo/bench.py --no-cpu-baseline --workload blr_front --front-n 200 --front-upd none --steps 2 --warmup 1 > $out/bench_blr_root_under_rocprof.json 2> $out/ktb.err
cp $out/ktb/kt_kernel_stats.csv $out/kernel_stats_bench_blr_front_200x200_root.csv 2>/dev/null; rm -rf $out/ktb
# launch-by-launch timeline of a kernel-matrix step; the new front end against the forms it replaced on awkward point sets
timeout 300 rocprofv3 --kernel-trace -d $out/ktl -o kt --output-format csv -- python /root/repo/bench.py --no-cpu-baseline --workload kernel --steps 2 --warmup 1 > /dev/null 2> $out/ktl.err
python /root/repo/tools/trace_tail.py $out/ktl > /dev/null 2>&1; [ -f $out/ktl/trace_tail.txt ] && cp $out/ktl/trace_tail.txt $out/kernel_timeline.txt; rm -rf $out/ktl
cd /root/repo
timeout 600 python tools/kernel_robust.py > $out/kernel_robust.txt 2>&1; grep -c "perm equal True" $out/kernel_robust.txt
timeout 300 python tools/knn_ab.py > $out/knn_ab.txt 2>&1; tail -2 $out/knn_ab.txt
timeout 300 python tools/cluster_ab.py > $out/cluster_ab.txt 2>&1; tail -2 $out/cluster_ab.txt
# secondary lines of the round (no counter passes, no CPU baseline)
for cfg in "kernel:--workload kernel --steps 4" "leaf512:--leaf 512" "nrhs64:--nrhs 64" "generated:--operand generated" "symmetric:--symmetric" "factor_ahead:--factor-ahead" \
           "blr_front_200x200_root:--workload blr_front --front-n 200 --front-upd none --steps 3 --warmup 1" "blr_front_200x100:--workload blr_front --front-n 200 --front-ny 100 --steps 2 --warmup 1"; do
  timeout 400 python bench.py --no-cpu-baseline ${cfg#*:} > $out/bench_${cfg%%:*}_n1.json 2> $out/bench_${cfg%%:*}.err; echo "${cfg%%:*} rc=$?"
done
unset STRUMPACK_AMD_BENCH_NO_PMC
timeout 400 python bench.py --workload blr_front --front-n 64 --steps 5 --warmup 2 > $out/bench_blr_front_n1.json 2> $out/bench_blr_front.err; echo "blr_front(dsep 4096, with the reference's CPU line) rc=$?"
timeout 500 python tools/scale_model.py --ranks 8 > $out/scale_model_8.json 2> $out/scale_model_8.err; echo "scale model rc=$?"
timeout 500 python tools/scale_model.py --ranks 8 --factor-ahead > $out/scale_model_8_factor_ahead.json 2> $out/scale_model_8fa.err; echo "scale model fa rc=$?"
python - $out <<'PY'
import json,sys,glob
for f in sorted(glob.glob(sys.argv[1]+"/bench_*_n1.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], "ms %.2f"%d["ms_per_step"], {k:round(v*1e3,3) for k,v in d.get("phases_s",{}).items()} or d.get("phases_ms"), (d.get("phase_roofline") or d.get("roofline") or {}).get("frac"), {k:(round(v["ms"],3), round(v["GBps"])) for k,v in (d.get("sweeps") or {}).items()})
    except Exception as e: print(f, "failed", e)
PY
python - $out/bench_n1.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d.get("roofline",{})
print("ms %.2f"%d["ms_per_step"], "value %.1f %s"%(d["value"],d["unit"]), "roofline", r.get("bound"), "%.3f"%r.get("frac",0), "traffic", r.get("traffic"), "cpu", (d.get("cpu_baseline") or {}).get("value"), d.get("phases_s"))
PY
