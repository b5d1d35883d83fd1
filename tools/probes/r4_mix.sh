# round 4: generated operand as a sliding window; factor ahead (tests, bench with / without, timeline); BLR with pooled allocations
O=/root/repo/gpurun_out/r04f; mkdir -p $O; cd /root/repo
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_hss_gpu.py -x -q -k "gen or factor_ahead" > $O/pytest_a.log 2>&1; echo "tests rc=$?"; tail -3 $O/pytest_a.log
timeout 300 python bench.py --no-cpu-baseline --operand generated > $O/bench_generated.json 2> $O/bench_generated.err; echo "bench gen rc=$?"
timeout 300 python bench.py --no-cpu-baseline > $O/bench_ahead.json 2> $O/bench_ahead.err; echo "bench ahead rc=$?"
timeout 300 python bench.py --no-cpu-baseline --no-factor-ahead > $O/bench_noahead.json 2> $O/bench_noahead.err; echo "bench noahead rc=$?"
cd /tmp; export TMPDIR=/tmp
STRUMPACK_AMD_BENCH_NO_PMC=1 timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python /root/repo/bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/kt.err
python /root/repo/tools/trace_tail.py $O/kt > /dev/null 2>&1
cp $O/kt/kt_kernel_stats.csv $O/kernel_stats_bench_n100k.csv 2>/dev/null; [ -f $O/kt/trace_tail.txt ] && cp $O/kt/trace_tail.txt $O/trace_tail.txt
rm -rf $O/kt
cd /root/repo
for la in 1 8 16; do
  STRUMPACK_AMD_BLR_LOOKAHEAD=$la timeout 300 python bench.py --workload blr_front --front-n 96 --steps 3 --warmup 1 --no-cpu-baseline > $O/blr96_la$la.json 2> $O/blr96_la$la.err
done
timeout 300 python bench.py --workload blr_front --steps 3 --warmup 1 --no-cpu-baseline > $O/blr64.json 2> $O/blr64.err
timeout 600 python bench.py --workload blr_front --front-n 200 --front-ny 100 --steps 2 --warmup 1 > $O/blr_200x100.json 2> $O/blr_200x100.err; echo "200x100 rc=$?"
timeout 600 python bench.py --workload blr_front --front-n 200 --front-upd none --steps 2 --warmup 1 > $O/blr_200x200_root.json 2> $O/blr_200x200_root.err; echo "root rc=$?"
python - <<'PY'
import json,glob
for f in sorted(glob.glob("/root/repo/gpurun_out/r04f/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], "ms %.2f"%d["ms_per_step"], d.get("phases_ms") or d.get("phases_s"), "roof", d["roofline"].get("bound"), "%.3f"%d["roofline"].get("frac",0), d.get("checks"))
    except Exception as e: print(f, "failed", e)
PY
