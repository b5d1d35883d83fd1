# round 4, third GPU call: the generated-operand path (kernel level, HSS level incl. N = 250000), the bench line with the
# generated operand next to the resident one, a larger BLR front for planning
O=/root/repo/gpurun_out/r04c; mkdir -p $O; cd /root/repo
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "gen or dgemm" > $O/pytest_kernels.log 2>&1; echo "kernels rc=$?"; tail -3 $O/pytest_kernels.log
timeout 900 python -m pytest tests/test_hss_gpu.py -x -q -k "generated" > $O/pytest_hss.log 2>&1; echo "hss rc=$?"; tail -5 $O/pytest_hss.log
timeout 300 python bench.py --no-cpu-baseline --operand generated > $O/bench_generated.json 2> $O/bench_generated.err; echo "bench gen rc=$?"; cut -c1-1500 $O/bench_generated.json
timeout 300 python bench.py --no-cpu-baseline > $O/bench_resident.json 2> $O/bench_resident.err; echo "bench res rc=$?"; python - <<'PY'
import json
for f in ("bench_generated","bench_resident"):
    try:
        d=json.loads(open("/root/repo/gpurun_out/r04c/%s.json"%f).read().strip().splitlines()[-1])
        print(f, "ms %.2f"%d["ms_per_step"], "frac %.3f"%d["roofline"]["frac"], d["phases_s"], d["roofline"].get("traffic"))
    except Exception as e: print(f, "failed", e)
PY
timeout 600 python bench.py --workload blr_front --front-n 96 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_blr96.json 2> $O/bench_blr96.err; echo "blr96 rc=$?"; cut -c1-2500 $O/bench_blr96.json; tail -3 $O/bench_blr96.err
