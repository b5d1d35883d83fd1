O=/root/repo/gpurun_out/r04d; mkdir -p $O; cd /root/repo
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "gen" > $O/pytest_kernels.log 2>&1; echo "kernels rc=$?"; tail -3 $O/pytest_kernels.log
timeout 900 python -m pytest tests/test_hss_gpu.py -x -q -k "generated" > $O/pytest_hss.log 2>&1; echo "hss rc=$?"; tail -5 $O/pytest_hss.log
timeout 300 python bench.py --no-cpu-baseline --operand generated > $O/bench_generated.json 2> $O/bench_generated.err; echo "bench gen rc=$?"
python - <<'PY'
import json
for f in ("bench_generated",):
    try:
        d=json.loads(open("/root/repo/gpurun_out/r04d/%s.json"%f).read().strip().splitlines()[-1])
        print(f, "ms %.2f"%d["ms_per_step"], "frac %.3f"%d["roofline"]["frac"], d["phases_s"], d["roofline"].get("traffic"))
    except Exception as e: print(f, "failed", e)
PY
