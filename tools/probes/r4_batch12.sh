# round 4: one-workgroup LU with the next tile of the trailing update loaded under the current one's products
O=/root/repo/gpurun_out/r04s; mkdir -p $O; cd /root/repo
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "lu or getrf or trsm or random_shapes" > $O/pytest_k.log 2>&1; echo "kernels rc=$?"; tail -2 $O/pytest_k.log
timeout 900 python -m pytest tests/test_blr_front_gpu.py -x -q -m gpu > $O/pytest_blr.log 2>&1; echo "blr rc=$?"; tail -2 $O/pytest_blr.log
for t in "blr64:--steps 3 --warmup 1" "blr_root:--front-n 200 --front-upd none --steps 2 --warmup 1"; do
  timeout 600 python bench.py --workload blr_front --no-cpu-baseline ${t#*:} > $O/${t%%:*}.json 2> $O/${t%%:*}.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("/root/repo/gpurun_out/r04s/blr*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], "ms %.2f"%d["ms_per_step"], d["phases_ms"], d.get("checks"))
    except Exception as e: print(f, "failed", e)
PY
