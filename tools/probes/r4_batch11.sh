# round 4: column sets on the device (A/B against the host form)
O=/root/repo/gpurun_out/r04r; mkdir -p $O; cd /root/repo
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "colsets or gather" > $O/pytest_k.log 2>&1; echo "kernels rc=$?"; tail -2 $O/pytest_k.log
timeout 900 python -m pytest tests/test_kernel_gpu.py -x -q -m gpu > $O/pytest_km.log 2>&1; echo "kernel tests rc=$?"; tail -2 $O/pytest_km.log
STRUMPACK_AMD_KERNEL_HOST_SETS=1 timeout 600 python bench.py --workload kernel --no-cpu-baseline --steps 4 > $O/bench_kernel_host.json 2> $O/bench_kernel_host.err; echo "host rc=$?"
timeout 600 python bench.py --workload kernel --no-cpu-baseline --steps 4 > $O/bench_kernel_dev.json 2> $O/bench_kernel_dev.err; echo "dev rc=$?"
python - <<'PY'
import json,glob
for f in sorted(glob.glob("/root/repo/gpurun_out/r04r/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], "ms %.2f"%d["ms_per_step"], {k:round(v*1e3,3) for k,v in d["phases_s"].items()}, d.get("hss"), d.get("checks"))
    except Exception as e: print(f, "failed", e)
PY
