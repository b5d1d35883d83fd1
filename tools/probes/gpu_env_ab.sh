cd /root/repo
mkdir -p gpurun_out/r02f
timeout 600 python -m pytest tests/test_hss_gpu.py -m gpu -q -x -k "native or ctest_case" > gpurun_out/r02f/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r02f/pytest.log
for cfg in "base" "HSSK_ZERO_COPY_BYTES=0" "HIP_FORCE_DEV_KERNARG=1" "HSSK_ZERO_COPY_BYTES=0 HIP_FORCE_DEV_KERNARG=1"; do
  if [ "$cfg" = base ]; then python bench.py --no-cpu-baseline --steps 5 > gpurun_out/r02f/b.json 2>gpurun_out/r02f/b.err; else env $cfg python bench.py --no-cpu-baseline --steps 5 > gpurun_out/r02f/b.json 2>gpurun_out/r02f/b.err; fi
  python - "$cfg" <<'PY'
import json,sys
d=json.loads(open('/root/repo/gpurun_out/r02f/b.json').read().strip().splitlines()[-1])
p=d["phases_s"]; print(sys.argv[1], "ms", round(d["ms_per_step"],2), "tree", round(p["tree"]*1e3,2), "factor", round(p["factor"]*1e3,2), "solve", round(p["solve"]*1e3,3), "apply_ms", round(d["sweeps"]["apply"]["ms"],3), "solve_ms", round(d["sweeps"]["solve"]["ms"],3))
PY
done
