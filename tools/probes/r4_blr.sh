# round 4: generated operand after the difference-run rewrite; BLR look-ahead depths on the fixture front and a larger one;
# the 200^3 problem's own front sizes
O=/root/repo/gpurun_out/r04e; mkdir -p $O; cd /root/repo
timeout 400 python -m pytest tests/test_kernels_gpu.py tests/test_hss_gpu.py -x -q -k "gen" > $O/pytest_gen.log 2>&1; echo "gen tests rc=$?"; tail -2 $O/pytest_gen.log
timeout 300 python bench.py --no-cpu-baseline --operand generated > $O/bench_generated.json 2> $O/bench_generated.err; echo "bench gen rc=$?"
timeout 900 python -m pytest tests/test_blr_front_gpu.py -x -q > $O/pytest_blr.log 2>&1; echo "blr tests rc=$?"; tail -3 $O/pytest_blr.log
for la in 1 4 8 16; do
  STRUMPACK_AMD_BLR_LOOKAHEAD=$la timeout 300 python bench.py --workload blr_front --steps 3 --warmup 1 --no-cpu-baseline > $O/blr64_la$la.json 2> $O/blr64_la$la.err
  STRUMPACK_AMD_BLR_LOOKAHEAD=$la timeout 300 python bench.py --workload blr_front --front-n 96 --steps 2 --warmup 1 --no-cpu-baseline > $O/blr96_la$la.json 2> $O/blr96_la$la.err
done
timeout 600 python bench.py --workload blr_front --front-n 200 --front-ny 100 --steps 1 --warmup 1 > $O/blr_200x100.json 2> $O/blr_200x100.err; echo "200x100 rc=$?"; tail -2 $O/blr_200x100.err
timeout 600 python bench.py --workload blr_front --front-n 200 --front-upd none --steps 1 --warmup 1 > $O/blr_200x200_root.json 2> $O/blr_200x200_root.err; echo "root rc=$?"; tail -2 $O/blr_200x200_root.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("/root/repo/gpurun_out/r04e/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], "ms %.2f"%d["ms_per_step"], d.get("phases_ms") or d.get("phases_s"), "roof", d["roofline"].get("bound"), "%.3f"%d["roofline"].get("frac",0), d.get("checks"))
    except Exception as e: print(f, "failed", e)
PY
