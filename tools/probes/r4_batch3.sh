# round 4: whole GPU tier after the day's changes; generated operand (shift-copy); symmetric hint; leaf 512; 64 right-hand sides;
# per-rank scale model; BLR lines with the adaptive look-ahead
O=/root/repo/gpurun_out/r04g; mkdir -p $O; cd /root/repo
timeout 1200 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout 300 python bench.py --no-cpu-baseline --operand generated > $O/bench_generated.json 2> $O/bench_generated.err; echo "gen rc=$?"
timeout 300 python bench.py --no-cpu-baseline --symmetric > $O/bench_symmetric.json 2> $O/bench_symmetric.err; echo "sym rc=$?"
timeout 300 python bench.py --no-cpu-baseline --leaf 512 > $O/bench_leaf512.json 2> $O/bench_leaf512.err; echo "leaf512 rc=$?"
timeout 300 python bench.py --no-cpu-baseline --nrhs 64 > $O/bench_nrhs64.json 2> $O/bench_nrhs64.err; echo "nrhs64 rc=$?"
timeout 600 python tools/scale_model.py > $O/scale_model.json 2> $O/scale_model.err; echo "scale rc=$?"; tail -c 1500 $O/scale_model.json
timeout 300 python bench.py --workload blr_front --steps 3 --warmup 1 --no-cpu-baseline > $O/blr64.json 2> $O/blr64.err
timeout 300 python bench.py --workload blr_front --front-n 96 --steps 3 --warmup 1 --no-cpu-baseline > $O/blr96.json 2> $O/blr96.err
timeout 600 python bench.py --workload blr_front --front-n 200 --front-ny 100 --steps 2 --warmup 1 > $O/blr_200x100.json 2> $O/blr_200x100.err
timeout 600 python bench.py --workload blr_front --front-n 200 --front-upd none --steps 2 --warmup 1 > $O/blr_200x200_root.json 2> $O/blr_200x200_root.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("/root/repo/gpurun_out/r04g/b*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], "ms %.2f"%d["ms_per_step"], "value %.0f"%d["value"], d.get("phases_ms") or d.get("phases_s"), "roof", d["roofline"].get("bound"), "%.3f"%d["roofline"].get("frac",0), d.get("sweeps",{}).get("apply",{}).get("ms"), d.get("sweeps",{}).get("solve",{}).get("ms"))
    except Exception as e: print(f, "failed", e)
PY
