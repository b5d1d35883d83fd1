O=/root/repo/gpurun_out/r04i; mkdir -p $O; cd /root/repo
timeout 200 python tools/dgemm_shard.py 8 > $O/dgemm_shard8.log 2>&1; cat $O/dgemm_shard8.log | grep split
timeout 200 python tools/dgemm_shard.py 4 "0,8,16,32" > $O/dgemm_shard4.log 2>&1; cat $O/dgemm_shard4.log | grep split
timeout 600 python tools/scale_model.py --ranks 8 > $O/scale_model_8.json 2> $O/scale8.err; echo "scale rc=$?"
timeout 600 python tools/scale_model.py --ranks 8 --factor-ahead > $O/scale_model_8_ahead.json 2> $O/scale8a.err; echo "scale ahead rc=$?"
python - <<'PY'
import json
for f in ("scale_model_8","scale_model_8_ahead"):
    try:
        d=json.load(open("/root/repo/gpurun_out/r04i/%s.json"%f))
        for g,v in d["ranks"].items(): print(f, g, {k:round(v[k],3) for k in ("step_ms","sketch_ms","tree_ms","factor_ms","solve_ms","construct_ms","predicted_step_ms")})
    except Exception as e: print(f, "failed", e)
PY
timeout 600 python -m pytest tests/test_cpp_driver.py -x -q -m gpu -k "user_defined" > $O/pytest_uk.log 2>&1; echo "uk rc=$?"; tail -2 $O/pytest_uk.log
for t in "blr64:--steps 3 --warmup 1" "blr_root:--front-n 200 --front-upd none --steps 2 --warmup 1" "blr_200x100:--front-n 200 --front-ny 100 --steps 2 --warmup 1"; do
  timeout 600 python bench.py --workload blr_front --no-cpu-baseline ${t#*:} > $O/${t%%:*}.json 2> $O/${t%%:*}.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("/root/repo/gpurun_out/r04i/blr*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], "ms %.2f"%d["ms_per_step"], d["phases_ms"])
    except Exception as e: print(f, "failed", e)
PY
