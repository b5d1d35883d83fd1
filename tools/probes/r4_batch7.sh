# round 4: panel groups of the fused block reflector (HSSK_QR_GROUP 1 / 2 / 4), 512-thread T factors, tiled transposed gather
O=/root/repo/gpurun_out/r04n; mkdir -p $O; cd /root/repo
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "qr or gather or random_shapes" > $O/pytest_k.log 2>&1; echo "kernels rc=$?"; tail -2 $O/pytest_k.log
timeout 900 python -m pytest tests/test_hss_gpu.py -x -q -m gpu -k "leaf512 or leaf256 or golden" > $O/pytest_h.log 2>&1; echo "hss rc=$?"; tail -2 $O/pytest_h.log
for g in 1 2 4; do
  HSSK_QR_GROUP=$g timeout 300 python bench.py --no-cpu-baseline --steps 5 --leaf 512 > $O/bench_leaf512_g$g.json 2> $O/bench_leaf512_g$g.err; echo "leaf512 g$g rc=$?"
done
timeout 300 python bench.py --no-cpu-baseline --steps 5 > $O/bench_default.json 2> $O/bench_default.err; echo "default rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o leaf512 --output-format csv -- python /root/repo/bench.py --no-cpu-baseline --steps 3 --warmup 1 --leaf 512 > $O/prof.log 2>&1; echo "prof rc=$?"
cd /root/repo
python - <<'PY'
import json,glob,csv
for f in sorted(glob.glob("/root/repo/gpurun_out/r04n/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], "ms %.2f"%d["ms_per_step"], {k:round(v*1e3,3) for k,v in d["phases_s"].items()})
    except Exception as e: print(f, "failed", e)
for f in sorted(glob.glob("/root/repo/gpurun_out/r04n/prof/**/*kernel_stats.csv", recursive=True)):
    rows=list(csv.DictReader(open(f)))
    rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
    for r in rows[2:16]: print("%-90s calls %6s total %9.3f ms avg %9.1f us"%(r["Name"][:90], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3))
PY
