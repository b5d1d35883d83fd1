# round 4: fused block reflector, second form (loads of a pass in flight together, V prefetch): 256 / 512 threads x 16 / 32 columns
O=/root/repo/gpurun_out/r04m; mkdir -p $O; cd /root/repo
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "qr" > $O/pytest_qr.log 2>&1; echo "qr rc=$?"; tail -2 $O/pytest_qr.log
for m in 0 16 512016 512032; do
  HSSK_QR_WY=$m timeout 300 python bench.py --no-cpu-baseline --steps 5 --leaf 512 > $O/bench_leaf512_wy$m.json 2> $O/bench_leaf512_wy$m.err; echo "leaf512 wy$m rc=$?"
done
cd /tmp && export TMPDIR=/tmp
for m in 16 512016; do
HSSK_QR_WY=$m timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof$m -o leaf512 --output-format csv -- python /root/repo/bench.py --no-cpu-baseline --steps 3 --warmup 1 --leaf 512 > $O/prof$m.log 2>&1; echo "prof rc=$?"
done
cd /root/repo
python - <<'PY'
import json,glob,csv
for f in sorted(glob.glob("/root/repo/gpurun_out/r04m/bench_leaf512_wy*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], "ms %.2f"%d["ms_per_step"], d.get("phases_ms"))
    except Exception as e: print(f, "failed", e)
for f in sorted(glob.glob("/root/repo/gpurun_out/r04m/prof*/**/*kernel_stats.csv", recursive=True)):
    print(f)
    rows=list(csv.DictReader(open(f)))
    rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
    for r in rows[2:12]: print("%-90s calls %6s total %9.3f ms avg %9.1f us"%(r["Name"][:90], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3))
PY
