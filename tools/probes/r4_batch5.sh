# round 4: the fused block-reflector kernel of the blocked QR (leaf 512): parity on the GPU, then leaf-512 lines with it off / 16 / 32 columns per workgroup
O=/root/repo/gpurun_out/r04j; mkdir -p $O; cd /root/repo
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "qr" > $O/pytest_qr.log 2>&1; echo "qr rc=$?"; tail -2 $O/pytest_qr.log
for m in 0 16 32; do
  HSSK_QR_WY=$m timeout 300 python bench.py --no-cpu-baseline --steps 5 --leaf 512 > $O/bench_leaf512_wy$m.json 2> $O/bench_leaf512_wy$m.err; echo "leaf512 wy$m rc=$?"
done
cd /tmp && export TMPDIR=/tmp
HSSK_QR_WY=16 timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof16 -o leaf512 --output-format csv -- python /root/repo/bench.py --no-cpu-baseline --steps 3 --warmup 1 --leaf 512 > $O/prof16.log 2>&1; echo "prof rc=$?"
cd /root/repo
python - <<'PY'
import json,glob,csv
for f in sorted(glob.glob("/root/repo/gpurun_out/r04j/bench_leaf512_wy*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], "ms %.2f"%d["ms_per_step"], {k:round(v*1e3,3) for k,v in d["phases_s"].items()} if "phases_s" in d else d.get("phases_ms"))
    except Exception as e: print(f, "failed", e)
for f in glob.glob("/root/repo/gpurun_out/r04j/prof16/**/*kernel_stats.csv", recursive=True):
    rows=list(csv.DictReader(open(f)))
    rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
    for r in rows[:14]: print("%-90s calls %6s total %9.3f ms avg %9.1f us"%(r["Name"][:90], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3))
PY
