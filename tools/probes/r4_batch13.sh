# round 4: pair merges of the TSQR with unconditional prefetches
O=/root/repo/gpurun_out/r04t; mkdir -p $O; cd /root/repo
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "tpqr or qr" > $O/pytest_k.log 2>&1; echo "kernels rc=$?"; tail -2 $O/pytest_k.log
timeout 600 python bench.py --workload kernel --no-cpu-baseline --steps 4 > $O/bench_kernel.json 2> $O/bench_kernel.err; echo "kernel rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/profk -o kernel --output-format csv -- python /root/repo/bench.py --workload kernel --no-cpu-baseline --steps 3 --warmup 1 > $O/profk.log 2>&1; echo "profk rc=$?"
cd /root/repo
python - <<'PY'
import json,glob,csv
for f in sorted(glob.glob("/root/repo/gpurun_out/r04t/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], "ms %.2f"%d["ms_per_step"], {k:round(v*1e3,3) for k,v in d["phases_s"].items()})
    except Exception as e: print(f, "failed", e)
for f in sorted(glob.glob("/root/repo/gpurun_out/r04t/prof*/**/*kernel_stats.csv", recursive=True)):
    rows=list(csv.DictReader(open(f)))
    rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
    for r in rows[:10]: print("%-90s calls %6s total %9.3f ms avg %9.1f us"%(r["Name"][:90], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3))
PY
