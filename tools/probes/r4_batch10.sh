# round 4: Q of the ULV leaf panels from compact-WY pairs (A/B against the register form), whole leaf QR on the blocked path
O=/root/repo/gpurun_out/r04q; mkdir -p $O; cd /root/repo
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "qr or random_shapes" > $O/pytest_k.log 2>&1; echo "kernels rc=$?"; tail -2 $O/pytest_k.log
timeout 900 python -m pytest tests/test_hss_gpu.py -x -q -m gpu > $O/pytest_h.log 2>&1; echo "hss rc=$?"; tail -2 $O/pytest_h.log
HSSK_QR_FORMQ_WY=0 timeout 300 python bench.py --no-cpu-baseline --steps 5 > $O/bench_formq_reg.json 2> $O/bench_formq_reg.err; echo "reg rc=$?"
timeout 300 python bench.py --no-cpu-baseline --steps 5 > $O/bench_formq_wy.json 2> $O/bench_formq_wy.err; echo "wy rc=$?"
HSSK_QR_BLOCKED_ROWS=128 timeout 300 python bench.py --no-cpu-baseline --steps 5 > $O/bench_blocked128.json 2> $O/bench_blocked128.err; echo "blocked rc=$?"
timeout 300 python bench.py --no-cpu-baseline --steps 5 --nrhs 64 > $O/bench_nrhs64.json 2> $O/bench_nrhs64.err; echo "nrhs64 rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o n100k --output-format csv -- python /root/repo/bench.py --no-cpu-baseline --steps 3 --warmup 1 > $O/prof.log 2>&1; echo "prof rc=$?"
cd /root/repo
python - <<'PY'
import json,glob,csv
for f in sorted(glob.glob("/root/repo/gpurun_out/r04q/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], "ms %.2f"%d["ms_per_step"], {k:round(v*1e3,3) for k,v in d["phases_s"].items()}, d.get("sweeps"))
    except Exception as e: print(f, "failed", e)
for f in sorted(glob.glob("/root/repo/gpurun_out/r04q/prof*/**/*kernel_stats.csv", recursive=True)):
    rows=list(csv.DictReader(open(f)))
    rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
    for r in rows[2:22]: print("%-90s calls %6s total %9.3f ms avg %9.1f us"%(r["Name"][:90], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3))
PY
