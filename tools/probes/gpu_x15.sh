cd /root/repo; O=gpurun_out/r03_x15; mkdir -p $O
STRUMPACK_AMD_BENCH_NO_PMC=1 timeout 200 python bench.py --no-cpu-baseline > $O/b.json 2>$O/b.err
python - $O/b.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("G=1 ms %.2f"%d["ms_per_step"], {k: round(v*1e3,3) for k,v in d["phases_s"].items()}, "frac %.3f"%d["roofline"]["frac"])
PY
timeout 600 python tools/scale_model.py > $O/scale_model.json 2> $O/scale_model.err; tail -4 $O/scale_model.err
