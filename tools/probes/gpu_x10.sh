cd /root/repo; O=gpurun_out/r03_x10; mkdir -p $O
for z in 65536 16384 4096; do
  HSSK_ZERO_COPY_BYTES=$z STRUMPACK_AMD_BENCH_NO_PMC=1 timeout 200 python bench.py --no-cpu-baseline --nrhs 64 > $O/b_$z.json 2>$O/b_$z.err
  python - $O/b_$z.json $z <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("zero_copy", sys.argv[2], "ms %.2f"%d["ms_per_step"], {k: round(v*1e3,3) for k,v in d["phases_s"].items() if k in("tree","factor","solve")}, {k:round(v["ms"],3) for k,v in d["sweeps"].items()})
PY
done
timeout 600 python tools/scale_model.py > $O/scale_model.json 2> $O/scale_model.err; tail -4 $O/scale_model.err
