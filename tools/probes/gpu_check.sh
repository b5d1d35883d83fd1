#!/bin/bash
# One GPU call: GPU test tier, default bench line, rocprofv3 kernel trace of a short bench run.
# usage (GPU box, repo root): bash tools/gpu_check.sh <tag> [pytest args]
tag=${1:-chk}; shift
out=/root/repo/gpurun_out/$tag
mkdir -p $out
cd /root/repo
timeout 900 python -m pytest tests -m gpu -q -x "$@" > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.log
timeout 300 python bench.py --no-cpu-baseline > $out/bench_n1.json 2> $out/bench_n1.err; echo "bench rc=$?"
python - $out/bench_n1.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("ms_per_step", d["ms_per_step"], "phases", d["phases_s"]); print("sweeps", d["sweeps"]); print("roofline frac", d["roofline"]["frac"], "checks", d["checks"])
except Exception as e: print("bench parse failed", e); print(open(sys.argv[1].replace(".json",".err")).read()[-2000:])
PY
cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $out/kt -o kt --output-format csv -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/bench_under_rocprof.json 2> $out/kt.err
head -40 $out/kt/kt_kernel_stats.csv | cut -c1-150
# keep the trace small enough to travel: last step only
python - $out <<'PY'
import csv,sys,glob
out=sys.argv[1]
fs=glob.glob(out+"/kt/**/*kernel_trace.csv",recursive=True)
if fs:
    rows=list(csv.DictReader(open(fs[0])))
    rows.sort(key=lambda r:int(r["Start_Timestamp"]))
    # last 3000 dispatches
    rows=rows[-3000:]
    t0=int(rows[0]["Start_Timestamp"])
    with open(out+"/trace_tail.txt","w") as f:
        prev_end=t0
        for r in rows:
            s=int(r["Start_Timestamp"]);e=int(r["End_Timestamp"])
            f.write("%10.1f gap %7.1f dur %8.1f grid %8s wg %5s %s\n"%((s-t0)/1e3,(s-prev_end)/1e3,(e-s)/1e3,r.get("Grid_Size_X",r.get("Grid_Size","")),r.get("Workgroup_Size_X",r.get("Workgroup_Size","")),r["Kernel_Name"][:70]))
            prev_end=e
    import os
    for f in fs: os.remove(f)
PY
