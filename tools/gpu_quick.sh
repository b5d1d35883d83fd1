#!/bin/bash
# GPU tier (optionally a -k filter as $2) + default bench under rocprofv3 --stats
cd /root/repo; mkdir -p gpurun_out/$1; O=gpurun_out/$1
timeout 900 python -m pytest tests -m gpu -q -x ${2:+-k "$2"} 2>&1 | tail -3
(cd /tmp; export TMPDIR=/tmp; timeout 300 rocprofv3 --kernel-trace --stats -d /root/repo/$O/kt -o kt --output-format csv -- python /root/repo/bench.py --no-cpu-baseline ${BENCH_ARGS} > /root/repo/$O/bench.json 2> /root/repo/$O/kt.err)
python - $O <<'PY'
import json,sys,csv
O=sys.argv[1]
d=json.loads(open(O+'/bench.json').read().strip().splitlines()[-1])
print(round(d["ms_per_step"],2), {k: round(v*1e3,3) for k,v in d["phases_s"].items()}, d.get("sweeps",{}).get("solve",{}).get("ms"))
rows=list(csv.DictReader(open(O+'/kt/kt_kernel_stats.csv')))
steps=d["steps"]+d["warmup"]
tot=0
for r in rows:
    if 'dgemm_kernel' in r['Name'] or 'fill_toeplitz' in r['Name']: continue
    tot+=int(r['TotalDurationNs'])
print("non-sketch kernel ms/step", tot/steps/1e6, "launches/step", sum(int(r['Calls']) for r in rows)/steps)
for r in rows[:26]:
    n=r['Name']; n=n[n.find('::')+2:] if '::' in n else n
    print("%-60s %5d %9.1f us/step  avg %8.1f" % (n[:60], int(r['Calls']), int(r['TotalDurationNs'])/steps/1e3, float(r['AverageNs'])/1e3))
PY
