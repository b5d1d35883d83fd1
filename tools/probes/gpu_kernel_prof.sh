#!/bin/bash
# kernel-matrix workload under rocprofv3 --stats: step time, phases, top kernels
O=/root/repo/gpurun_out/${1:-rk}; mkdir -p $O
(cd /tmp; export TMPDIR=/tmp; timeout 200 rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python /root/repo/bench.py --workload kernel --no-cpu-baseline > $O/b.json 2> $O/b.err)
python - $O <<'PY'
import json,csv,sys
O=sys.argv[1]
d=json.loads(open(O+"/b.json").read().strip().splitlines()[-1]); print(d["ms_per_step"], {k: round(v*1e3,2) for k,v in d["phases_s"].items()})
rows=list(csv.DictReader(open(O+"/kt/kt_kernel_stats.csv")))
st=d["steps"]+d["warmup"]
for r in rows[:10]:
    n=r["Name"]; n=n[n.find("::")+2:] if "::" in n else n
    print("%-60s %5d %9.1f us/step avg %8.1f" % (n[:60], int(r["Calls"]), int(r["TotalDurationNs"])/st/1e3, float(r["AverageNs"])/1e3))
PY
