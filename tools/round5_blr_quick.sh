#!/bin/bash
# kernel summary of the BLR root front line only.  usage: bash tools/round5_blr_quick.sh <tag>
tag=${1:-r05_blrq}; out=/root/repo/gpurun_out/$tag; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; export STRUMPACK_AMD_BENCH_NO_PMC=1
timeout 500 rocprofv3 --kernel-trace --stats -d $out/kt -o kt --output-format csv -- python /root/repo/bench.py --no-cpu-baseline --workload blr_front --front-n 200 --front-upd none --steps 2 --warmup 1 > $out/bench.json 2> $out/kt.err
cp $out/kt/kt_kernel_stats.csv $out/kernel_stats.csv 2>/dev/null; rm -rf $out/kt
head -12 $out/kernel_stats.csv | cut -c1-170
python - $out/bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("ms %.2f"%d["ms_per_step"], d["phases_ms"], d["checks"])
PY
