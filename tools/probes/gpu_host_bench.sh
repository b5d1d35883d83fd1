#!/bin/bash
# host-resident operand line (the drop-in SP_d_struct_from_dense call) and the default line with the CPU baseline at full size
cd /root/repo; mkdir -p gpurun_out/$1; O=gpurun_out/$1
free -g | head -2; nproc
timeout 300 python bench.py --workload host --size 32768 --steps 2 > $O/host32k.json 2> $O/host32k.err; echo "host32k rc=$?"; tail -c 1200 $O/host32k.json; tail -3 $O/host32k.err
timeout 600 python bench.py --workload host --steps 2 > $O/host100k.json 2> $O/host100k.err; echo "host100k rc=$?"; tail -c 1500 $O/host100k.json; tail -3 $O/host100k.err
( time timeout 900 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err ) 2>&1 | grep real; echo "bench rc=$?"; python - <<'PY'
import json
d=json.loads(open('/root/repo/gpurun_out/'+__import__('sys').argv[1]+'/bench_n1.json').read().strip().splitlines()[-1]) if False else None
PY
tail -c 900 $O/bench_n1.json
