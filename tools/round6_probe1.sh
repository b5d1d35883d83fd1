#!/bin/bash
# round 6, first call of the session: the default line, one rank's shard of the sketch GEMM, the 8-rank cost model with the
# launch timeline of rank 0's replay.   usage (GPU box, repo root): bash tools/round6_probe1.sh <tag>
tag=${1:-r06p1}; out=/root/repo/gpurun_out/$tag; mkdir -p $out
cd /root/repo
timeout 300 python bench.py --no-cpu-baseline > $out/bench_n1.json 2> $out/bench_n1.err; tail -c 1500 $out/bench_n1.json
timeout 200 python tools/dgemm_shard.py 8 "0,8" > $out/shard8.txt 2>&1; cat $out/shard8.txt
timeout 200 python tools/dgemm_shard.py 4 "0" > $out/shard4.txt 2>&1; cat $out/shard4.txt
timeout 200 python tools/dgemm_shard.py 2 "0" > $out/shard2.txt 2>&1; cat $out/shard2.txt
cd /tmp; export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats -d $out/kt -o kt --output-format csv -- python /root/repo/tools/scale_model.py --ranks 8 --steps 1 > $out/scale8.json 2> $out/scale8.err
python /root/repo/tools/trace_tail.py $out/kt
cat $out/scale8.json | head -50
