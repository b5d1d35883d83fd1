#!/bin/bash
# Round 5: launch-by-launch timeline of the BLR root front (200 x 200 plane, dsep 40000) and of the dsep-4096 front.
# usage (GPU box, repo root): bash tools/round5_blr_trace.sh <tag>
tag=${1:-r05_blr}; out=/root/repo/gpurun_out/$tag; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; export STRUMPACK_AMD_BENCH_NO_PMC=1
i=0
for args in "--workload blr_front --front-n 200 --front-upd none --steps 2 --warmup 1" "--workload blr_front --front-n 64 --steps 3 --warmup 1"; do
  i=$((i+1))
  timeout 500 rocprofv3 --kernel-trace --stats -d $out/kt$i -o kt --output-format csv -- python /root/repo/bench.py --no-cpu-baseline $args > $out/bench$i.json 2> $out/kt$i.err
  python /root/repo/tools/trace_tail.py $out/kt$i > /dev/null 2>&1
  cp $out/kt$i/kt_kernel_stats.csv $out/kernel_stats$i.csv 2>/dev/null
  [ -f $out/kt$i/trace_tail.txt ] && cp $out/kt$i/trace_tail.txt $out/trace_tail$i.txt
  rm -rf $out/kt$i
  head -16 $out/kernel_stats$i.csv | cut -c1-170
done
