#!/bin/bash
# rocprofv3 kernel summaries of bench.py invocations; usage (GPU box, repo root): bash tools/gpu_prof.sh <tag> "<bench args>" ["<bench args>" ...]
tag=${1:-p}; shift; out=/root/repo/gpurun_out/$tag; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
i=0
for args in "$@"; do
  i=$((i+1))
  timeout 400 rocprofv3 --kernel-trace --stats -d $out/kt$i -o kt --output-format csv -- python /root/repo/bench.py --no-cpu-baseline $args > $out/bench$i.json 2> $out/kt$i.err
  echo "== $args"; head -28 $out/kt$i/kt_kernel_stats.csv | cut -c1-150
  python /root/repo/tools/trace_tail.py $out/kt$i > /dev/null 2>&1
done
