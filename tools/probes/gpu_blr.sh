#!/bin/bash
# One GPU call for the BLR frontal-matrix path: its GPU tests, the bench line, the rocprofv3 kernel summary of the same command.
# usage (GPU box, repo root): bash tools/gpu_blr.sh <tag> [front-n]
tag=${1:-blr}; fn=${2:-64}
out=/root/repo/gpurun_out/$tag
mkdir -p $out
cd /root/repo
timeout 900 python -m pytest tests -m gpu -q -x -k "blr" > $out/pytest_blr.log 2>&1; echo "pytest rc=$?"; tail -5 $out/pytest_blr.log
timeout 600 python bench.py --workload blr_front --front-n $fn > $out/bench_blr_front.json 2> $out/bench_blr_front.err; echo "bench rc=$?"
tail -c 3000 $out/bench_blr_front.json; tail -5 $out/bench_blr_front.err
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $out/kt -o kt --output-format csv -- python /root/repo/bench.py --workload blr_front --front-n $fn --steps 2 --warmup 1 --no-cpu-baseline > $out/bench_under_rocprof.json 2> $out/kt.err
head -30 $out/kt/kt_kernel_stats.csv | cut -c1-170
find $out/kt -name "*kernel_trace.csv" -delete
