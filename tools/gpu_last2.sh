#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/$1; O=gpurun_out/$1
(cd /tmp; export TMPDIR=/tmp; timeout 40 rocprofv3 --kernel-trace --stats -d /root/repo/$O/kt -o kt --output-format csv -- python /root/repo/bench.py --no-cpu-baseline --steps 4 --warmup 1 > /root/repo/$O/bench_under_rocprof.json 2> /root/repo/$O/kt.err); echo "rocprof rc=$?"
timeout 25 python tools/time_typed_upload.py 16384 > $O/typed_upload.json 2> $O/typed_upload.err; echo "typed rc=$?"; cat $O/typed_upload.json
