#!/bin/bash
# GPU test tier with an optional -k filter; usage (GPU box, repo root): bash tools/gpu_tests.sh <tag> ["<-k expression>"]
tag=${1:-t}; out=/root/repo/gpurun_out/$tag; mkdir -p $out; cd /root/repo
timeout 1500 python -m pytest tests -m gpu -q -x ${2:+-k "$2"} --durations=15 > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -40 $out/pytest.log
