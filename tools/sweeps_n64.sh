#!/bin/bash
# the sweeps with 64 right-hand sides on the device clock under environment switches; usage: bash tools/sweeps_n64.sh <tag>
tag=${1:-n64}; cd /root/repo
for env in "" "STRUMPACK_AMD_SIDE_EARLY=1" "STRUMPACK_AMD_NO_SIDE_STREAM=1"; do
  echo "== ${env:-default}"
  env $env python tools/sweep_ab.py 100000 256 64 2>&1 | grep -v amdgpu.ids
done > gpurun_out/sweeps_n64_$tag.txt 2>&1
cat gpurun_out/sweeps_n64_$tag.txt
