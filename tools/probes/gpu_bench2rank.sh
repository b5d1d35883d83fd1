#!/bin/bash
# functional check of bench.py's N > 1 code path on the one GPU of the test box: 2 ranks share the device over gloo
# (torch-callback process group); the native RCCL group cannot be created with two ranks on one device -- that attempt
# must end in the agreed fallback, not in a hang (bounded by timeout)
cd /root/repo; mkdir -p gpurun_out/$1; O=gpurun_out/$1
export STRUMPACK_AMD_SHARE_GPU=1 STRUMPACK_AMD_BACKEND=gloo
STRUMPACK_AMD_BENCH_COMM=torch timeout 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --size 16384 --steps 2 --warmup 1 > $O/b2_torch.json 2> $O/b2_torch.err; echo "torch-path rc=$?"; tail -c 600 $O/b2_torch.json; tail -3 $O/b2_torch.err
timeout 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 2 --size 16384 --steps 2 --warmup 1 > $O/b2_rccl.json 2> $O/b2_rccl.err; echo "rccl-attempt rc=$?"; tail -c 600 $O/b2_rccl.json; tail -5 $O/b2_rccl.err
