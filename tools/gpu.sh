#!/bin/bash
# build the product library for gfx950, then run the given command on an MI355X box.
# The command runs under its own `timeout` (30 s short of gpurun's limit) with stdin closed, so a
# stray `head`/`cat` without a file or a hung kernel cannot eat the GPU budget.
# usage: tools/gpu.sh <timeout_s> '<command>'
set -e
cd "$(dirname "$0")/.."
make -C strumpack_amd/csrc -j8 2>&1 | grep -E " error|Error" -A5 && exit 1
printf '%s\n' "$2" > gpurun_cmd.sh
/usr/local/graft/bin/gpurun --timeout "$1" -- "timeout $(( $1 - 30 )) bash gpurun_cmd.sh < /dev/null"
