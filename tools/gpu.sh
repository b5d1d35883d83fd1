#!/bin/bash
# build the product library for gfx950, then run the given command on an MI355X box
# usage: tools/gpu.sh <timeout_s> '<command>'
set -e
cd "$(dirname "$0")/.."
make -C strumpack_amd/csrc -j8 2>&1 | grep -E " error|Error" -A5 && exit 1
/usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
