#!/bin/bash
# Development aid (CPU only): compiles the product's kernel + host sources against the SIMT emulator (tests/emu) with
# AddressSanitizer + UBSan and runs the reference-shaped C++ driver (tests/cpp/test_HSS_seq.cpp) on a few CTest lines,
# including the SJLT sketch, the Schur complement calls and the child views.  usage: bash tools/asan_emu.sh [build dir]
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
B=${1:-/tmp/strumpack_amd_asan}
SRC=$ROOT/strumpack_amd/csrc
FLAGS="-O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -std=c++17 -fPIC -I$ROOT/tests/emu -I$ROOT/include -I$SRC/kernels -I$SRC/host -w"
mkdir -p $B && cd $B
for f in $SRC/kernels/*.hip; do g++ $FLAGS -x c++ -c $f -o k_$(basename $f .hip).o & done
for f in $SRC/host/*.cpp; do g++ $FLAGS -c $f -o h_$(basename $f .cpp).o & done
g++ $FLAGS -DEMU_SWAPCONTEXT -c $ROOT/tests/emu/emu_runtime.cpp -o emu_runtime.o &
wait
g++ $FLAGS $ROOT/tests/cpp/test_HSS_seq.cpp *.o -o test_HSS_seq_asan -lpthread
export ASAN_OPTIONS=detect_stack_use_after_return=0:detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 HSSK_EMU_THREADS=4
while read -r line; do
  echo "== $line"
  timeout 900 ./test_HSS_seq_asan $line 2>&1 | grep -E "ERROR|runtime error|AddressSanitizer|SUMMARY|# exiting"
done <<'LINES'
U 200 --hss_leaf_size 16 --hss_rel_tol 1e-1 --hss_abs_tol 1e-10 --hss_compression_algorithm stable --hss_d0 128 --hss_dd 4
U 300 --hss_leaf_size 32 --hss_rel_tol 1e-5 --hss_abs_tol 1e-10 --hss_compression_algorithm stable --hss_d0 16 --hss_dd 8 --hss_compression_sketch SJLT --hss_SJLT_algo perm --hss_nnz0 2 --hss_nnz 2
T 260 --hss_leaf_size 32 --hss_rel_tol 1e-6 --hss_compression_algorithm original --hss_d0 16 --hss_dd 8 --hss_compression_sketch SJLT --hss_SJLT_algo chunk --hss_nnz0 4 --hss_nnz 8
L 10 --hss_leaf_size 3 --hss_rel_tol 1e-5 --hss_abs_tol 1e-10 --hss_d0 32 --hss_dd 4
T 1 --hss_leaf_size 16
LINES
# the Python-driven cases (kernel level + C interface) on the same objects as a shared library
g++ -shared -fsanitize=address,undefined -o libstrumpack_amd_emu_asan.so k_*.o h_*.o emu_runtime.o -lpthread
LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) timeout 2400 python $ROOT/tools/asan_emu_cases.py $B/libstrumpack_amd_emu_asan.so 2>&1 | grep -E "ok$|ERROR|runtime error|AddressSanitizer|SUMMARY|Traceback|Error"
