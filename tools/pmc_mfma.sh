#!/bin/bash
# Matrix-core utilisation counters of the kernels north_star names (sketch GEMM, fused leaf update, batched GEMM, the
# many-right-hand-side sweeps), per kernel: separate rocprofv3 --pmc passes (counters only, with --kernel-trace) over one
# step of bench.py with 64 right-hand sides.  Summary -> gpurun_out/<tag>/pmc_mfma.json (copied to profiles/ by hand).
#   usage (GPU box, repo root): bash tools/pmc_mfma.sh <tag>
tag=${1:-r04_pmc}; out=/root/repo/gpurun_out/$tag; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; export STRUMPACK_AMD_BENCH_NO_PMC=1
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CU_CYCLES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $out/pass$i -o p --output-format csv -- python /root/repo/bench.py --steps 1 --warmup 0 --no-cpu-baseline --nrhs 64 > $out/pass$i.log 2>&1 || tail -3 $out/pass$i.log
done
python /root/repo/tools/pmc_mfma_summary.py $out $i > $out/pmc_mfma.json; head -c 3000 $out/pmc_mfma.json
